"""Build recipe: hipcc cross-compiles csrc/ for gfx950 into an IN-TREE shared library
(fast-lio-sam-qn_amd/libqn_engine.so) so that it travels to the GPU box with the repo snapshot."""
import os
import subprocess

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libqn_engine.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the f32 distance / transform arithmetic must be plain mul+add in source order
# (bit-for-bit the reference's non-FMA x86 build; see DESIGN.md "numerics").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
         "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include")]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h", ".inc"))] + [os.path.join(ROOT, "include", "qn_engine.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _deps())


def build(force=False, verbose=False):
    """One object per .hip translation unit (cached by mtime: qn_cloud.hip pulls in hipCUB's radix sort, minutes to
    compile), then one link into the in-tree libqn_engine.so."""
    if not force and not needs_build():
        return LIB
    objs = []
    cflags = [f for f in FLAGS if f != "-shared"]
    for src in sources():
        obj = src[:-4] + ".o"
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(p) for p in [src] + _deps())
        if stale:
            cmd = [HIPCC] + cflags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
