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


N_INST_GROUPS = 11     # = QN_NUM_INST_GROUPS in csrc/qn_instances.h


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h", ".inc"))] + [os.path.join(ROOT, "include", "qn_engine.h")]


def csrc_sha1():
    """sha1 over the kernel / host sources of the library (names + contents, sorted): what a stored profile is tagged with, so that bench.py can tell whether
    profiles/pmc_latest.json and profiles/valu_budget_latest.json were collected on THIS source state (`stale` in the bench line)."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(sources() + _deps()):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_stale(src, obj) or os.path.getmtime(obj) > t for src, obj, _ in _units())


def _units():
    """(source, object, extra flags): qn_inst.hip is compiled once per instantiation group (csrc/qn_instances.h)."""
    units = []
    for src in sources():
        if os.path.basename(src) == "qn_inst.hip":
            units += [(src, src[:-4] + "_g%d.o" % g, ["-DQN_INST_GROUP=%d" % g]) for g in range(1, N_INST_GROUPS + 1)]
        else:
            units.append((src, src[:-4] + ".o", []))
    return units


def _stale(src, obj):
    """Per-unit dependencies from the compiler's own -MD file (so the minutes-long k-NN units are rebuilt only when
    qn_knn_kernels.cuh / qn_device.cuh change)."""
    if not os.path.exists(obj) or not os.path.exists(obj + ".d"):
        return True
    t = os.path.getmtime(obj)
    deps = [w for w in open(obj + ".d").read().replace("\\\n", " ").split()[1:] if os.path.abspath(w).startswith(ROOT)]
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in [src] + deps)


def build(force=False, verbose=False, jobs=None):
    """One object per translation unit, compiled in parallel (the search kernels take minutes each; they are split into
    explicit-instantiation units for that reason), cached by mtime, then one link into the in-tree libqn_engine.so."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    cflags = [f for f in FLAGS if f != "-shared"]
    todo, objs = [], []
    for src, obj, extra in _units():
        objs.append(obj)
        if force or _stale(src, obj):
            todo.append([HIPCC] + cflags + extra + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=jobs or max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(run, todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
