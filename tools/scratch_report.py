"""Which kernels of libqn_engine.so use scratch (private segment)?  On gfx950 a dispatch that needs scratch costs ~5 us more at the kernel
boundary than one that does not (tools/micro/launch_gap2.hip), so kernels in the per-registration chain must not spill or index local arrays
dynamically.  Reads the code objects' metadata notes; no GPU needed."""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "fast-lio-sam-qn_amd", "libqn_engine.so")
tmp = tempfile.mkdtemp()
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, tmp + "/fat.bin"])
d = open(tmp + "/fat.bin", "rb").read()
pos = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)]
txt = ""
for i, p in enumerate(pos):
    b = d[p:(pos[i + 1] if i + 1 < len(pos) else len(d))]
    ne = struct.unpack_from("<Q", b, 24)[0]; o = 32
    for e in range(ne):
        off, size, tl = struct.unpack_from("<QQQ", b, o); o += 24
        trip = b[o:o + tl].decode(); o += tl
        if "gfx950" in trip and size > 0:
            f = "%s/co%d_%d.elf" % (tmp, i, e); open(f, "wb").write(b[off:off + size])
            txt += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
rows = []
for e in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, e) else 0
    rows.append((re.search(r"\.name:\s+(\S+)", e).group(1), g("private_segment_fixed_size"), g("vgpr_count"), g("group_segment_fixed_size"), g("vgpr_spill_count")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
show_all = "--all" in sys.argv
for (r, nm) in sorted(zip(rows, names), key=lambda x: -x[0][1]):
    if r[1] > 0 or show_all:
        print("%6d B scratch  vgpr %3d  lds %6d  spill %3d  %s" % (r[1], r[2], r[3], r[4], nm[:130]))
print(len(rows), "kernels,", sum(1 for r in rows if r[1] > 0), "with scratch")
