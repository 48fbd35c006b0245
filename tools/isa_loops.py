"""Static instruction counts per loop of one kernel (no GPU needed): compiles an instantiation group of csrc/qn_inst.hip (or another
translation unit) to gfx950 assembly and prints, for every backward branch, the VALU / SALU / LDS / VMEM instructions of the loop body.
usage: python tools/isa_loops.py <mangled-name-substring> [group=1] [unit=qn_inst.hip]
e.g.   python tools/isa_loops.py k_knn_histILb0ELi32 1"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]; group = sys.argv[2] if len(sys.argv) > 2 else "1"; unit = sys.argv[3] if len(sys.argv) > 3 else "qn_inst.hip"
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                       "-DQN_INST_GROUP=" + group, "--cuda-device-only", "-S", unit, "-o", out], cwd=os.path.join(ROOT, "fast-lio-sam-qn_amd", "csrc"), stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(want), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
print(lines[start].split(":")[0])
blocks = []; cur = ["entry", start, 0, 0, 0, 0, []]; blocks.append(cur)
for i in range(start + 1, end):
    l = lines[i]; m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: cur = [m.group(1), i, 0, 0, 0, 0, []]; blocks.append(cur); continue
    t = l.strip()
    if not t or t[0] in ";.": continue
    op = t.split()[0]; cur[6].append(t)
    if op.startswith("v_"): cur[2] += 1
    elif op.startswith("s_"): cur[3] += 1
    elif op.startswith("ds_"): cur[4] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur[5] += 1
print("static: valu %d salu %d lds %d vmem %d, %d basic blocks" % (*[sum(b[k] for b in blocks) for k in (2, 3, 4, 5)], len(blocks)))
lab = {b[0]: k for k, b in enumerate(blocks)}; seen = set()
for k, b in enumerate(blocks):
    for t in b[6]:
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in lab and lab[m.group(1)] <= k and (lab[m.group(1)], k) not in seen:
            j = lab[m.group(1)]; seen.add((j, k))
            print("loop %-12s .. %-12s asm line %6d: valu %4d salu %4d lds %3d vmem %2d" % (m.group(1), b[0], blocks[j][1] + 1, *[sum(x[c] for x in blocks[j:k + 1]) for c in (2, 3, 4, 5)]))
