"""align()-only wall time as a function of the forced iteration count: slope = cost of one tracked tick in the chain, intercept = unseeded ticks + fixed overhead."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
N = 100000
src, tgt, T = synth.make_pair(0, N)
ctx = engine.Context(N + 1024)
for kk, vv in json.loads(os.environ.get("QN_DEBUG_KNOBS", "{}")).items():
    ctx.debug_set(kk, float(vv))
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer(sys.argv[1] if len(sys.argv) > 1 else "gn")
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
res = {}
for it in (1, 2, 3, 4, 6, 10, 20, 30):
    g.setMaximumIterations(it); g.setForceIterations(it)
    for _ in range(3):
        g.align()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); g.align(); ts.append(1e3 * (time.perf_counter() - t0))
    res[it] = round(float(np.median(ts)), 4)
print(res)
print("per tick 10->30: %.2f us; 4->10: %.2f us" % (1e3 * (res[30] - res[10]) / 20, 1e3 * (res[10] - res[4]) / 6))
# device-clock probe of the tracked ticks (100 MHz wall clock): where inside the kernel, and between kernels, the time goes
import ctypes as C
ctx.debug_set("clk_probe", 1)
g.setMaximumIterations(20); g.setForceIterations(20)
g.align(); ctx.debug_set("clk_probe", 1); g.align()
buf = (C.c_ulonglong * (256 * 8 + 1024 * 12))(); n = C.c_uint32()
ctx.check(ctx._l.qn_debug_get_clk(ctx.h, buf, C.byref(n)))
extra = np.array(buf[256 * 8 + 1024 * 8:], dtype=np.uint64).astype(np.uint32).view(np.float32).reshape(1024, 4)
blk_raw = np.array(buf[256 * 8:256 * 8 + 1024 * 8], dtype=np.uint64).reshape(1024, 8)
blk = blk_raw.astype(np.int64)
a = np.array(buf[:256 * 8], dtype=np.uint64).reshape(256, 8)[:n.value]
raw7 = a[:, 7].copy(); a = a.astype(np.int64)
t0 = a[0, 0]
print("tick: start  +rows  +ctrl  +nn  +search  +end   (us, 100 MHz clock) | period | first block start, last block end (rel. block 0 start)")
for i in range(len(a)):
    r = a[i]; per = (a[i + 1, 0] - r[0]) / 100.0 if i + 1 < len(a) else 0
    first = int(~raw7[i] & np.uint64(0xFFFFFFFFFFFFFFFF)) if raw7[i] else int(r[0])
    print("%3d: %8.2f %6.2f %6.2f %6.2f %6.2f %6.2f | %6.2f | %6.2f %6.2f" % (i, (r[0] - t0) / 100.0, (r[1] - r[0]) / 100.0, (r[2] - r[1]) / 100.0, (r[3] - r[2]) / 100.0, (r[4] - r[3]) / 100.0, (r[5] - r[4]) / 100.0, per,
          (int(first) - r[0]) / 100.0, (r[6] - r[0]) / 100.0))

keep = blk[:, 0] > 0; blk = blk[keep]; blk_raw = blk_raw[keep]; extra = extra[keep]; b0 = blk[:, 0].min()
print("per-block stamps of the latest tick (%d blocks), us rel. earliest start:" % len(blk))
for q in (0, 10, 50, 90, 100):
    print("  p%-3d start %6.2f  prologue %6.2f  nn %6.2f  emit %6.2f  end %6.2f" % (q, np.percentile(blk[:, 0] - b0, q) / 100, np.percentile(blk[:, 1] - blk[:, 0], q) / 100,
          np.percentile(blk[:, 2] - blk[:, 1], q) / 100, np.percentile(blk[:, 3] - blk[:, 2], q) / 100, np.percentile(blk[:, 3] - b0, q) / 100))
order = np.argsort(blk[:, 3])[-8:]
print("  slowest blocks:", [(int(i), round((blk[i, 0] - b0) / 100, 2), round((blk[i, 1] - blk[i, 0]) / 100, 2), round((blk[i, 2] - blk[i, 1]) / 100, 2), round((blk[i, 3] - blk[i, 2]) / 100, 2), int(blk[i, 4]), int(blk[i, 5]), int(blk[i, 6])) for i in order])
print('  big / rescanned / far lanes over all blocks:', int(blk[:, 4].sum()), int(blk[:, 5].sum()), int(blk[:, 6].sum()), ' blocks with any big lane:', int((blk[:, 4] > 0).sum()))

for i in range(len(blk_raw)):
    if blk_raw[i, 4] or blk_raw[i, 5]:
        print("  block %d: t %d j0 %d  d0 %.6g (sqrt %.6g)  ref.w %.6g  delta %.3g  r %.4g" % (i, int(blk_raw[i, 7] >> np.uint64(32)), int(np.int32(blk_raw[i, 7] & np.uint64(0xFFFFFFFF))), extra[i, 0], np.sqrt(extra[i, 0]), extra[i, 1], extra[i, 2], extra[i, 3]))
