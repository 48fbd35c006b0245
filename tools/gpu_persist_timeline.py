"""developer probe: wall-clock timeline of the persistent align kernel (knob persist_probe), BASELINE configs[1] workload"""
import sys, os, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
import torch; torch.cuda.init()
from qn_amd import engine, synth
N = int(os.environ.get("N", "100000"))
src, tgt, _ = synth.make_pair(0, N, shift=float(os.environ["SHIFT"]) if "SHIFT" in os.environ else None)
ctx = engine.Context(N + 1024)
for k, v in json.loads(os.environ.get("QN_DEBUG_KNOBS", "{}")).items(): ctx.debug_set(k, float(v))
ctx.debug_set("persist_probe", 1)
g = engine.NanoGICP(ctx); g.setCorrespondenceRandomness(20); g.setMaximumIterations(20); g.setMaxCorrespondenceDistance(52.5); g.setOptimizer("gn"); g.setForceIterations(20)
g.setInputSource(src); g.calculateSourceCovariances(); g.setInputTarget(tgt); g.calculateTargetCovariances()
for _ in range(3): g.align()
buf = np.zeros(64 * 16 + 16, dtype=np.uint64)
ctx.check(ctx._l.qn_debug_get_persist_clk(ctx.h, buf.ctypes.data_as(C.c_void_p)))
t0 = int(buf[64 * 16]); t00 = t0; clk = buf[:64 * 16].reshape(64, 16).astype(np.int64)
us = lambda x: (int(x) - t0) / 100.0 if x else float("nan")
print("persist launches", ctx.debug_get("persist_launches"), " (times in us since the launch start of block 0; 100 MHz clock)")
print("tick | reducer: rows-in  sums  ctrl-done  published | worker0: pose-seen body-done row-out | last worker: pose-seen body-done row-out")
prev = None
for gi in range(24):
    r = clk[gi]
    if not r.any(): break
    line = "%3d  | %8.2f %8.2f %8.2f %8.2f | %8.2f %8.2f %8.2f | %8.2f %8.2f %8.2f" % (gi, us(r[0]), us(r[1]), us(r[2]), us(r[3]), us(r[4]), us(r[5]), us(r[6]), us(r[8]), us(r[9]), us(r[10]))
    if prev is not None and r[3] and prev[3]: line += "   period %.2f" % ((int(r[3]) - int(prev[3])) / 100.0)
    if r[11]: line += "  | body-done over blocks: first %.2f (blk %d) last %.2f (blk %d); reducer lane0 saw its rows %.2f (first arrival %.2f, %d spins); worker0 stores acked %.2f" % ((int(r[12]) >> 16) / 100.0 + (t00 - t0) / 100.0, int(r[12]) & 0xffff, (int(r[11]) >> 16) / 100.0 + (t00 - t0) / 100.0, int(r[11]) & 0xffff, us(r[7]), us(r[15]), int(r[14]), us(r[13]))
    print(line); prev = r
