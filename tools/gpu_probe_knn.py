"""Developer probe: cost of the k-NN + covariance stage under different knobs (grid cell size, first radius, rounds)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = int(os.environ.get("N", "100000"))
src, tgt, T = synth.make_pair(int(os.environ.get("PAIR", "0")), N)
if os.environ.get("WHICH") == "tgt": src = tgt
if os.environ.get("OUTLIERS"):
    rng = np.random.default_rng(3); m = int(float(os.environ["OUTLIERS"]) * N); lo, hi = src.min(0), src.max(0); hi[2] = lo[2] + 25.0
    src = src.copy(); src[rng.choice(N, m, replace=False)] = rng.uniform(lo, hi, size=(m, 3)).astype(np.float32)
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(int(os.environ.get("K", "20")))
g.setInputSource(src)
cell0 = ctx.grid_info(0)["cell"]
print("auto cell %.4f dims %s" % (cell0, ctx.grid_info(0)["dims"]))
def counters():
    out = (C.c_uint32 * 16)(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); return list(out)
def run(label, **kn):
    for k, v in kn.items(): ctx.debug_set(k, v)
    if "cell" in kn: g.setInputSource(src)
    for _ in range(3): g.calculateSourceCovariances()
    ctx.synchronize()
    ctx.debug_set("dbg_counters", 1); g.calculateSourceCovariances(); ctx.synchronize(); c = counters(); ctx.debug_set("dbg_counters", 0)
    best = 1e9
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.calculateSourceCovariances()
        ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print("%-44s %.3f ms  clusters %6d cand %9d retries %6d list %6d  dbg8-11 %s" % (label, best, c[0], c[1], c[3], c[4], c[8:12]))
if os.environ.get("SWEEP"):
    for cf in (0.8, 0.9, 1.0, 1.15, 1.3):
        for m in (1.5, 1.75, 2.0, 2.25, 2.5):
            run("cell x%.2f margin %.2f" % (cf, m), cell=cell0 * cf, knn_hist=1, margin_knn=m * 1.0, knn_rounds=2)
else:
    run("default", knn_hist=1)
    for m in (0, 2.0, 2.5):
        run("single_all margin %.1f" % m, knn_hist=1, knn_single_all=1, margin_knn=m)
    ctx.debug_set("knn_single_all", 0); ctx.debug_set("margin_knn", 0)
    for m in (2.0, 2.5, 3.0):
        for rd in (1, 2, 3):
            run("margin %.2f rounds %d" % (m, rd), knn_hist=1, margin_knn=m, knn_rounds=rd)
