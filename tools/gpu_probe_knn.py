"""Developer probe (GPU box): what the k-NN selection pass does per wave - clusters, candidates and retries (debug counters 0 / 1 / 3) for one 100k cloud.
usage: python tools/gpu_probe_knn.py [pair_id=0] [k=20]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np
from qn_amd import engine, synth
N = 100000
pid = int(sys.argv[1]) if len(sys.argv) > 1 else 0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
src, tgt, T = synth.make_pair(pid, N)
ctx = engine.Context(N + 1024)
g = engine.NanoGICP(ctx)
g.setCorrespondenceRandomness(k)
def counters():
    out = (C.c_uint32 * 16)(); ctx.check(ctx._l.qn_debug_get_counters(ctx.h, out)); return list(out)
g.setInputSource(src)
ctx.debug_set("dbg_counters", 1); g.calculateSourceCovariances(); ctx.synchronize(); c = counters(); ctx.debug_set("dbg_counters", 0)
groups = (N + 15) // 16
print("k = %d: %d groups of 16 queries; sum of clusters over rounds %d (%.2f per group), candidates %d (%.1f per group, %.1f per cluster-round), retried lanes %d, list-pass entries %d, one-per-wave %d"
      % (k, groups, c[0], c[0] / groups, c[1], c[1] / groups, c[1] / max(1, c[0]), c[3], c[4], c[8]))
ctx.close()
