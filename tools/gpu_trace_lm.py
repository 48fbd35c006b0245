import os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
from qn_amd import engine, synth
N = int(os.environ.get("N", "100000")); src, tgt, T = synth.make_pair(5, N)
ctx = engine.Context(N + 1024)
for _ in range(3): r = engine.icp_alignment(ctx, src, tgt)
print(r["iterations"], r["valid"])
