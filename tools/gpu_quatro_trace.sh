#!/bin/bash
# kernel timeline of the LAST quatro::align of tools/gpu_quatro_stage.py's 30k MFMA leg (rocprofv3 --kernel-trace).  usage: tools/gpu_quatro_trace.sh <tag>
TAG=${1:-qtrace}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/q30.py <<'PY'
import os, sys, time, json
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
npts = int(sys.argv[1])
qs, qt, _ = synth.make_pair(400 + npts // 1000, npts, mode="quatro")
ctx = engine.Context(npts + 1024)
q = engine.Quatro(ctx)
for _ in range(4): q.align(qs, qt)
lat = []
for _ in range(5):
    t0 = time.perf_counter(); T, valid = q.align(qs, qt); lat.append(1e3 * (time.perf_counter() - t0))
print("align ms", np.median(lat), valid)
PY
for N in 30000 100000; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$N -o k -- python /tmp/q30.py $N > $OUT/run$N.log 2>&1
  find $OUT/p$N -name '*kernel_trace.csv' -exec cp {} $OUT/trace$N.csv \;
  rm -rf $OUT/p$N
  tail -2 $OUT/run$N.log
  python tools/trace_summary.py $OUT/trace$N.csv > $OUT/timeline$N.txt; tail -75 $OUT/timeline$N.txt
done
