#!/bin/bash
# SQ counter passes over quatro::align at 30k (one pair, one context): per-kernel VALU / wait figures of the FPFH and matching kernels.  usage: tools/gpu_sq_quatro.sh <tag> [npts]
# Counters only with --kernel-trace (never with sys/hip/hsa traces: gpurun refuses that combination).
TAG=${1:-sqq}; N=${2:-30000}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/q1.py <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import numpy as np, torch
torch.cuda.init()
from qn_amd import engine, synth
npts = int(sys.argv[1])
qs, qt, _ = synth.make_pair(402 if npts <= 30000 else 400 + npts // 1000, npts, mode="quatro")
ctx = engine.Context(npts + 1024); ctx.debug_set("pair_pipeline", 0)
q = engine.Quatro(ctx)
for _ in range(4): q.align(qs, qt)
PY
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $OUT/p$i -o p -- python /tmp/q1.py $N > /dev/null 2> $OUT/p$i.err; echo "pass $i exit $?"
  find $OUT/p$i -name '*counter_collection.csv' -exec cp {} $OUT/sq_pass$i.csv \;
  rm -rf $OUT/p$i
done
python tools/sq_summary.py $OUT > $OUT/sq_counters.json
python - <<PY
import json
d = json.load(open("$OUT/sq_counters.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    w = max(v.get("SQ_WAVES", 1), 1); wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print("%-44s waves %7.0f  VALU/wave %6.0f SALU %5.0f LDS %5.0f VMEM_RD %5.0f SMEM %4.0f | wave life %7.0f quad-cyc  VALU-active %4.1f %%  wait-any %4.1f %%  busy %8.0f" % (
        k[:44], w, v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w, v.get("SQ_INSTS_VMEM_RD", 0) / w, v.get("SQ_INSTS_SMEM", 0) / w,
        wc / w, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_BUSY_CYCLES", 0)))
PY
