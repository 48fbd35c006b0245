#!/bin/bash
# Does the number of HSA hardware queues HIP multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4) bound the throughput bench?
# usage: tools/gpu_hwq.sh <tag>     (headline only, in-flight 4 / 6 / 8 under 4 / 8 queues)
TAG=${1:-hwq}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for Q in 4 8; do for F in 4 6 8; do
  GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py --no-cpu-baseline --no-quatro --no-extras --steps 192 --in-flight $F --pairs 16 > $OUT/q${Q}_f$F.json 2> $OUT/q${Q}_f$F.err
  python - $Q $F $OUT/q${Q}_f$F.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[3]).readline()); c=j["config"]
    print("queues %s in-flight %s  value %8.1f  ms_per_step %.4f  single %.3f" % (sys.argv[1], sys.argv[2], j["value"], j["ms_per_step"], c["ms_per_registration_single_stream"]))
except Exception as e:
    print(sys.argv[1:3], "FAILED", e)
PY
done; done
