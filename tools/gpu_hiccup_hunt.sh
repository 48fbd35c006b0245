#!/bin/bash
# runs the driver's bench invocation repeatedly with the batched path's host timeline on stderr; keeps the logs of runs whose first region was slow
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/hunt
for i in $(seq 1 ${1:-14}); do
  QN_DEBUG_KNOBS='{"batch_trace":1}' timeout 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-quatro --repeats 4 > gpurun_out/hunt/run$i.json 2> gpurun_out/hunt/run$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/hunt/run$i.json")); v=[d["value"]]+d["config"]["value_repeats"]["values"]
print("run $i", [round(x) for x in v], "SLOW" if min(v) < 0.8*max(v) else "")
PY
done
