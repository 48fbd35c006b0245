#!/bin/bash
# A/B of engine knobs on the GPU box: usage [BENCH_ARGS='--shift 24'] [SKIP_TESTS=1] tools/gpu_ab.sh <tag> '<json knobs 1>' '<json knobs 2>' ...   (bench headline only)
TAG=${1:-ab}; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_gpu_gicp.py tests/test_golden.py tests/test_gpu_adversarial.py -m gpu -q -x > $OUT/pytest_quick.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_quick.log
i=0
for K in "$@"; do
  i=$((i+1))
  QN_DEBUG_KNOBS="$K" timeout 300 python bench.py --no-cpu-baseline --no-quatro --no-extras --steps 200 $BENCH_ARGS > $OUT/ab_$i.json 2> $OUT/ab_$i.err
  python - "$K" $OUT/ab_$i.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).readline()); c=j["config"]; r=j["roofline"]
    print("%-40s value %8.1f  single %.3f  align %.3f  tick %.4f ms  fam %s" % (sys.argv[1], j["value"], c["ms_per_registration_single_stream"], c["ms_per_align"], r["kernels"].get("gn_tick_fused",{}).get("avg_launch_ms",0), {k:round(v,3) for k,v in r["family_ms_per_registration"].items() if k in ("gn_tick_fused","solve","knn_select","nn_search","nn_fallback")}))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
