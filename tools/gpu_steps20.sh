#!/bin/bash
# the driver's invocation (--steps 20 --warmup 5) for several contexts x lanes settings.  usage: tools/gpu_steps20.sh ["3 8" "2 10" ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/steps20
CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=("3 8" "2 10" "4 5" "1 20" "2 16" "3 7")
for cfg in "${CFGS[@]}"; do set -- $cfg
  for rep in 1 2; do
  timeout 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-quatro --in-flight $1 --lanes $2 --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1x$2', d['value'], d['config']['value_repeats']['values'])"
  done
done
