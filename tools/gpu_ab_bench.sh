#!/bin/bash
# A/B of debug knobs on the full bench line (developer tool): usage tools/gpu_ab_bench.sh '<json knobs>' ['<json knobs>' ...]
mkdir -p gpurun_out/ab
for kn in "$@"; do
  QN_DEBUG_KNOBS="$kn" timeout 300 python bench.py --no-quatro --no-cpu-baseline > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err
  echo "knobs $kn"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/ab/b.json').read().strip().splitlines()[-1])
c=d['config']
print("  value", d['value'], "single", c['ms_per_registration_single_stream'], "host", c['ms_per_registration_from_host_buffers'], "align", c['ms_per_align'], "ov80", c['overlap80']['registrations_per_s'], "ov80 align", c['overlap80']['ms_per_align_stats']['median'], "rop", c['reference_operating_point']['100k']['gpu_ms_from_host_buffers']['median'], "b64", c['batch64']['pairs_per_s'])
PY
done
