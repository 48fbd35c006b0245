mkdir -p gpurun_out/r6_a5
timeout 900 python bench.py > gpurun_out/r6_a5/bench.json 2> gpurun_out/r6_a5/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_a5/bench.json'))
c=d['config']
print('value',d['value'],'overlap80',d.get('value_overlap80'))
print('repeats',c.get('value_repeats'))
b=c.get('batch64',{})
for k in ('pairs_per_s','reference_operating_point','shared_query','per_rank_share_at_8','coarse_to_fine'):
    print(k, json.dumps(b.get(k))[:600])
for k in ('ms_per_registration_single_stream','ms_per_align','ms_per_registration_from_host_buffers'):
    print(k, json.dumps(c.get(k))[:300])
PY
