"""Developer probe: per-stage device time of one registration (single stream) + batched throughput, no oracle."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))
import subprocess
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-quatro", "--steps", "48", "--warmup", "8"], capture_output=True, text=True)
d = json.loads(out.stdout.strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f reg/s  ms/step %.4f  single %.4f  align %.4f" % (d["value"], d["ms_per_step"], d["config"]["ms_per_registration_single_stream"], d["config"]["ms_per_align"]))
print(json.dumps(r["family_ms_per_registration"]))
print({k: v["avg_launch_ms"] for k, v in r["kernels"].items()})
