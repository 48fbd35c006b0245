"""VALU issue budget of one registration on the batched path: SQ_ACTIVE_INST_VALU summed over every launch of a tools/gpu_sq_batch.sh run / the registrations of
that run.  A quad-cycle of VALU-active is one wave64 VALU instruction on one SIMD; the chip offers 1024 SIMDs x clock / 4 of them per second.
usage: python tools/valu_budget_batch.py <dir with sq_pass*.csv> <registrations in the run> <tag> [clock_GHz=2.1]  -> table on stdout, <dir>/valu_budget.json"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-lio-sam-qn_amd"))
from qn_amd.build import csrc_sha1
from collections import defaultdict
d, nreg, tag = sys.argv[1], float(sys.argv[2]), sys.argv[3]; clock = float(sys.argv[4]) if len(sys.argv) > 4 else 2.1
tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(d, "sq_pass*.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("qn::k_lanes<qn::", "").replace("qn::", "").rstrip(" >")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
rows = []; total = 0.0
for k, v in tot.items():
    va = v.get("SQ_ACTIVE_INST_VALU", 0.0) / nreg; total += va
    rows.append((va, k, cnt[k].get("SQ_ACTIVE_INST_VALU", 0) / nreg, v))
for va, k, n, v in sorted(rows, reverse=True)[:14]:
    w = max(v.get("SQ_WAVES", 0.0), 1.0)
    print("%-34s %5.2f launches/reg  %6.2f M VALU quad-cycles per registration  %5.1f %%   VALU/wave %5.0f  SALU/wave %5.0f  LDS/wave %4.0f  VMEM_RD/wave %4.0f  VALU-active / wave lifetime %4.1f %%  wait-any / lifetime %4.1f %%" % (
        k[:34], n, va / 1e6, 100 * va / max(total, 1), v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w, v.get("SQ_INSTS_VMEM_RD", 0) / w,
        100 * v.get("SQ_ACTIVE_INST_VALU", 0) / max(v.get("SQ_WAVE_CYCLES", 0), 1), 100 * v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 0), 1)))
us = total / 1024 / (clock * 1e9 / 4) * 1e6
print("one registration: %.1f M VALU quad-cycles = %.0f us of a chip whose 1024 SIMDs issue VALU every cycle (%.1f GHz); run: %d registrations, batched path (1 context x 8 lanes)" % (total / 1e6, us, clock, nreg))
json.dump({"csrc_sha1": csrc_sha1(), "quad_cycles_per_registration": round(total, 0), "clock_ghz": clock, "us_of_a_fully_issuing_chip": round(us, 1), "source": "profiles/%s_valu_budget.txt (tools/gpu_sq_batch.sh: rocprofv3 --pmc SQ_ACTIVE_INST_VALU over the batched path, 1 context x 8 lanes)" % tag,
           "per_kernel_quad_cycles_per_registration": {k: round(va, 0) for va, k, n, v in sorted(rows, reverse=True)[:14]}}, open(os.path.join(d, "valu_budget.json"), "w"), indent=1)
