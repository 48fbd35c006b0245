"""VALU issue budget of one registration: SQ_ACTIVE_INST_VALU per launch (tools/gpu_sq.sh -> sq_counters.json) x launches per registration
(tools/trace_summary.py -> last_registration_trace.txt).  A quad-cycle of VALU-active is one wave64 VALU instruction on one SIMD; the chip
offers 1024 SIMDs x clock/4 of them per second.  usage: python tools/valu_budget.py <sq_counters.json> <last_registration_trace.txt> [clock_GHz=2.1]"""
import collections, json, re, sys
d = json.load(open(sys.argv[1])); clock = float(sys.argv[3]) if len(sys.argv) > 3 else 2.1
cnt = collections.Counter()
for l in open(sys.argv[2]):
    m = re.search(r"gap\s+[-\d.]+\s+(\S.*)$", l)
    if m: cnt[m.group(1).strip().split("(")[0].strip()] += 1
rows = []; tot = 0.0
for name, v in d.items():
    short = name.replace("qn::", "").split("(")[0].strip(); n = cnt.get(short, 0); va = v.get("SQ_ACTIVE_INST_VALU", 0.0)
    rows.append((va * n, short, n, va, v.get("SQ_INSTS_VALU", 0.0), v.get("SQ_WAVES", 0.0), v.get("SQ_WAVE_CYCLES", 0.0), v.get("SQ_INSTS_SALU", 0.0))); tot += va * n
for r in sorted(rows, reverse=True)[:12]:
    print("%-30s x%2d  %6.2f M VALU quad-cycles per launch  %5.1f %% of the registration  VALU/wave %5.0f  SALU/wave %5.0f  VALU-active / wave lifetime %4.1f %%" % (
        r[1][:30], r[2], r[3] / 1e6, 100 * r[0] / tot, r[4] / max(r[5], 1), r[7] / max(r[5], 1), 100 * r[3] / max(r[6], 1)))
print("one registration: %.1f M VALU quad-cycles = %.0f us of a chip whose 1024 SIMDs issue VALU every cycle (%.1f GHz)" % (tot / 1e6, tot / 1024 / (clock * 1e9 / 4) * 1e6, clock))
