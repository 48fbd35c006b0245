"""Per-kernel HBM traffic of the BATCHED path (k_lanes<F> launches of tools/gpu_batch_sweep.py 1x8) from the rocprofv3 --pmc passes of tools/gpu_round4.sh.
usage: python tools/pmc_summary_batch.py gpurun_out/<tag> <lanes>     (reads pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv; prints JSON with a "batched" section)
Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): separate passes (TCC slot limit), KB, FETCH_SIZE x2 on gfx950.
A batched launch carries `entries` registrations (or clouds): the per-registration-launch figure is the launch's traffic / entries."""
import csv, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fast-lio-sam-qn_amd"))
from qn_amd.build import csrc_sha1
from collections import defaultdict
FAMILY = [("KnnHistK<false, 32", "knn_select", 2), ("TickK<512, 4, 0,", "gn_tick_fused", 1), ("TickK<512, 4, 1,", "closing_pass", 1), ("NnLaneK<0>", "nn_search", 1),
          ("NnSearchK<0, true,", "nn_fallback", 1), ("AccumulateK", "accumulate", 1), ("CovFromIdxK", "cov_from_idx", 2), ("PackBBoxK", "grid_pack", 2), ("ScatterK", "grid_scatter", 2)]
def per_kernel(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        tot[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt
d, lanes = sys.argv[1], int(sys.argv[2])
fetch, nf = per_kernel(os.path.join(d, "pmc_FETCH_SIZE.csv")); write, _ = per_kernel(os.path.join(d, "pmc_WRITE_SIZE.csv"))
out = {}
for pat, fam, per_lane in FAMILY:
    ks = [k for k in fetch if pat in k and "k_lanes" in k]
    if not ks:
        continue
    k = ks[0]; f, w = fetch[k], write.get(k, 0.0); entries = lanes * per_lane
    out[fam] = {"kernel": k.split("(")[0], "launches_sampled": nf[k], "entries_per_launch": entries, "FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
                "hbm_bytes_per_launch": int((2.0 * f + w) * 1024), "hbm_bytes_per_registration_launch": int((2.0 * f + w) * 1024 / entries),
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KB; FETCH_SIZE x2 (gfx950 under-count, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated; working set is MALL resident"}
print(json.dumps({"batched": out, "_meta": {"tag": os.path.basename(os.path.normpath(d)), "lanes": lanes, "csrc_sha1": csrc_sha1()}}, indent=1))
