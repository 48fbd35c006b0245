"""Summarise the rocprofv3 --pmc passes of tools/gpu_round.sh into per-kernel-family HBM traffic per launch.

usage: python tools/pmc_summary.py gpurun_out/<tag>      (reads pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv; prints JSON)

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the two counters are collected in
SEPARATE passes (TCC slot limit), are reported in KB, and on gfx950 FETCH_SIZE under-counts by 2x (doubled here).
bench.py reads the committed copy (profiles/pmc_latest.json) for the `roofline.traffic` field of its JSON line.
"""
import csv, json, os, sys
from collections import defaultdict

FAMILY = [("k_knn_hist<false, 32>", "knn_select"), ("k_tick<512, 4, 0,", "gn_tick_fused"), ("k_tick<512, 4, 1,", "closing_pass"), ("k_align_persist<512, false>", "align_persist"), ("k_far(", "far_refresh"),
          ("k_nn_search<0, false,", "nn_search"), ("k_nn_search<0, true,", "nn_fallback"), ("k_accumulate", "accumulate"),
          ("k_solve<512>", "solve"), ("k_cov_from_idx", "cov_from_idx"), ("k_scatter", "grid_scatter"), ("k_fitness_partial", "fitness")]


def per_kernel(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main(d):
    fetch, nf = per_kernel(os.path.join(d, "pmc_FETCH_SIZE.csv"))
    write, _ = per_kernel(os.path.join(d, "pmc_WRITE_SIZE.csv"))
    out = {}
    for pat, fam in FAMILY:
        ks = [k for k in fetch if pat in k]
        if not ks:
            continue
        k = ks[0]
        f, w = fetch[k], write.get(k, 0.0)
        out[fam] = {"kernel": k.split("(")[0], "launches_sampled": nf[k], "FETCH_SIZE_KB_per_launch": round(f, 1),
                    "WRITE_SIZE_KB_per_launch": round(w, 1), "hbm_bytes_per_launch": int((2.0 * f + w) * 1024),
                    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KB; FETCH_SIZE x2 (gfx950 under-count, "
                            "MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated; working set is L2/MALL resident"}
    out["_meta"] = {"tag": os.path.basename(os.path.normpath(d))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
