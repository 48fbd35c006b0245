#!/bin/bash
# PMC pass over the Quatro stage (matrix-core matching): MFMA busy cycles vs kernel duration.  usage: tools/gpu_quatro_pmc.sh <tag>
TAG=${1:-qpmc}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_BUSY_CYCLES|SQ_BUSY_CU_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_MFMA|SQ_INSTS_VALU\b" | sort -u > $OUT/counters.txt; cat $OUT/counters.txt | tr '\n' ' '; echo
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  N=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/p_$N -o p -- python tools/gpu_quatro_stage.py > /dev/null 2> $OUT/pmc_$N.err; echo "pmc $C exit $?"
  find $OUT/p_$N -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$N.csv \;
  find $OUT/p_$N -name '*kernel_trace.csv' -exec cp {} $OUT/trace_$N.csv \;
  rm -rf $OUT/p_$N
done
python - $OUT <<'PY'
import csv, sys, os, collections, glob
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_feat_mm" in n:
            agg[(n.split("(")[0][-16:], r.get("Grid_Size", r.get("Grid_Size_X", "")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(os.path.basename(f), k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
