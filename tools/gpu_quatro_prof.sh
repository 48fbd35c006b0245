#!/bin/bash
# rocprofv3 kernel stats of the Quatro coarse stage (tools/gpu_quatro_stage.py): usage tools/gpu_quatro_prof.sh <tag>
TAG=${1:-qprof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python tools/gpu_quatro_stage.py > $OUT/stage.log 2> $OUT/prof.err; echo "rocprof exit $?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/prof
tail -6 $OUT/stage.log
python - $OUT/kernel_trace.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# the last mfma-path align of the 100k run = find last k_feat_mm<1> pair; print per-kernel durations by grid size for feat kernels
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_feat" in n or "k_fpfh" in n or "k_spfh" in n or "k_normals" in n:
        key = (n.split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"])
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print("%-62s grid %8s x %4s  n %3d  median %9.1f us" % (k[0], k[1], k[2], len(v), sorted(v)[len(v) // 2]))
PY
