#!/bin/bash
# rocprofv3 kernel trace of a short headline run: usage tools/gpu_trace.sh <tag> ['<json knobs>'] [extra bench args]
TAG=${1:-tr}; KN="${2:-{\}}"; shift; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
QN_DEBUG_KNOBS="$KN" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python bench.py --no-cpu-baseline --no-quatro --no-extras --steps 20 --warmup 3 $@ > $OUT/bench_prof.json 2> $OUT/prof.err; echo "rocprof exit $?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec sh -c 'python tools/trace_summary.py {} > '$OUT'/last_registration_trace.txt 2>&1' \;
rm -rf $OUT/prof
head -8 $OUT/kernel_stats.csv | cut -c1-160
