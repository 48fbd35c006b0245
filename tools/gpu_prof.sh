#!/bin/bash
# rocprofv3 kernel stats + launch-order trace of the last registration (single stream).  usage: tools/gpu_prof.sh <tag>
TAG=${1:-prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python bench.py --no-cpu-baseline --no-quatro --steps 20 --warmup 3 --in-flight 1 > $OUT/bench_prof.json 2> $OUT/prof.err; echo "rocprof exit $?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec sh -c 'python tools/trace_summary.py {} > '$OUT'/last_registration_trace.txt 2>&1' \;
rm -rf $OUT/prof
cut -c1-160 $OUT/kernel_stats.csv | head -14; tail -70 $OUT/last_registration_trace.txt
