#!/bin/bash
# One GPU-box round: parity tests, bench line, rocprofv3 kernel stats (+ optional PMC passes).
# usage: tools/gpu_round.sh <tag> [pmc]     outputs under gpurun_out/<tag>/
TAG=${1:-run}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log | cut -c1-400
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json | cut -c1-1500
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python bench.py --no-cpu-baseline --no-quatro --no-extras --steps 20 --warmup 3 > $OUT/bench_prof.json 2> $OUT/prof.err; echo "rocprof exit $?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec sh -c 'python tools/trace_summary.py {} > '$OUT'/last_registration_trace.txt 2>&1' \;
rm -rf $OUT/prof
head -12 $OUT/kernel_stats.csv | cut -c1-200
if [ "$2" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$C -o p -- python bench.py --no-cpu-baseline --no-quatro --no-extras --steps 4 --warmup 1 --in-flight 1 > /dev/null 2> $OUT/pmc_$C.err; echo "pmc $C exit $?"
    find $OUT/pmc_$C -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$C.csv \;
    rm -rf $OUT/pmc_$C
  done
  python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json 2>&1; head -c 1500 $OUT/pmc_summary.json
fi
