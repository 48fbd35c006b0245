#!/bin/bash
# One GPU-box round on the batched path: parity tests, bench line, rocprofv3 kernel stats of the bench command (+ optional PMC / SQ passes over the batched sweep).
# usage: tools/gpu_round4.sh <tag> [pmc] [sq]     outputs under gpurun_out/<tag>/
TAG=${1:-run}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rx --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log | cut -c1-300
timeout 700 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; head -c 600 $OUT/bench.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-quatro > $OUT/bench_steps20.json 2> /dev/null; echo "bench(20) exit $?"; head -c 300 $OUT/bench_steps20.json; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python bench.py --no-cpu-baseline --no-quatro --no-extras --steps 48 --warmup 5 --repeats 0 > $OUT/bench_prof.json 2> $OUT/prof.err; echo "rocprof exit $?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/prof
head -8 $OUT/kernel_stats.csv | cut -c1-160
shift
for W in "$@"; do
  if [ "$W" = "pmc" ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$C -o p -- python tools/gpu_batch_sweep.py 16 1x8 > /dev/null 2> $OUT/pmc_$C.err; echo "pmc $C exit $?"
      find $OUT/pmc_$C -name '*counter_collection.csv' -exec cp {} $OUT/pmc_$C.csv \;
      rm -rf $OUT/pmc_$C
    done
    python tools/pmc_summary_batch.py $OUT 8 > $OUT/pmc_summary.json 2>&1; head -c 800 $OUT/pmc_summary.json; echo
  fi
  if [ "$W" = "sq" ]; then tools/gpu_sq_batch.sh $TAG/sq | tail -18; fi
done
