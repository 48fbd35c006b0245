#!/bin/bash
# SQ counter passes over the BATCHED path (tools/gpu_batch_sweep.py, one context x 8 lanes alone on the GPU).  usage: tools/gpu_sq_batch.sh <tag>
# Counters only with --kernel-trace (never with sys/hip/hsa traces: gpurun refuses that combination).  -> gpurun_out/<tag>/valu_budget.{txt,json}
TAG=${1:-sqb}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
STEPS=16; LANES=8
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $OUT/p$i -o p -- python tools/gpu_batch_sweep.py $STEPS 1x$LANES ${QN_SWEEP_SHIFT:--} > /dev/null 2> $OUT/p$i.err; echo "pass $i exit $?"
  find $OUT/p$i -name '*counter_collection.csv' -exec cp {} $OUT/sq_pass$i.csv \;
  rm -rf $OUT/p$i
done
python tools/valu_budget_batch.py $OUT $((2 * LANES + 3 * STEPS)) $TAG | tee $OUT/valu_budget.txt
