#!/bin/bash
# SQ counter passes (instruction mix, wait cycles, LDS conflicts) for the hot kernels.  usage: tools/gpu_sq.sh <tag>
# Counters only with --kernel-trace (never with sys/hip/hsa traces: gpurun refuses that combination).
TAG=${1:-sq}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $SET -d $OUT/p$i -o p -- python bench.py --no-cpu-baseline --no-quatro --no-extras --steps 2 --warmup 1 --in-flight 1 > /dev/null 2> $OUT/p$i.err; echo "pass $i exit $?"
  find $OUT/p$i -name '*counter_collection.csv' -exec cp {} $OUT/sq_pass$i.csv \;
  rm -rf $OUT/p$i
done
python tools/sq_summary.py $OUT > $OUT/sq_counters.json; python - <<PY
import json
d=json.load(open("$OUT/sq_counters.json"))
for k in ("k_knn_hist<false, 32>","k_tick<512, 4, 0>","k_nn_search<0, false>","k_nn_search<0, true>","k_cov_from_idx"):
    for name,v in d.items():
        if k in name: print(name[:40], json.dumps(v))
PY
