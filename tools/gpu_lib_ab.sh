#!/bin/bash
# usage: tools/gpu_lib_ab.sh <steps> <cfgs> <libA.so> <libB.so> ...   -- developer A/B of library builds on ONE box (the boxes differ by a few %): each build is copied over the
# in-tree library of the box's scratch copy and the batch sweep runs, the whole list twice (A B A B)
STEPS=$1; CFG=$2; shift 2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/lib_ab
for rep in 1 2; do
  for L in "$@"; do
    cp $L fast-lio-sam-qn_amd/libqn_engine.so
    echo "== $L (round $rep)" | tee -a gpurun_out/lib_ab/ab.log
    timeout 200 python tools/gpu_batch_sweep.py $STEPS $CFG 2>&1 | grep -v "^{\|amdgpu.ids" | tee -a gpurun_out/lib_ab/ab.log
  done
done
