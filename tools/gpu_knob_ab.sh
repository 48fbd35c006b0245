#!/bin/bash
# usage: tools/gpu_knob_ab.sh <tag> <steps> <cfgs> '<knobs json>' ['<knobs json>' ...]   -- registrations/s of the batch sweep under each knob set
TAG=$1; STEPS=$2; CFG=$3; shift 3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
for K in "$@"; do
  echo "== knobs $K" | tee -a gpurun_out/$TAG/ab.log
  QN_DEBUG_KNOBS="$K" timeout 200 python tools/gpu_batch_sweep.py $STEPS $CFG 2>&1 | grep -v "^{\|amdgpu.ids\|QN_DEBUG_KNOBS applied" | tee -a gpurun_out/$TAG/ab.log
done
