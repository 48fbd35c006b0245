cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r6c_final; mkdir -p $OUT
bash tools/gpu_round4.sh r6c_final pmc sq > $OUT/round.log 2>&1; grep -E "exit|passed" $OUT/round.log | head
echo "== tools/gpu_parity_sweep.py 40 34 --lanes 4 --hard" >> $OUT/parity_sweeps_extra.txt; timeout 600 python tools/gpu_parity_sweep.py 40 34 --lanes 4 --hard 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/parity_sweeps_extra.txt
