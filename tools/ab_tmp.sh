cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6c
python -m pytest tests -m gpu -x -q > gpurun_out/r6c/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r6c/pytest_gpu.log | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6c/kt -o kt -- python tools/gpu_batch_sweep.py 64 1x8 > gpurun_out/r6c/kt.log 2>&1
find gpurun_out/r6c/kt -name '*kernel_stats.csv' -exec cp {} gpurun_out/r6c/kernel_stats.csv \; ; rm -rf gpurun_out/r6c/kt
head -9 gpurun_out/r6c/kernel_stats.csv | cut -c1-60,100-200
bash tools/gpu_sq_batch.sh r6c/sq 2>&1 | grep -i "KnnHistK<false\|NnSearchK\|TickK<512, 4, 0\|one registration"
python tools/gpu_knob_sweep.py "{\"cfgs\": [\"3x8\"], \"knobs\": [{}, {}], \"lone_knobs\": [{}], \"steps\": 200}" 2>&1 | grep -v amdgpu.ids | tail -3
python tools/gpu_knob_sweep.py "{\"cfgs\": [\"3x8\"], \"knobs\": [{}], \"lone_knobs\": [], \"steps\": 200, \"shift\": 24}" 2>&1 | grep -v amdgpu.ids | tail -1
