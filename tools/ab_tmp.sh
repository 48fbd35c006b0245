cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6d
python -m pytest tests -m gpu -x -q > gpurun_out/r6d/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed" gpurun_out/r6d/pytest_gpu.log | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6d/kt -o kt -- python tools/gpu_batch_sweep.py 64 1x8 24 > gpurun_out/r6d/kt.log 2>&1
find gpurun_out/r6d/kt -name '*kernel_stats.csv' -exec cp {} gpurun_out/r6d/kernel_stats_24.csv \; ; rm -rf gpurun_out/r6d/kt
python tools/gpu_knob_sweep.py "{\"cfgs\": [\"3x8\"], \"knobs\": [{}, {}], \"lone_knobs\": [], \"steps\": 200, \"shift\": 24}" 2>&1 | grep -v amdgpu.ids | tail -2
python tools/gpu_knob_sweep.py "{\"cfgs\": [\"3x8\"], \"knobs\": [{}], \"lone_knobs\": [], \"steps\": 200}" 2>&1 | grep -v amdgpu.ids | tail -1
