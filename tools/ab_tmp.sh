cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6d
python -m pytest tests/test_gpu_gicp.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6d/kt -o kt -- python tools/gpu_batch_sweep.py 64 1x8 > gpurun_out/r6d/kt.log 2>&1
find gpurun_out/r6d/kt -name '*kernel_stats.csv' -exec cp {} gpurun_out/r6d/kernel_stats.csv \; ; rm -rf gpurun_out/r6d/kt
grep "NnLaneK\|NnSearchK" gpurun_out/r6d/kernel_stats.csv | cut -c20-60,100-190
python tools/gpu_knob_sweep.py "{\"cfgs\": [\"3x8\"], \"knobs\": [{}, {}], \"lone_knobs\": [], \"steps\": 200}" 2>&1 | grep -v amdgpu.ids | tail -2
