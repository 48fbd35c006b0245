mkdir -p gpurun_out/r6_a4
python tools/gpu_share8_segments.py '{}' 2>/dev/null | grep SEG > gpurun_out/r6_a4/seg.out
echo "== cap 3 (off)" >> gpurun_out/r6_a4/seg.out
python tools/gpu_share8_segments.py '{"unseeded_cap":3}' 2>/dev/null | grep SEG >> gpurun_out/r6_a4/seg.out
cat gpurun_out/r6_a4/seg.out
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_batch_oracle.py tests/test_gpu_persist.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r6_a4/pytest.log
