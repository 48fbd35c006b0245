#!/bin/bash
# Quatro parity tests only (golden fixtures, small cases, 30k / 100k stages against the oracle)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_quatro.py tests/test_gpu_quatro_fullsize.py -m gpu -q -x 2>&1 | tail -25
