#!/bin/bash
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout -k 10 600 python tools/r05_sparse_exp.py > gpurun_out/r05/sparse_exp.txt 2>&1 || true
cat gpurun_out/r05/sparse_exp.txt
timeout -k 10 300 python -m pytest tests/test_gpu_ransac.py -x -q > gpurun_out/r05/ransac_tests.txt 2>&1 || true
tail -3 gpurun_out/r05/ransac_tests.txt
