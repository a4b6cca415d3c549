#!/bin/bash
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_corr_screened.py -x -q > gpurun_out/r05/screened_tests.txt 2>&1 || true
tail -30 gpurun_out/r05/screened_tests.txt
timeout -k 10 600 python tools/r05_sparse_exp.py > gpurun_out/r05/sparse_exp.txt 2>&1 || true
tail -9 gpurun_out/r05/sparse_exp.txt
bash tools/r05_abl.sh 2>&1 | tail -4
MODE=0 bash tools/r05_abl.sh 2>&1 | tail -3
