#!/bin/bash
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_ransac.py tests/test_gpu_config4.py tests/test_gpu_sequence.py tests/test_gpu_golden.py -x -q 2>&1 | tail -4
bash tools/r05_ab_bench.sh "tree||--no-screened-step" "gnold|gnold|--no-screened-step"
cp gpurun_out/r05/ab_bench.txt gpurun_out/r05/ab_bench_gn.txt
bash tools/r05_side.sh 2>&1 | tail -2
