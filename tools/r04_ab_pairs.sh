#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_corr.py -x -q 2>&1 | tail -2
bash tools/ab_lib.sh libisr_prev tools/time_corr.py 4915200 20000 64 2>&1 | grep "==\|planted bf16-log2:"
bash tools/ab_lib.sh libisr_prev tools/time_corr_f32.py 307200 20000 64 0 2>&1 | grep "==\|f32 exact"
bash tools/ab_lib.sh libisr_prev tools/time_corr_f32.py 280960 80000 12 0 2>&1 | grep "==\|f32 exact"
