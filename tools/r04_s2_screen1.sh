#!/bin/bash
# round 4, session 2: the sum screen of K1's maxima — corr tests, then tree (screen) against ab_tmp/noscreen.so, alternated
# alt library: bash tools/build_ab_lib.sh noscreen corr_argmax.hip -DISR_K1_SCREEN=0   (after the tree's build, on the build host)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_config4.py tests/test_gpu_ref_golden.py -x -q -m gpu > gpurun_out/s2/screen_tests.txt 2>&1 || { tail -40 gpurun_out/s2/screen_tests.txt; exit 1; }
tail -3 gpurun_out/s2/screen_tests.txt
timeout -k 10 500 bash tools/ab_lib.sh noscreen tools/time_corr.py 9830400 20000 64 > gpurun_out/s2/screen_ab.txt 2>&1
grep -E "^==|planted bf16-log2:|random bf16-log2:|0.6x" gpurun_out/s2/screen_ab.txt
