#!/bin/bash
# round 4, session 2: key ranges of the exact recheck by list length (tree) against always as many as the scratch holds (ab_tmp/rr_old.so)
# alt library: bash tools/build_ab_lib.sh rr_old corr_argmax.hip -DISR_K1_RECHECK_UNITS=4000000000000
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_config4.py -x -q -m gpu > gpurun_out/s2/recheck_tests.txt 2>&1 || { tail -40 gpurun_out/s2/recheck_tests.txt; exit 1; }
tail -2 gpurun_out/s2/recheck_tests.txt
for rep in 1 2; do
for lib in "" rr_old; do
  echo "== ${lib:-tree}"
  export ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so}
  timeout -k 10 300 python tools/time_corr_ties.py 2>&1 | grep -E "^revolution"
  timeout -k 10 200 python tools/time_corr.py 9830400 20000 64 2>&1 | grep -E "random bf16-log2:"
done; done > gpurun_out/s2/recheck_ab.txt 2>&1
unset ISR_HIP_LIB
cat gpurun_out/s2/recheck_ab.txt
timeout -k 10 400 python tools/stress_corr.py > gpurun_out/s2/stress_corr2.txt 2>&1 || { tail -20 gpurun_out/s2/stress_corr2.txt; exit 1; }
tail -1 gpurun_out/s2/stress_corr2.txt
