#!/bin/bash
# round 4, session 2: per-wave tile cull of the ICP searches (tree) against the same sources built without it (ab_tmp/nocull.so)
# alt library: bash tools/build_ab_lib.sh nocull nn_batched.hip -DISR_ICP_CULL=0
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 800 python -m pytest tests/test_gpu_nn.py tests/test_gpu_registration.py tests/test_gpu_bench_size_parity.py tests/test_gpu_sequence.py tests/test_gpu_golden.py tests/test_gpu_ref_golden.py -x -q -m gpu > gpurun_out/s2/icp_tests.txt 2>&1 || { tail -40 gpurun_out/s2/icp_tests.txt; exit 1; }
tail -2 gpurun_out/s2/icp_tests.txt
for lib in "" nocull ""; do
  echo "== ${lib:-tree}"
  ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so} timeout -k 10 300 python tools/check_icp_cull.py 2>&1 | grep -E "^N="
done > gpurun_out/s2/icp_cull_check.txt 2>&1
cat gpurun_out/s2/icp_cull_check.txt
