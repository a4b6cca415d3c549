#!/bin/bash
# round 4, session 2: batched brute-force NN with one packed atomic min per query (tree) against per-split partial arrays (ab_tmp/nopacked.so);
# tests, timing alternated, then the HBM-traffic counters of K1 and the NN shape again (tools/pmc_all.sh -> profiles/*_hbm_traffic.json)
# alt library: bash tools/build_ab_lib.sh nopacked nn_batched.hip -DISR_NN_PACKED_BATCH=0
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 800 python -m pytest tests/test_gpu_nn.py tests/test_gpu_registration.py tests/test_gpu_bench_size_parity.py tests/test_gpu_sequence.py tests/test_gpu_golden.py tests/test_gpu_ref_golden.py tests/test_gpu_config3_sharded.py -x -q -m gpu > gpurun_out/s2/packed_tests.txt 2>&1 || { tail -40 gpurun_out/s2/packed_tests.txt; exit 1; }
tail -2 gpurun_out/s2/packed_tests.txt
for rep in 1 2 3; do
for lib in "" nopacked; do
  echo "== ${lib:-tree}"
  ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so} timeout -k 10 200 python tools/time_nn_brute.py 2>&1 | grep -E "brute-force"
done; done > gpurun_out/s2/packed_ab.txt 2>&1
cat gpurun_out/s2/packed_ab.txt
timeout -k 10 900 bash tools/pmc_all.sh gpurun_out/s2/pmc_all > gpurun_out/s2/pmc_all.log 2>&1 || { tail -20 gpurun_out/s2/pmc_all.log; exit 1; }
tail -5 gpurun_out/s2/pmc_all.log
cp profiles/k1_hbm_traffic.json profiles/nn_hbm_traffic.json gpurun_out/s2/
