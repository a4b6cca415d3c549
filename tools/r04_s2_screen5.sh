#!/bin/bash
# round 4, session 2: tile sum as four interleaved partial sums (tree) against one chain (ab_tmp/chain1.so), alternated
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_corr.py -x -q -m gpu > gpurun_out/s2/chain_tests.txt 2>&1 || { tail -40 gpurun_out/s2/chain_tests.txt; exit 1; }
tail -2 gpurun_out/s2/chain_tests.txt
for rep in 1 2; do
for lib in "" chain1; do
  echo "== ${lib:-tree}"
  export ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so}
  for shape in "9830400 20000 64" "9830400 20000 16" "1048576 200000 128"; do
    timeout -k 10 200 python tools/time_corr.py $shape 2>&1 | grep -E "planted bf16-log2:|random bf16-log2:"
  done
  timeout -k 10 200 python tools/time_corr_f32.py 307200 20000 32 0 2>&1 | grep -E "^f32 exact"
  timeout -k 10 200 python tools/time_corr_f32.py 307200 20000 128 0 2>&1 | grep -E "^f32 exact"
  timeout -k 10 200 python tools/time_corr_f32.py 280960 80000 12 0 2>&1 | grep -E "^f32 exact"
  timeout -k 10 200 python tools/time_corr_f32.py 280960 80000 12 2 2>&1 | grep -E "^f32 exact"
done; done > gpurun_out/s2/chain_ab.txt 2>&1
cat gpurun_out/s2/chain_ab.txt
