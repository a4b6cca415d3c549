#!/bin/bash
# A/B on one box: K1 (bf16, D = 64) with its key stages through registers (the tree's library) against buffer_load ... lds
# (ab_tmp/libisr_prev.so: corr_argmax.hip built with -DISR_DIRECT_DMA=1, the tree's other objects), alternated.
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for lib in "" "$GRAFT_REPO_ROOT/ab_tmp/libisr_prev.so"; do
    echo "== ${lib:-tree}"
    ISR_HIP_LIB=$lib python tools/time_corr.py 4915200 20000 64 2>&1 | grep "bf16-log2" | head -2
  done
done
ISR_HIP_LIB=$GRAFT_REPO_ROOT/ab_tmp/libisr_prev.so timeout -k 10 600 python -m pytest tests/test_gpu_corr.py -x -q 2>&1 | tail -2
