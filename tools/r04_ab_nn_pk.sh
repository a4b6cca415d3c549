#!/bin/bash
# A/B: K3's filter loop with v_pk_fma_f32 (tree) against scalar fmas (ab_tmp/nn_nopk.so), alternated on one box
set -eo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r04_nn_pk_ab.txt; : > $out
for rep in 1 2; do
  for lib in "" "$GRAFT_REPO_ROOT/ab_tmp/nn_nopk.so"; do
    echo "== ${lib:-tree (packed)}" >> $out
    ISR_HIP_LIB=$lib timeout -k 10 300 python tools/time_nn.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
