#!/bin/bash
# K1's hot loop on v_mfma_f32_32x32x16_bf16 (tree) against the same loop issuing v_mfma_f32_16x16x32_bf16 (timing-only ablation
# ISR_ABL_MFMA16), alternated on one box: wall per 32-image launch and the shader clock held under the kernel, planted and random data
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
out=gpurun_out/r05/k1_mfma_shape_ab.txt
: > $out
for rep in 1 2 3; do
  for name in tree full16 nomax nomax16; do
    lib=""; [ "$name" != tree ] && lib="$GRAFT_REPO_ROOT/ab_tmp/$name.so"
    echo "== $name (rep $rep)" >> $out
    ISR_HIP_LIB=$lib python tools/time_corr.py 9830400 20000 64 2>&1 | grep "bf16-log2:" >> $out
  done
done
cat $out
