#!/bin/bash
# run on the GPU box: bash tools/ablate_corr.sh [extra -D flags...]
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -w $*"
for v in "" "-DISR_ABL_NOUPD" "-DISR_ABL_NOMAX" "-DISR_ABL_NOEXP" "-DISR_ABL_NOEXP -DISR_ABL_NOMAX"; do
  hipcc $F $v tools/ablate_corr.hip -o /tmp/abl 2>/dev/null && printf "%-40s " "[$* $v]" && /tmp/abl
done
