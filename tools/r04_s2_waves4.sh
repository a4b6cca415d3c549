#!/bin/bash
# round 4, session 2: plain-row K1 compiled for two waves per SIMD (ab_tmp/waves2.so) against three (tree)
# alt library: bash tools/build_ab_lib.sh waves2 corr_argmax.hip -DISR_K1_PLAIN_WAVES=2   (waves4: =4)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for rep in 1 2; do
for lib in "" waves2; do
  echo "== ${lib:-tree}"
  export ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so}
  for shape in "9830400 20000 64"; do
    timeout -k 10 200 python tools/time_corr.py $shape 2>&1 | grep -E "planted bf16-log2:|random bf16-log2:"
  done
done; done > gpurun_out/s2/waves2_ab.txt 2>&1
cat gpurun_out/s2/waves2_ab.txt
