#!/bin/bash
# round 4, session 2: screen ablations (timing only) and the SQ counters of the screened kernel
# alt libraries: bash tools/build_ab_lib.sh noscreen corr_argmax.hip -DISR_K1_SCREEN=0; abl_never: -DISR_ABL_SCREEN=1; abl_always: -DISR_ABL_SCREEN=2; abl_nomax: -DISR_ABL_DNOMAX
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for rep in 1 2; do
for lib in "" noscreen abl_never abl_always abl_nomax; do
  echo "== ${lib:-tree}"
  ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so} timeout -k 10 120 python tools/time_corr.py 9830400 20000 64 2>&1 | grep -E "bf16-log2:"
done; done > gpurun_out/s2/screen_abl.txt 2>&1
cat gpurun_out/s2/screen_abl.txt
PMC_N=20000 timeout -k 10 600 bash tools/pmc_corr.sh gpurun_out/s2/pmc_screen 4915200 20000 64 > gpurun_out/s2/pmc_screen.log 2>&1 || { tail -20 gpurun_out/s2/pmc_screen.log; exit 1; }
cat gpurun_out/s2/pmc_screen/summary.txt | grep -A40 "direct_kernel<log2>"
