#!/bin/bash
# round 4, session 2: plane-route screen by plane count (tree: SP = 1, 8; ab_tmp/planes2.so: SP = 2 as well), the stress tools,
# then the whole GPU suite on the tree's library
# alt libraries: bash tools/build_ab_lib.sh planes2 corr_argmax.hip -DISR_K1_SCREEN_PLANES=0x106 (the shipped value since; the tree then had 0x102); noscreen: -DISR_K1_SCREEN=0
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for rep in 1 2; do
for lib in "" planes2 noscreen; do
  echo "== ${lib:-tree}"
  export ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so}
  timeout -k 10 200 python tools/time_corr_f32.py 307200 20000 32 2>&1 | grep -E "^f32 exact"
  timeout -k 10 200 python tools/time_corr_f32.py 307200 20000 64 0 2>&1 | grep -E "^f32 exact"
done; done > gpurun_out/s2/screen_planes_ab.txt 2>&1
unset ISR_HIP_LIB
cat gpurun_out/s2/screen_planes_ab.txt
timeout -k 10 400 python tools/stress_corr.py > gpurun_out/s2/stress_corr.txt 2>&1 || { tail -20 gpurun_out/s2/stress_corr.txt; exit 1; }
tail -4 gpurun_out/s2/stress_corr.txt
timeout -k 10 400 python tools/stress_corr_f32.py > gpurun_out/s2/stress_corr_f32.txt 2>&1 || { tail -20 gpurun_out/s2/stress_corr_f32.txt; exit 1; }
tail -4 gpurun_out/s2/stress_corr_f32.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2/tests_screen.txt 2>&1 || { tail -40 gpurun_out/s2/tests_screen.txt; exit 1; }
tail -3 gpurun_out/s2/tests_screen.txt
