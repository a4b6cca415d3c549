#!/bin/bash
# round 4, session 2: the screened K1 (plain and plane routes): corr tests, every route alone (tree against ab_tmp/noscreen.so,
# alternated), the step with either library
# alt library: bash tools/build_ab_lib.sh noscreen corr_argmax.hip -DISR_K1_SCREEN=0
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_config4.py tests/test_gpu_ref_golden.py tests/test_gpu_estimate_pose.py tests/test_gpu_refine_pose.py -x -q -m gpu > gpurun_out/s2/screen_tests.txt 2>&1 || { tail -40 gpurun_out/s2/screen_tests.txt; exit 1; }
tail -3 gpurun_out/s2/screen_tests.txt
for rep in 1 2; do
for lib in "" noscreen; do
  echo "== ${lib:-tree}"
  export ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so}
  for shape in "9830400 20000 16" "9830400 20000 32" "1048576 200000 128"; do
    timeout -k 10 200 python tools/time_corr.py $shape 2>&1 | grep -E "planted bf16-log2:|random bf16-log2:"
  done
  timeout -k 10 200 python tools/time_corr_f32.py 307200 20000 64 2>&1 | grep -E "^f32 exact"
  timeout -k 10 200 python tools/time_corr_f32.py 307200 20000 128 0 2>&1 | grep -E "^f32 exact"
  timeout -k 10 200 python tools/time_corr_f32.py 280960 80000 12 2>&1 | grep -E "^f32 exact"
done; done > gpurun_out/s2/screen_routes_ab.txt 2>&1
cat gpurun_out/s2/screen_routes_ab.txt
for rep in 1 2; do
for lib in "" noscreen; do
  echo "== ${lib:-tree}"
  ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so} timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-estimate-pose 2> gpurun_out/s2/step_ab.err | python tools/bench_brief.py
done; done > gpurun_out/s2/step_ab.txt 2>&1 || { cat gpurun_out/s2/step_ab.txt; tail -5 gpurun_out/s2/step_ab.err; exit 1; }
cat gpurun_out/s2/step_ab.txt
