#!/bin/bash
# round 4, session 2: Gauss-Newton solve in a launch of its own (tree) against the last-workgroup solve behind __threadfence (ab_tmp/gnfused.so)
# alt library: bash tools/build_ab_lib.sh gnfused ransac.hip -DISR_GN_SPLIT_SOLVE=0
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_ransac.py tests/test_gpu_config4.py tests/test_gpu_sequence.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/s2/gn_tests.txt 2>&1 || { tail -40 gpurun_out/s2/gn_tests.txt; exit 1; }
tail -2 gpurun_out/s2/gn_tests.txt
for rep in 1 2 3; do
for lib in "" gnfused; do
  echo "== ${lib:-tree}"
  ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so} timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-estimate-pose --no-parity-check 2> gpurun_out/s2/gn_ab.err | python tools/bench_brief.py | cut -c1-110
done; done > gpurun_out/s2/gn_ab.txt 2>&1 || { cat gpurun_out/s2/gn_ab.txt; tail -5 gpurun_out/s2/gn_ab.err; exit 1; }
cat gpurun_out/s2/gn_ab.txt
