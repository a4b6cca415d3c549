#!/bin/bash
# round 4, session 2: the step with the ICP's tile cull (tree) against the same sources built without it (ab_tmp/nocull.so), alternated
# alt library: bash tools/build_ab_lib.sh nocull nn_batched.hip -DISR_ICP_CULL=0
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for rep in 1 2 3; do
for lib in "" nocull; do
  echo "== ${lib:-tree}"
  ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so} timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-estimate-pose 2> gpurun_out/s2/icp_step.err | python tools/bench_brief.py | cut -c1-100,240-
done; done > gpurun_out/s2/icp_step_ab.txt 2>&1 || { cat gpurun_out/s2/icp_step_ab.txt; tail -5 gpurun_out/s2/icp_step.err; exit 1; }
cat gpurun_out/s2/icp_step_ab.txt
