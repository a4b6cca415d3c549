#!/bin/bash
# round 4, session 2: what the side work costs the step at the screened K1 (bench.py variants on one box; not valid bench lines)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for v in "" "--ablate noverify" "--refine-iters 0" "--ablate noverify --refine-iters 0" "--ablate noverify --itr 32 --refine-iters 0" ""; do
  echo "== $v"
  timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline --no-estimate-pose --no-parity-check $v 2> gpurun_out/s2/abl.err | python tools/bench_brief.py || { tail -5 gpurun_out/s2/abl.err; }
done > gpurun_out/s2/step_ablations.txt 2>&1
cat gpurun_out/s2/step_ablations.txt
