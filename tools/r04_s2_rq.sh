#!/bin/bash
# round 4, session 2: the step after the cold-pass fix, and with one query per lane in the brute-force searches (thin waves beside K1)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for rep in 1 2; do
for v in "" "--tune nn_plan_rq=1"; do
  echo "== $v"
  timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-estimate-pose $v 2> gpurun_out/s2/rq.err | python tools/bench_brief.py | cut -c1-100,240-
done; done > gpurun_out/s2/rq_ab.txt 2>&1 || { cat gpurun_out/s2/rq_ab.txt; tail -5 gpurun_out/s2/rq.err; exit 1; }
cat gpurun_out/s2/rq_ab.txt
