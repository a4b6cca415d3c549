#!/bin/bash
# round 4, session 2: K1's waves at priority 3 (tree) against the default 0 (ab_tmp/noprio.so): the step, alternated, and the kernel alone
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for rep in 1 2 3; do
for lib in "" noprio; do
  echo "== ${lib:-tree}"
  ISR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/ab_tmp/$lib.so} timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-estimate-pose --no-parity-check 2> gpurun_out/s2/prio_ab.err | python tools/bench_brief.py | cut -c1-110
done; done > gpurun_out/s2/prio_ab.txt 2>&1 || { cat gpurun_out/s2/prio_ab.txt; tail -5 gpurun_out/s2/prio_ab.err; exit 1; }
cat gpurun_out/s2/prio_ab.txt
