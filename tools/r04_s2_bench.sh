#!/bin/bash
# round 4, session 2: the bench line (default flags, 20 steps) and the rocprofv3 kernel table of the same command (16 steps, no CPU legs)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 700 python bench.py --steps 20 > gpurun_out/s2/bench_n1.json 2> gpurun_out/s2/bench_n1.err || { tail -20 gpurun_out/s2/bench_n1.err; exit 1; }
python tools/bench_brief.py < gpurun_out/s2/bench_n1.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2/prof -- python3 bench.py --steps 16 --no-cpu-baseline --no-parity-check --no-estimate-pose > gpurun_out/s2/bench_under_rocprof.json 2> gpurun_out/s2/bench_under_rocprof.err || { tail -20 gpurun_out/s2/bench_under_rocprof.err; exit 1; }
f=$(ls gpurun_out/s2/prof/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/s2/kernel_stats.csv
python tools/kstats.py gpurun_out/s2/kernel_stats.csv 12
python tools/bench_brief.py < gpurun_out/s2/bench_under_rocprof.json
rm -rf gpurun_out/s2/prof
