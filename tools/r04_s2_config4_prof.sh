#!/bin/bash
# round 4, session 2: kernel table of configs[3] as written (where do K1's 82 ms per launch go on near-tie-rich keys?)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2/prof4 -- python3 bench.py --object revolution --keys 50000 --itr 4096 --confidence 1 --steps 3 --no-cpu-baseline --no-parity-check --no-estimate-pose > gpurun_out/s2/bench_config4_rocprof.json 2> gpurun_out/s2/bench_config4_rocprof.err || { tail -20 gpurun_out/s2/bench_config4_rocprof.err; exit 1; }
f=$(ls gpurun_out/s2/prof4/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/s2/config4_kernel_stats.csv
python tools/kstats.py gpurun_out/s2/config4_kernel_stats.csv 14
rm -rf gpurun_out/s2/prof4
