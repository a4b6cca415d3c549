#!/bin/bash
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
rm -rf gpurun_out/r05/prof_bench
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/prof_bench -- python3 bench.py --steps 12 --no-cpu-baseline --no-parity-check --no-estimate-pose --no-f32-step --no-screened-step > gpurun_out/r05/bench_under_rocprof.json 2> gpurun_out/r05/bench_under_rocprof.err || { tail -20 gpurun_out/r05/bench_under_rocprof.err; exit 1; }
f=$(ls gpurun_out/r05/prof_bench/*/*kernel_trace.csv | head -1)
python3 tools/side_budget.py "$f" 12 | tee gpurun_out/r05/side_work_budget.txt
cp $(ls gpurun_out/r05/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/r05/bench_kernel_stats.csv
head -1 "$f" > gpurun_out/r05/trace_header.txt
rm -rf gpurun_out/r05/prof_bench
