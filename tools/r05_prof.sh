#!/bin/bash
# usage: bash tools/r05_prof.sh <name> <python script and args>  -> gpurun_out/r05/<name>_stats.csv
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; shift
mkdir -p gpurun_out/r05
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/prof_$name -- python3 "$@" > gpurun_out/r05/prof_$name.log 2>&1
f=$(ls gpurun_out/r05/prof_$name/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/r05/${name}_stats.csv
head -12 gpurun_out/r05/${name}_stats.csv | cut -c1-220
