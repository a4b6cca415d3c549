#!/bin/bash
# A/B on one box: the tree's libisr_hip.so against ab_tmp/<name>.so (a build of the same tree with one switch changed),
# alternated: bash tools/ab_lib.sh <name> <python tool and its arguments ...>
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; alt="$GRAFT_REPO_ROOT/ab_tmp/$1.so"; shift
for rep in 1 2 3; do
  for lib in "" "$alt"; do
    echo "== ${lib:-tree}"
    ISR_HIP_LIB=$lib python "$@" 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
