#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_corr.py -x -q > gpurun_out/r04/t_corr.log 2>&1 || { tail -40 gpurun_out/r04/t_corr.log; exit 1; }
tail -2 gpurun_out/r04/t_corr.log
for shape in "307200 20000 128" "307200 20000 100" "307200 20000 64"; do
  timeout -k 10 300 python tools/time_corr_f32.py $shape 2>&1 | grep -v amdgpu.ids
done
