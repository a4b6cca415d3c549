#!/bin/bash
# round 4: the three-plane f32 route after a kernel edit — parity tests, then the routes' timings
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04; rm -f gpurun_out/r04/k1_f32_routes.txt
timeout -k 10 900 python -m pytest tests/test_gpu_corr.py -x -q > gpurun_out/r04/t_corr.log 2>&1 || { tail -40 gpurun_out/r04/t_corr.log; exit 1; }
tail -2 gpurun_out/r04/t_corr.log
for shape in "307200 20000 64" "307200 20000 32" "280960 80000 12" "307200 50000 64" "50176 80000 12"; do
  timeout -k 10 300 python tools/time_corr_f32.py $shape >> gpurun_out/r04/k1_f32_routes.txt 2>&1 || { tail -20 gpurun_out/r04/k1_f32_routes.txt; exit 1; }
done
grep -v amdgpu.ids gpurun_out/r04/k1_f32_routes.txt
