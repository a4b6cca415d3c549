#!/bin/bash
# round 4, second GPU pass: the three-plane f32 route (tests + timings), K2 after the scalar-mask edit, 4-rank gloo rehearsal
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_corr.py tests/test_gpu_ransac.py tests/test_gpu_config4.py -x -q > gpurun_out/r04/t_corr.log 2>&1 || { tail -40 gpurun_out/r04/t_corr.log; exit 1; }
tail -2 gpurun_out/r04/t_corr.log
for shape in "307200 20000 64" "307200 20000 32" "280960 80000 12" "307200 50000 64"; do
  timeout -k 10 300 python tools/time_corr_f32.py $shape >> gpurun_out/r04/k1_f32_routes.txt 2>&1 || { tail -20 gpurun_out/r04/k1_f32_routes.txt; exit 1; }
done
cat gpurun_out/r04/k1_f32_routes.txt
timeout -k 10 300 python tools/time_ransac.py --chain > gpurun_out/r04/k2_alone_v2.txt 2>&1 || { tail -20 gpurun_out/r04/k2_alone_v2.txt; exit 1; }
cat gpurun_out/r04/k2_alone_v2.txt
ISR_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 4 --images 64 --steps 3 --no-estimate-pose > gpurun_out/r04/bench_gloo_4ranks.json 2> gpurun_out/r04/bench_gloo_4ranks.err || { tail -30 gpurun_out/r04/bench_gloo_4ranks.err; exit 1; }
python - <<'PY'
import json; d=json.loads(open("gpurun_out/r04/bench_gloo_4ranks.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","n_gpus","ms_per_step","dist","acceptance","per_rank_ms_per_step")}); print(d["parity_check"]); print(d["last_step"])
PY
