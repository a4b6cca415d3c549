#!/bin/bash
# round 4, first GPU pass: configs[2] tests, K2 alone + PMC, bench lines (configs[1], configs[3]), 6-rank gloo rehearsal
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_config3_sharded.py -x -q > gpurun_out/r04/t_config3.log 2>&1 || { tail -30 gpurun_out/r04/t_config3.log; exit 1; }
tail -2 gpurun_out/r04/t_config3.log
timeout -k 10 300 python tools/time_ransac.py --chain > gpurun_out/r04/k2_alone.txt 2>&1 || { tail -20 gpurun_out/r04/k2_alone.txt; exit 1; }
cat gpurun_out/r04/k2_alone.txt
timeout -k 10 600 bash tools/pmc_k2.sh gpurun_out/r04/pmc_k2 > gpurun_out/r04/pmc_k2.log 2>&1 || { tail -20 gpurun_out/r04/pmc_k2.log; exit 1; }
tail -30 gpurun_out/r04/pmc_k2.log
timeout -k 10 600 python bench.py --steps 8 > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err || { tail -20 gpurun_out/r04/bench_n1.err; exit 1; }
python - <<'PY'
import json; d=json.loads(open("gpurun_out/r04/bench_n1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","dist","acceptance")}); print(d["roofline_ransac"]); print(d["parity_check"])
PY
timeout -k 10 600 python bench.py --object revolution --keys 50000 --itr 4096 --confidence 1 --steps 3 --no-cpu-baseline --no-estimate-pose > gpurun_out/r04/bench_config4.json 2> gpurun_out/r04/bench_config4.err || { tail -20 gpurun_out/r04/bench_config4.err; exit 1; }
python - <<'PY'
import json; d=json.loads(open("gpurun_out/r04/bench_config4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","acceptance","final_chamfer")}); print(d["config"]["hypotheses_scored_mean"], d["parity_check"])
PY
ISR_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 6 --images 64 --steps 3 --no-estimate-pose > gpurun_out/r04/bench_gloo_6ranks.json 2> gpurun_out/r04/bench_gloo_6ranks.err || { tail -30 gpurun_out/r04/bench_gloo_6ranks.err; exit 1; }
python - <<'PY'
import json; d=json.loads(open("gpurun_out/r04/bench_gloo_6ranks.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","n_gpus","ms_per_step","dist","acceptance","per_rank_ms_per_step")}); print(d["parity_check"]); print(d["last_step"])
PY
