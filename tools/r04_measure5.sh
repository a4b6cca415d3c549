#!/bin/bash
# round 4: K2 counters after the scalar-mask edit, configs[3] as written, RCCL at world size 1, the bench under rocprofv3
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
R=$GRAFT_REPO_ROOT; cd "$R"; mkdir -p gpurun_out/r04
timeout -k 10 600 bash tools/pmc_k2.sh gpurun_out/r04/pmc_k2_v2 > gpurun_out/r04/pmc_k2_v2.log 2>&1 || { tail -20 gpurun_out/r04/pmc_k2_v2.log; exit 1; }
tail -22 gpurun_out/r04/pmc_k2_v2.log
timeout -k 10 600 python bench.py --object revolution --keys 50000 --itr 4096 --confidence 1 --steps 4 --no-cpu-baseline --no-estimate-pose > gpurun_out/r04/bench_config4.json 2> gpurun_out/r04/bench_config4.err || { tail -20 gpurun_out/r04/bench_config4.err; exit 1; }
python - <<'PY'
import json; d=json.loads(open("gpurun_out/r04/bench_config4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","acceptance","final_chamfer")}); print(d["config"]["hypotheses_scored_mean"], d["stage_ms_per_step"])
PY
ISR_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 600 python bench.py --steps 6 --no-cpu-baseline --no-estimate-pose > gpurun_out/r04/bench_rccl1.json 2> gpurun_out/r04/bench_rccl1.err || { tail -20 gpurun_out/r04/bench_rccl1.err; exit 1; }
python - <<'PY'
import json; d=json.loads(open("gpurun_out/r04/bench_rccl1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","dist","acceptance")})
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r04/prof_bench" -o b -- python3 "$R/bench.py" --steps 16 --no-cpu-baseline --no-parity-check --no-estimate-pose > "$R/gpurun_out/r04/bench_under_rocprof.json" 2> /dev/null
cd "$R"; F=$(ls gpurun_out/r04/prof_bench/*/b_kernel_stats.csv gpurun_out/r04/prof_bench/b_kernel_stats.csv 2>/dev/null | head -1); cp "$F" gpurun_out/r04/bench_kernel_stats.csv; python tools/kstats.py gpurun_out/r04/bench_kernel_stats.csv 14
