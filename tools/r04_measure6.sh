#!/bin/bash
# round 4: whole GPU suite, smoke, the bench line (20 steps)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04/t_all.log 2>&1 || { tail -40 gpurun_out/r04/t_all.log; exit 1; }
tail -3 gpurun_out/r04/t_all.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.log 2>&1 || { tail -20 gpurun_out/r04/smoke.log; exit 1; }
tail -1 gpurun_out/r04/smoke.log
timeout -k 10 900 python bench.py --steps 20 > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err || { tail -20 gpurun_out/r04/bench_n1.err; exit 1; }
python - <<'PY'
import json; d=json.loads(open("gpurun_out/r04/bench_n1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step")}); r=d["roofline"]; print({k: r[k] for k in ("frac","ms_per_launch","clock_mhz_under_kernel","frac_at_held_clock")}, r["alone"]); print(r["f32_exact"]); print(d["parity_check"]); print(d["cpu_baseline"]["value"])
PY
