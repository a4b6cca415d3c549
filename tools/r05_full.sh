#!/bin/bash
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests.txt 2>&1 || true
tail -15 gpurun_out/r05/gpu_tests.txt
timeout -k 10 600 python bench.py --steps 10 > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err || true
tail -3 gpurun_out/r05/bench_n1.err
python3 - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05/bench_n1.json').read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "K1 frac", d["roofline"]["frac"], "ms/launch", d["roofline"]["ms_per_launch"], "alone", d["roofline"]["alone"])
    print("f32_step", d.get("f32_step"))
    print("f32_exact", {k: v for k, v in d["roofline"]["f32_exact"].items() if not isinstance(v, dict)})
    print("parity", d.get("parity_check"))
except Exception as e:
    print("bench parse failed", e)
PY
