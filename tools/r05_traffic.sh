#!/bin/bash
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_screened.py -x -q > gpurun_out/r05/corr_tests.txt 2>&1 || true
tail -4 gpurun_out/r05/corr_tests.txt
bash tools/pmc_all.sh gpurun_out/r05/pmc_traffic > gpurun_out/r05/pmc_traffic.log 2>&1 || true
tail -5 gpurun_out/r05/pmc_traffic.log
cp profiles/k1_hbm_traffic.json gpurun_out/r05/k1_hbm_traffic.json
cp profiles/nn_hbm_traffic.json gpurun_out/r05/nn_hbm_traffic.json
python3 - <<'PY'
import json
d = json.load(open('profiles/k1_hbm_traffic.json'))
for k, v in d['per_launch'].items():
    print(k, v['hbm_bytes'], v['algorithmic_bytes'], v['hbm_bytes'] / v['algorithmic_bytes'], v['by_kernel']['corr_bf16_direct_kernel'])
PY
