#!/bin/bash
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
python tools/mfma_scale_probe.py > gpurun_out/r05/scale_probe.txt 2>&1 || true
tail -30 gpurun_out/r05/scale_probe.txt
timeout -k 10 500 python tools/r05_skip_exp.py > gpurun_out/r05/skip_exp.txt 2>&1
cat gpurun_out/r05/skip_exp.txt
