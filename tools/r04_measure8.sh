#!/bin/bash
# round 4: configs[4] K1 stress shape, the north star's 50 000-key variant of the bench
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout -k 10 600 python tools/time_corr.py 1048576 200000 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/k1_config5.txt || { tail gpurun_out/r04/k1_config5.txt; exit 1; }
cat gpurun_out/r04/k1_config5.txt
timeout -k 10 600 python bench.py --keys 50000 --steps 6 --no-cpu-baseline --no-estimate-pose > gpurun_out/r04/bench_50k.json 2> gpurun_out/r04/bench_50k.err || { tail -20 gpurun_out/r04/bench_50k.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r04/bench_50k.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['ms_per_launch'], r['alone'], d['parity_check'])"
