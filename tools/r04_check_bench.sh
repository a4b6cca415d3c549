#!/bin/bash
# quick check of bench.py after an edit: one rank (4 steps) and two gloo ranks sharing the GPU
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
timeout -k 10 600 python bench.py --steps 4 --no-cpu-baseline --no-estimate-pose > gpurun_out/r04/b1.json 2> gpurun_out/r04/b1.err || { tail -20 gpurun_out/r04/b1.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r04/b1.json').read().strip().splitlines()[-1]); print(d['value'], d['parity_check'])"
ISR_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --images 16 --steps 3 --no-cpu-baseline --no-estimate-pose > gpurun_out/r04/b2.json 2> gpurun_out/r04/b2.err || { tail -20 gpurun_out/r04/b2.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r04/b2.json').read().strip().splitlines()[-1]); print(d['value'], d['dist'], d['parity_check'].get('k1_in_step_equals_alone'), d['parity_check'].get('pick_idx_equal'))"
