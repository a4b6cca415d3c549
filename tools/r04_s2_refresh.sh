#!/bin/bash
# round 4, session 2: new K1 tests, the multi-rank command (2 and 4 gloo ranks sharing the GPU), configs[3] as written and the 50 000-key variant at the final kernels
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_corr.py -x -q -m gpu -k "screened" > gpurun_out/s2/screen_f32_tests.txt 2>&1 || { tail -40 gpurun_out/s2/screen_f32_tests.txt; exit 1; }
tail -2 gpurun_out/s2/screen_f32_tests.txt
ISR_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --images 64 --steps 3 --no-cpu-baseline --no-estimate-pose > gpurun_out/s2/bench_gloo4.json 2> gpurun_out/s2/bench_gloo4.err || { tail -20 gpurun_out/s2/bench_gloo4.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/s2/bench_gloo4.json').read().strip().splitlines()[-1]); p=d['parity_check']; print('gloo4', round(d['value'],1), d['dist'], p.get('pick_idx_equal'), p.get('pairs_checked'), p.get('k1_in_step_equals_alone'), len(d['per_rank_ms_per_step']['all']), d['last_step'].get('icp_rank'))"
timeout -k 10 600 python bench.py --object revolution --keys 50000 --itr 4096 --confidence 1 --steps 8 --no-cpu-baseline --no-estimate-pose > gpurun_out/s2/bench_config4.json 2> gpurun_out/s2/bench_config4.err || { tail -20 gpurun_out/s2/bench_config4.err; exit 1; }
python tools/bench_brief.py < gpurun_out/s2/bench_config4.json | cut -c1-120
python -c "
import json; d=json.loads(open('gpurun_out/s2/bench_config4.json').read().strip().splitlines()[-1]); print(d['acceptance'], d['config'].get('hypotheses_scored_mean'))"
timeout -k 10 600 python bench.py --keys 50000 --steps 8 --no-cpu-baseline --no-estimate-pose > gpurun_out/s2/bench_50k.json 2> gpurun_out/s2/bench_50k.err || { tail -20 gpurun_out/s2/bench_50k.err; exit 1; }
python tools/bench_brief.py < gpurun_out/s2/bench_50k.json | cut -c1-120
