#!/bin/bash
# round 4, session 2: select_top in six launches — its tests (and every test that runs it), then the bench's stage figure
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests/test_gpu_select.py tests/test_gpu_ref_golden.py tests/test_gpu_sequence.py tests/test_gpu_golden.py tests/test_gpu_registration.py tests/test_gpu_prep.py tests/test_gpu_config4.py -x -q -m gpu > gpurun_out/s2/select_tests.txt 2>&1 || { tail -40 gpurun_out/s2/select_tests.txt; exit 1; }
tail -2 gpurun_out/s2/select_tests.txt
timeout -k 10 300 python bench.py --steps 12 --no-cpu-baseline --no-estimate-pose > gpurun_out/s2/bench_select.json 2> gpurun_out/s2/bench_select.err || { tail -5 gpurun_out/s2/bench_select.err; exit 1; }
python tools/bench_brief.py < gpurun_out/s2/bench_select.json
python -c "
import json; d=json.loads(open('gpurun_out/s2/bench_select.json').read().strip().splitlines()[-1]); print(d['stage_ms_per_step'])"
