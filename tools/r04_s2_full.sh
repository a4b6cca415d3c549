#!/bin/bash
# round 4, session 2: whole GPU suite + smoke + a bench line at HEAD
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s2/tests.txt 2>&1 || { tail -40 gpurun_out/s2/tests.txt; exit 1; }
tail -3 gpurun_out/s2/tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s2/smoke.txt 2>&1 || { tail -20 gpurun_out/s2/smoke.txt; exit 1; }
tail -1 gpurun_out/s2/smoke.txt
timeout -k 10 500 python bench.py > gpurun_out/s2/bench.json 2> gpurun_out/s2/bench.err || { tail -20 gpurun_out/s2/bench.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/s2/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms_per_step'])"
