#!/bin/bash
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout -k 10 600 python tools/r05_sparse_exp.py > gpurun_out/r05/sparse_exp.txt 2>&1 || true
cat gpurun_out/r05/sparse_exp.txt
bash tools/r05_prof.sh sparse2 tools/r05_sparse_prof.py 5 > /dev/null 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r05/sparse2_stats.csv')):
    if 'corr' in r['Name']:
        print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6)
PY
