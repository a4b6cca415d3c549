#!/bin/bash
# per-kernel times of the screened route for the tree's library and each ab_tmp/<name>.so given: bash tools/r05_abl.sh name1 name2 ...
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for name in tree "$@"; do
  lib=""; [ "$name" != tree ] && lib="$GRAFT_REPO_ROOT/ab_tmp/$name.so"
  export ISR_HIP_LIB=$lib
  rm -rf gpurun_out/r05/prof_abl
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/prof_abl -- python3 tools/r05_sparse_prof.py ${MODE:-5} ${PQ:-4915200} > gpurun_out/r05/prof_abl.log 2>&1
  f=$(ls gpurun_out/r05/prof_abl/*/*kernel_stats.csv | head -1)
  echo "== $name"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'corr_fp6' in r['Name'] or 'direct' in r['Name']:
        print("  ", r['Name'].split('(')[1 if r['Name'].startswith('(') else 0][-40:] if False else r['Name'][:60], r['Calls'], "%.3f ms" % (float(r['AverageNs'])/1e6))
PY
done
rm -rf gpurun_out/r05/prof_abl
