#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# On the GPU box: rebuild ransac.hip with different (correspondences per lane, hypotheses per block)
# and time the score kernel alone.   bash tools/score_sweep.sh
cd "$GRAFT_REPO_ROOT"
for v in "4 32" "4 64" "8 32" "2 32" "4 16" "8 64" "2 64"; do
  set -- $v
  python3 - <<PY > /dev/null 2>&1
import sys, os
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import build as B
B.EXTRA_FLAGS["ransac.hip"] = ["-DISR_SCORE_CPL=$1", "-DISR_SCORE_HC=$2"]
os.utime(str(B.CSRC / "ransac.hip"))
B.build_hip()
PY
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  rm -rf gpurun_out/prof_sw
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sw -- python3 tools/time_ransac.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_sw/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "score_kernel" in r["Name"]: print("CPL=$1 HC=$2 score_kernel", r["AverageNs"], "ns")
PY
done
