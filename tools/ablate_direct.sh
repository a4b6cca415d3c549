#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# On the GPU box: timing-only ablations of corr_bf16_direct_kernel (rebuilds corr_argmax.hip with -D flags).
cd "$GRAFT_REPO_ROOT"
for v in "" "-DISR_ABL_DNOMAX" "-DISR_ABL_DNOEXP" "-DISR_ABL_DNOMAX -DISR_ABL_DNOEXP"; do
  python3 - <<PY > /dev/null 2>&1
import sys, os
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import build as B
B.EXTRA_FLAGS["corr_argmax.hip"] = ["-fno-honor-nans"] + "$v".split()
os.utime(str(B.CSRC / "corr_argmax.hip"))
B.build_hip()
PY
  echo "[$v]"; python3 tools/time_corr.py 4915200 20000 64 2>&1 | grep "bf16-log2:" | head -1
done
python3 - <<PY > /dev/null 2>&1
import sys, os
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import build as B
os.utime(str(B.CSRC / "corr_argmax.hip"))
B.build_hip()
PY
