#!/bin/bash
# K2 (score_kernel) instruction mix and HBM traffic, separate PMC passes (MI355X_MICROARCH.md): bash tools/pmc_k2.sh [outdir]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the snapshot root)}"
R=$GRAFT_REPO_ROOT; out=${1:-gpurun_out/pmc_k2}
mkdir -p "$R/$out"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/time_ransac.py --reps 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/trace" -- $CMD > "$R/$out/trace.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d "$R/$out/insts" -- $CMD > "$R/$out/insts.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$out/busy" -- $CMD > "$R/$out/busy.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$out/fetch" -- $CMD > "$R/$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$R/$out/write" -- $CMD > "$R/$out/write.log" 2>&1
cd "$R"
python3 tools/pmc_k2_summary.py "$out" | tee "$out/summary.txt"
