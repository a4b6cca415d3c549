#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# SQ counters of the exact-f32 K1 at the crop-batch shape: bash tools/pmc_f32.sh <outdir> [kernel-name substring]
# (default substring: corr_bf16_direct_kernel<8 — the split route, which tools/time_corr_f32.py takes by default; corr_f32_kernel with
# ISR_TUNE_K1_F32_CHAIN: run round 3's first half, profiles/r03_k1_f32_pmc.txt)
out=${1:-gpurun_out/pmc_f32}; R=$GRAFT_REPO_ROOT; export ISR_PMC_KERNEL=${2:-corr_bf16_direct_kernel<8}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$out
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$out/$name" -- python3 $R/tools/time_corr_f32.py > "$R/$out/$name.log" 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VMEM
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
kern = os.environ.get("ISR_PMC_KERNEL", "corr_f32_kernel")
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*counter_collection.csv") + glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {c: sum(v) / len(v) for c, v in agg.items()}
with open(out + "/summary.txt", "w") as g:
    for c in sorted(m): g.write(f"{c:32s} n={len(agg[c]):3d} mean={m[c]:.6g}\n")
    if "SQ_WAVES" in m:
        tiles = m["SQ_WAVES"] * 2 * ((80000 + 31) // 32)
        g.write(f"-- per 32x32 tile and query block: VALU+MFMA instructions {m['SQ_INSTS_VALU'] / tiles:.1f} (MFMA {m['SQ_INSTS_MFMA'] / tiles:.2f}, transcendental {m.get('SQ_INSTS_VALU_TRANS_F32', 0) / tiles:.1f}), "
                f"SALU {m.get('SQ_INSTS_SALU', 0) / tiles:.2f}, LDS {m.get('SQ_INSTS_LDS', 0) / tiles:.2f}, branches {m.get('SQ_INSTS_BRANCH', 0) / tiles:.2f}\n")
        if "SQ_BUSY_CU_CYCLES" in m:
            g.write(f"-- VALU active {m['SQ_ACTIVE_INST_VALU'] / m['SQ_BUSY_CU_CYCLES']:.3f}, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CU_CYCLES'] / 4:.3f}, "
                    f"co-exec {m.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / m['SQ_BUSY_CU_CYCLES'] / 4:.3f}, LDS bank conflicts {m.get('SQ_LDS_BANK_CONFLICT', 0):.0f}, wave cycles / busy {m['SQ_WAVE_CYCLES'] / m['SQ_BUSY_CYCLES']:.2f}\n")
print(open(out + "/summary.txt").read())
PY
