#!/bin/bash
# SQ counters of the three-plane exact-f32 K1: bash tools/pmc_k1_split3.sh [outdir] [P N D]   (default: one 640 x 480 x 64-D image vs 20 000 keys)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
out=${1:-gpurun_out/pmc_split3}; R=$GRAFT_REPO_ROOT
export ISR_PMC_P=${2:-307200} ISR_PMC_N=${3:-20000} ISR_PMC_D=${4:-64}
export ISR_PMC_KERNEL="corr_bf16_direct_kernel<$(( (ISR_PMC_D + 15) / 16 * 3 ))"
[ "$ISR_PMC_D" -gt 32 ] && export ISR_PMC_KERNEL="corr_bf16_direct_kernel<12"
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/$out"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$out/$name" -- python3 "$R/tools/time_corr_f32.py" $ISR_PMC_P $ISR_PMC_N $ISR_PMC_D ${ISR_PMC_ROUTE:-0} > "$R/$out/$name.log" 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE
cd "$R"
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
kern = os.environ["ISR_PMC_KERNEL"]
P, N, D = (int(os.environ[k]) for k in ("ISR_PMC_P", "ISR_PMC_N", "ISR_PMC_D"))
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
m = {c: sum(v) / len(v) for c, v in agg.items()}
with open(out + "/summary.txt", "w") as g:
    g.write(f"# {kern}...> at P = {P}, N = {N}, D = {D} f32; kernel {sum(dur) / max(len(dur), 1):.3f} ms per launch under the profiler (n = {len(dur)})\n")
    for c in sorted(m): g.write(f"{c:32s} n={len(agg[c]):3d} mean={m[c]:.6g}\n")
    if "SQ_WAVES" in m:
        tiles = m["SQ_WAVES"] * 2 * ((N + 31) // 32)
        g.write(f"-- per 32x32 tile and query block: VALU+MFMA instructions {m['SQ_INSTS_VALU'] / tiles:.1f} (MFMA {m['SQ_INSTS_MFMA'] / tiles:.2f}, transcendental {m.get('SQ_INSTS_VALU_TRANS_F32', 0) / tiles:.1f}), "
                f"SALU {m.get('SQ_INSTS_SALU', 0) / tiles:.2f}, LDS {m.get('SQ_INSTS_LDS', 0) / tiles:.2f}, VMEM {m.get('SQ_INSTS_VMEM', 0) / tiles:.2f}, branches {m.get('SQ_INSTS_BRANCH', 0) / tiles:.2f}\n")
        if "SQ_BUSY_CU_CYCLES" in m:
            g.write(f"-- VALU active {m['SQ_ACTIVE_INST_VALU'] / m['SQ_BUSY_CU_CYCLES']:.3f}, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CU_CYCLES'] / 4:.3f}, "
                    f"co-exec {m.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / m['SQ_BUSY_CU_CYCLES'] / 4:.3f}, LDS bank conflicts {m.get('SQ_LDS_BANK_CONFLICT', 0):.0f} "
                    f"of {m.get('SQ_LDS_IDX_ACTIVE', 0):.3g} LDS-active cycles, wave cycles / busy {m['SQ_WAVE_CYCLES'] / m['SQ_BUSY_CYCLES']:.2f}\n")
            g.write(f"-- of the wave cycles: waiting (s_waitcnt / barrier) {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}, issue stalls {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} "
                    f"(LDS issue {m.get('SQ_WAIT_INST_LDS', 0) / m['SQ_WAVE_CYCLES']:.3f}), issuing {m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}\n")
        if "GRBM_GUI_ACTIVE" in m and dur:
            g.write(f"-- effective clock {m['GRBM_GUI_ACTIVE'] / 8 / (sum(dur) / len(dur) * 1e-3) * 1e-6:.0f} MHz (GRBM_GUI_ACTIVE / 8 / wall)\n")
print(open(out + "/summary.txt").read())
PY
