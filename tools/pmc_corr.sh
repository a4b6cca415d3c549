#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# PMC passes over tools/time_corr.py (K1 alone).  Separate passes: SQ has 8 slots, TCC 4.
# Usage on the GPU box: bash tools/pmc_corr.sh <outdir> [P N D]
set -u
out=${1:-gpurun_out/pmc_corr}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python3 tools/time_corr.py ${ARGS:-} > "$out/$name.log" 2>&1; }
ARGS="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VMEM
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "corr_" not in k: continue
        if "finalize" in k: k = "corr_finalize"
        elif "direct" in k: k = "corr_bf16_direct_kernel<log2>" if "false>" in k else "corr_bf16_direct_kernel<natural>"
        elif "corr_bf16_kernel" in k: k = "corr_bf16_kernel(fallback, log2)" if ("true" in k or "Lb1" in k) else "corr_bf16_kernel"
        else: k = k.split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as g:
    for k, d in agg.items():
        g.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            g.write(f"  {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
        if "direct" in k and "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
            # per 32x32 tile: a wave owns 64 queries = 2 query blocks x ceil(N/32) key tiles
            import os
            N = int(os.environ.get("PMC_N", "20000"))
            tiles = (sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])) * 2 * ((N + 31) // 32)
            m = lambda c: sum(d[c]) / len(d[c])
            g.write(f"  -- per 32x32 tile: VALU+MFMA instructions {m('SQ_INSTS_VALU') / tiles:.1f} (MFMA {m('SQ_INSTS_MFMA') / tiles:.2f}, "
                    f"transcendental {m('SQ_INSTS_VALU_TRANS_F32') / tiles:.1f}), SALU {m('SQ_INSTS_SALU') / tiles:.2f}, LDS {m('SQ_INSTS_LDS') / tiles:.2f}\n")
            if "SQ_BUSY_CU_CYCLES" in d:
                g.write(f"  -- VALU active {m('SQ_ACTIVE_INST_VALU') / m('SQ_BUSY_CU_CYCLES'):.3f} (SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES), MFMA busy "
                        f"{m('SQ_VALU_MFMA_BUSY_CYCLES') / m('SQ_BUSY_CU_CYCLES') / 4:.3f} (SQ_VALU_MFMA_BUSY_CYCLES / 4 SQ_BUSY_CU_CYCLES), LDS bank conflicts {m('SQ_LDS_BANK_CONFLICT'):.0f}\n")
print(open(out + "/summary.txt").read())
PY
