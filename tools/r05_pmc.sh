#!/bin/bash
# SQ counters of the screened route's kernels: bash tools/r05_pmc.sh <name> [mode] [P]
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; mode=${2:-5}; P=${3:-4915200}
out=gpurun_out/r05/pmc_$name
mkdir -p $out
pass() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$n -- python3 tools/r05_sparse_prof.py $mode $P > $out/$n.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM
pass grbm GRBM_GUI_ACTIVE
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fp6_lower" in k: k = "pass0"
        elif "fp6_sparse" in k: k = "pass1"
        elif "direct" in k: k = "direct"
        else: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as g:
    for k, d in agg.items():
        g.write(f"== {k}\n")
        m = {c: sum(v) / len(v) for c, v in d.items()}
        for c in sorted(m): g.write(f"  {c:28s} {m[c]:.6g}\n")
        if "SQ_BUSY_CU_CYCLES" in m:
            b = m["SQ_BUSY_CU_CYCLES"]
            g.write(f"  -- VALU active {m.get('SQ_ACTIVE_INST_VALU',0)/b:.3f}  MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/4/b:.3f}  LDS active {m.get('SQ_ACTIVE_INST_LDS',0)/b:.3f}"
                    f"  VMEM active {m.get('SQ_ACTIVE_INST_VMEM',0)/b:.3f}  SCA active {m.get('SQ_ACTIVE_INST_SCA',0)/b:.3f}\n")
print(open(out + "/summary.txt").read())
PY
