#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# HBM traffic of K1 for a given launch size: bash tools/pmc_traffic.sh <outdir> P N D
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 tools/time_corr.py "$@" > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/write" -- python3 tools/time_corr.py "$@" > "$out/write.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "corr_bf16_direct_kernel" in k: name = "direct"
        elif "corr_bf16_kernel" in k and "Lb1" not in k and "true" not in k: name = "natural"
        else: continue
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, c), v in sorted(agg.items()):
    print(f"{name:8s} {c:16s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
