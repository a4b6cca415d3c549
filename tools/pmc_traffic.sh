#!/bin/bash
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
        if "corr_bf16_kernel" in r["Kernel_Name"] and "Lb1" not in r["Kernel_Name"] and "true" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:16s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
