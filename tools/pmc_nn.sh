#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# HBM traffic of the NN kernels (Chamfer-pair shape) from PMC counters: bash tools/pmc_nn.sh <outdir>
set -u
out=${1:-gpurun_out/pmc_nn}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 tools/time_nn.py > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/write" -- python3 tools/time_nn.py > "$out/write.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "nn_tile_search_kernel" if "nn_tile_search" in k else "nn_search_kernel<4>" if "nn_search_kernel<4" in k else None
        if name: agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, c), v in sorted(agg.items()):
    print(f"{name:24s} {c:14s} n={len(v):3d} mean={sum(v)/len(v):.6g} max={max(v):.6g}")
PY
