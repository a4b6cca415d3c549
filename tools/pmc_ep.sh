#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# HBM traffic (PMC, separate passes as MI355X_MICROARCH.md prescribes) of the matrix-free estimate_pose kernels: bash tools/pmc_ep.sh <outdir>
out=${1:-gpurun_out/pmc_ep}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/fetch -- python3 $R/tools/time_estimate_pose.py --only-call --reps 3 > $R/$out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$out/write -- python3 $R/tools/time_estimate_pose.py --only-call --reps 3 > $R/$out/write.log 2>&1
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv") + glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as g:
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
        if not any(s in k for s in ("ep_", "zbuf", "corr_")):
            continue
        fetch = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
        write = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
        hit, miss = sum(c.get("TCC_HIT_sum", [0])), sum(c.get("TCC_MISS_sum", [0]))
        # gfx950: FETCH_SIZE counts 128-byte requests at 64 B (x2), both in KiB
        g.write(f"{k:46s} per launch: read {2 * fetch * 1024 / 1e6:9.2f} MB  written {write * 1024 / 1e6:9.2f} MB  L2 hit rate {hit / max(hit + miss, 1):.3f}\n")
print(open(out + "/summary.txt").read())
PY
