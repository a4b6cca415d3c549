#!/bin/bash
# round 4, session 2: does the side work's launch traffic (cache maintenance at kernel boundaries) cost K1 its L2-resident keys?
# FETCH_SIZE / TCC hits of corr_bf16_direct_kernel inside the step against the same launch alone (separate --pmc passes)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; out=gpurun_out/s2/k1_fetch; mkdir -p $out
pass() { name=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python3 bench.py --steps 4 --no-cpu-baseline --no-parity-check --no-estimate-pose > "$out/$name.log" 2>&1; }
pass fetch FETCH_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "corr_bf16_direct_kernel<4" in r["Kernel_Name"] or "corr_bf16_direct_kernelILi4" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for c, v in rows.items():
    v.sort()
    print(c, " ".join(f"{x[1]:.4g}" for x in v))
PY
