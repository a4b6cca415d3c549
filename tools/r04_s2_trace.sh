#!/bin/bash
# round 4, session 2: the K1 stream's timeline inside the step (rocprofv3 --kernel-trace): gaps between consecutive direct-kernel launches and what runs in them
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2/trace -- python3 bench.py --steps 6 --no-cpu-baseline --no-parity-check --no-estimate-pose > gpurun_out/s2/bench_trace.json 2> gpurun_out/s2/bench_trace.err || { tail -20 gpurun_out/s2/bench_trace.err; exit 1; }
f=$(ls gpurun_out/s2/trace/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/s2/k1_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
k1 = [r for r in rows if "corr_bf16_direct_kernel<4, 2, false, 4, 0, false, false>" in r["Kernel_Name"]]
qid = k1[0].get("Queue_Id")
print("direct-kernel launches", len(k1), "columns", list(rows[0].keys()))
for a, b in zip(k1[2:14], k1[3:15]):
    gap = (b["s"] - a["e"]) / 1e3
    inside = [r for r in rows if r["s"] >= a["e"] and r["e"] <= b["s"] and r.get("Queue_Id") == a.get("Queue_Id")]
    names = ", ".join(f"{r['Kernel_Name'].split('(')[0].replace('void (anonymous namespace)::','').replace('(anonymous namespace)::','')[:28]} {(r['e']-r['s'])/1e3:.0f}us" for r in inside[:8])
    print(f"dur {(a['e']-a['s'])/1e6:7.3f} ms  gap to next {gap:8.1f} us  same-queue kernels in the gap: {names}")
PY
rm -rf gpurun_out/s2/trace
