"""Gaps between consecutive K1 launches in a rocprofv3 kernel trace of bench.py:
python tools/k1_gaps.py <kernel_trace.csv>  (what sits between the end of one corr_bf16_direct_kernel and the next)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k1 = [r for r in rows if "corr_bf16_direct_kernel" in r["Kernel_Name"]]
print(len(rows), "kernels,", len(k1), "K1 launches")
gaps = []
for a, b in zip(k1[:-1], k1[1:]):
    ea, sb = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    between = [r for r in rows if int(r["Start_Timestamp"]) >= ea and int(r["End_Timestamp"]) <= sb and r["Queue_Id"] == a["Queue_Id"]]
    gaps.append((sb - ea, [(r["Kernel_Name"][:50], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in between]))
gs = sorted(g for g, _ in gaps)
print("gap between K1 launches on its queue, us: median %.1f  mean %.1f  max %.1f" % (gs[len(gs) // 2] / 1e3, sum(gs) / len(gs) / 1e3, gs[-1] / 1e3))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in k1]
print("K1 duration ms: mean %.3f min %.3f max %.3f" % (sum(dur) / len(dur) / 1e6, min(dur) / 1e6, max(dur) / 1e6))
for g, btw in gaps[4:10]:
    print("  gap %.1f us:" % (g / 1e3), btw)
