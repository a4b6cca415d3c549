"""The side work inside K1's windows, from a rocprofv3 kernel trace of bench.py:
    python tools/side_budget.py <kernel_trace.csv> [steps]
Per side kernel: launches, workgroups per launch, mean duration, the part of its lifetime that falls inside a K1 launch, and
workgroup-launches per step — the side work costs K1 register slots one K1 workgroup at a time, so the
NUMBER of side workgroups that must find a slot matters beside their arithmetic."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = [int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
    w = [int(r[k]) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
    r["wgs"] = (g[0] // max(w[0], 1)) * (g[1] // max(w[1], 1)) * (g[2] // max(w[2], 1))
    r["vgpr"] = int(r.get("VGPR_Count", 0) or 0)
k1 = sorted((r for r in rows if "corr_bf16_direct_kernel<4, 2, false, 4, 0" in r["Kernel_Name"] and r["e"] - r["s"] > 5_000_000), key=lambda r: r["s"])
if steps is None:
    steps = len(k1) // 2
# the timed region: from the first of the last 2 * steps K1 launches
k1 = k1[-2 * steps:]
t0, t1 = k1[0]["s"], k1[-1]["e"]
win = [(r["s"], r["e"]) for r in k1]
agg = defaultdict(lambda: [0, 0, 0.0, 0.0, 0])
for r in rows:
    if r["e"] < t0 or r["s"] > t1 or r in k1:
        continue
    inside = sum(max(0, min(r["e"], b) - max(r["s"], a)) for a, b in win)
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:52]
    a = agg[name]
    a[0] += 1; a[1] += r["wgs"]; a[2] += (r["e"] - r["s"]) * 1e-6; a[3] += inside * 1e-6; a[4] = r["vgpr"]
k1_ms = sum(b - a for a, b in win) * 1e-6 / len(win)
print(f"{steps} steps, {len(k1)} K1 launches of {k1_ms:.2f} ms; side kernels between {t0} and {t1} ns, per STEP:")
print(f"{'kernel':54s} {'launches':>8s} {'WGs/launch':>10s} {'WG launches':>11s} {'ms (sum of durations)':>22s} {'inside K1':>10s} {'VGPRs':>6s}")
tot = [0, 0, 0.0, 0.0]
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    if a[3] / steps < 0.02:
        continue
    print(f"{name:54s} {a[0] / steps:8.1f} {a[1] / max(a[0], 1):10.0f} {a[1] / steps:11.0f} {a[2] / steps:22.2f} {a[3] / steps:10.2f} {a[4]:6d}")
    tot[0] += a[0] / steps; tot[1] += a[1] / steps; tot[2] += a[2] / steps; tot[3] += a[3] / steps
print(f"{'total (listed)':54s} {tot[0]:8.1f} {'':10s} {tot[1]:11.0f} {tot[2]:22.2f} {tot[3]:10.2f}")

# how many of the Gauss-Newton launches still do work (the refit stops on a step-norm criterion, csrc/ransac.hip gn_solve: later
# launches of the fixed 6 x 2 leave at once)
gn = sorted((r for r in rows if "gn_accumulate_kernel" in r["Kernel_Name"] and t0 <= r["s"] <= t1), key=lambda r: r["s"])
if gn:
    import statistics
    d = [(r["e"] - r["s"]) * 1e-3 for r in gn]
    short = sum(1 for x in d if x < 25.0)
    print(f"gn_accumulate_kernel: {len(d) / steps:.1f} launches per step, {short / steps:.1f} of them shorter than 25 us (every image converged: "
          f"early exit), median of the others {statistics.median([x for x in d if x >= 25.0]):.0f} us")

# K1's duty cycle over the timed region: gaps between consecutive K1 launches are time the chip runs side work only (or idles)
gaps = [(k1[i + 1]["s"] - k1[i]["e"]) * 1e-3 for i in range(len(k1) - 1)]
busy = sum(b - a for a, b in win)
print(f"K1 duty cycle {busy / (t1 - t0):.4f}: {len(gaps)} gaps, mean {sum(gaps) / len(gaps):.0f} us, max {max(gaps):.0f} us; "
      f"even gaps (inside a step) mean {sum(gaps[0::2]) / len(gaps[0::2]):.0f} us, odd gaps (between steps) mean {sum(gaps[1::2]) / max(len(gaps[1::2]), 1):.0f} us")

# what runs in the gap between two steps: the kernels alive between the end of a step's second K1 launch and the next step's first
print("all gaps in launch order, us: " + " ".join(f"{g:.0f}" for g in gaps))
odd = sorted(range(1, len(gaps), 2), key=lambda j: gaps[j])
if odd:
    i = odd[len(odd) // 2]                       # the median between-step gap
    a, b = k1[i]["e"], k1[i + 1]["s"]
    print(f"gap of {(b - a) * 1e-3:.0f} us between K1 launches {i} and {i + 1}; kernels overlapping it (start / end relative to the gap's start, us):")
    for r in sorted(rows, key=lambda r: r["s"]):
        if r["e"] > a - 100_000 and r["s"] < b + 50_000 and r not in k1:
            nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
            print(f"   {nm:46s} {(r['s'] - a) * 1e-3:9.0f} {(r['e'] - a) * 1e-3:9.0f}  wgs {r['wgs']:6d}  queue {r.get('Queue_Id', '?')} stream {r.get('Stream_Id', '?')}")
    print(f"   next K1 starts at {(b - a) * 1e-3:.0f}")
