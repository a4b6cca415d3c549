"""Print the top rows of a rocprofv3 kernel_stats.csv: python tools/kstats.py file.csv [rows]."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms in {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    print(f"{name:64s} calls {int(r['Calls']):6d} total {float(r['TotalDurationNs']) / 1e6:9.2f} ms avg {float(r['AverageNs']) / 1e3:9.1f} us"
          f" min {float(r['MinNs']) / 1e3:8.1f} max {float(r['MaxNs']) / 1e3:9.1f}")
