"""Reduce tools/pmc_k2.sh's passes to one table: per score_kernel launch, counters and what they mean per
(hypothesis, correspondence) pair.  gfx950: read bytes = 2 x FETCH_SIZE (KiB), WRITE_SIZE exact (KiB)."""
import collections, csv, glob, json, sys
out = sys.argv[1]
M, H = 245760, 4096
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
mean = lambda v: sum(v) / max(len(v), 1)
n_ok = None
for line in open(out + "/trace.log"):
    if line.startswith("{"):
        rec = json.loads(line)
        print("bench.measure_ransac under rocprofv3 --kernel-trace:", json.dumps(rec))
        n_ok = int(rec["shape"].split("(")[1].split()[0])
pairs = (n_ok or H) * M
print(f"\nkernels of isr_ransac_score, us per launch (rocprofv3 --kernel-trace, {len(dur.get('score_kernel', []))} score launches):")
for k in ("proj_matrix_kernel", "score_kernel", "best_kernel", "best_mask_kernel", "p3p_kernel"):
    if k in dur:
        print(f"  {k:22s} {mean(dur[k]):9.1f} us  (n = {len(dur[k])})")
c = agg.get("score_kernel", {})
if c:
    g = lambda n: mean(c.get(n, [0.0]))
    print(f"\nscore_kernel, per launch (H = {H}, {n_ok} scored, M = {M}: {pairs:.3e} pairs):")
    for n in sorted(c):
        print(f"  {n:24s} {g(n):16.1f}")
    # SQ_INSTS_* count per wave (one per 64 lanes); a pair is one lane's work
    valu = g("SQ_INSTS_VALU")
    print(f"  VALU instructions per pair        {valu * 64 / pairs:8.2f}   (wave instructions x 64 lanes / pairs)")
    print(f"  SALU instructions per pair        {g('SQ_INSTS_SALU') * 64 / pairs:8.2f}")
    print(f"  LDS instructions per pair         {g('SQ_INSTS_LDS') * 64 / pairs:8.3f}")
    if g("SQ_BUSY_CU_CYCLES"):
        print(f"  VALU active / CU busy cycles      {g('SQ_ACTIVE_INST_VALU') * 4 / g('SQ_BUSY_CU_CYCLES'):8.3f}   (SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CU_CYCLES, the K1 summaries' convention)")
    rd, wr = 2 * g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
    alg = 20.0 * M + 52.0 * H + M / 8.0
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    print(f"  HBM read {rd / 1e6:8.2f} MB  written {wr / 1e6:8.2f} MB  (algorithmic {alg / 1e6:.2f} MB)  L2 hit rate {hit / max(hit + miss, 1):.3f}")
