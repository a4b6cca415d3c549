"""Reduce the counter_collection.csv files of tools/pmc_all.sh to profiles/k1_hbm_traffic.json and
profiles/nn_hbm_traffic.json.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B read
requests at 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact; both are reported in KiB."""
import collections, csv, glob, json, sys
from pathlib import Path

out = Path(sys.argv[1])
ROOT = Path(__file__).resolve().parent.parent


def counters(run, kernels):
    """mean counter value per dispatch for each kernel-name fragment, skipping each kernel's first (warm-up) dispatch"""
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(str(out / run / "*" / "**" / "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            for frag in kernels:
                if frag in r["Kernel_Name"]:
                    agg[(frag, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v[1:]) / max(len(v) - 1, 1) if len(v) > 1 else v[0] for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def hbm(c, frag):
    f, w = c.get((frag, "FETCH_SIZE"), 0.0), c.get((frag, "WRITE_SIZE"), 0.0)
    hit, miss = c.get((frag, "TCC_HIT_sum"), 0.0), c.get((frag, "TCC_MISS_sum"), 0.0)
    return {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes": int(2 * f * 1024 + w * 1024),
            "tcc_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}


src = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes (tools/pmc_all.sh), round 4 kernels (second half)"
k1 = {"kernel": "corr_bf16_direct_kernel<4,2,false> (K1, ISR_DTYPE_BF16_LOG2) + its finalize / recheck passes, N=20000 D=64 bf16",
      "correction": "gfx950: read bytes = 2 x FETCH_SIZE (128-B requests tallied at 64 B), WRITE_SIZE exact, both in KiB",
      "per_launch": {}}
K1_KERNELS = ["corr_bf16_direct_kernel", "corr_finalize_kernel", "corr_bf16_kernel", "corr_recheck_kernel", "corr_keynorm_kernel",
              "corr_recheck_merge_kernel"]
for g, P in (("1", 307200), ("16", 4915200), ("32", 9830400)):
    c, n = counters(f"k1_{g}", K1_KERNELS)
    parts = {k: hbm(c, k) for k in K1_KERNELS}
    total = sum(p["hbm_bytes"] for p in parts.values())
    k1["per_launch"][g] = {"P": P, "hbm_bytes": total, "source": src,
                           "dominant_kernel_hbm_bytes": parts["corr_bf16_direct_kernel"]["hbm_bytes"],
                           "algorithmic_bytes": int(2 * P * 64 + 2 * 20000 * 64 + 8 * P), "by_kernel": parts}
(ROOT / "profiles" / "k1_hbm_traffic.json").write_text(json.dumps(k1, indent=2) + "\n")
c, n = counters("nn", ["nn_search_kernel", "nn_finalize_kernel"])
parts = {k: hbm(c, k) for k in ("nn_search_kernel", "nn_finalize_kernel")}
nn = {"20000x20000x32": {"hbm_bytes": sum(p["hbm_bytes"] for p in parts.values()), "source": src, "by_kernel": parts,
                         "algorithmic_bytes": int(12 * 20000 * 2 + 96 * 32 + 8 * 20000 * 32)}}
(ROOT / "profiles" / "nn_hbm_traffic.json").write_text(json.dumps(nn, indent=2) + "\n")
print(json.dumps({"k1_16": k1["per_launch"]["16"]["hbm_bytes"], "k1_1": k1["per_launch"]["1"]["hbm_bytes"], "nn": nn}, indent=1))
