"""One line per bench JSON line on stdin: value, ms/step, K1 ms per launch in the step / alone, frac, parity."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r, p = d["roofline"], d.get("parity_check", {})
print(round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms/step | K1", round(r["ms_per_launch"], 2), "in step",
      round(r["alone"]["ms_per_launch"], 2), "alone | frac", round(r["frac"], 4), "| f32_exact", 
      {k: round(v, 3) for k, v in r.get("f32_exact", {}).items() if isinstance(v, float)},
      "| parity", p.get("k1_idx_equal_rows"), p.get("pick_idx_equal"), p.get("k1_in_step_equals_alone"))
for k in ("f32_step", "screened_step"):
    v = d.get(k)
    if v:
        print(" ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, dict)},
              {a: {x: (round(y, 3) if isinstance(y, float) else y) for x, y in b.items()} for a, b in v.items() if isinstance(b, dict)})
if d.get("dist", {}).get("initialized"):
    print("  dist", d["dist"], "per-rank ms/step", d.get("per_rank_ms_per_step", {}).get("all"))
