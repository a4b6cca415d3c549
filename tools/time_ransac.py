"""K2 alone: python tools/time_ransac.py [--M 245760] [--H 4096] [--reps 5] [--chain]
isr_ransac_score (every hypothesis, one score_kernel launch) at configs[3]'s hypothesis count, the same call bench.py
reports as `roofline_ransac`; --chain also times the whole pnp_ransac chain (H = 500, 6 GN iterations).
Run directly under rocprofv3 (`-- python3 tools/time_ransac.py`) for the kernel trace and the PMC passes (tools/pmc_k2.sh)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=245760)
ap.add_argument("--H", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--chain", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
pts = torch.from_numpy(synth.tless_like(np.random.default_rng(20240), 20000)).to(dev)
K = synth.camera()
print(json.dumps(bench.measure_ransac(dev, pts, K, M=a.M, H=a.H, reps=a.reps)))
if a.chain:
    rng = np.random.default_rng(0)
    R, t = synth.random_poses(rng, 1)
    p3d, p2d, _ = synth.pnp_case(rng, pts.cpu().numpy(), K, R[0], t[0], a.M)
    p3, p2 = torch.from_numpy(p3d).to(dev), torch.from_numpy(p2d).to(dev)
    fn = lambda: ops.pnp_ransac(p3, p2, K, H=500, reperr=2.0, seed=1, refine_iters=6)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"pnp_ransac (H=500, confidence 0.99, 6 GN): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
