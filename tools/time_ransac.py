"""RANSAC stage timings alone: python tools/time_ransac.py  (M = 245 760 correspondences, H = 500)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, synth
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
pts = synth.tless_like(rng, 20000)
K = synth.camera()
R, t = synth.random_poses(rng, 1)
p3d, p2d, _ = synth.pnp_case(rng, pts, K, R[0], t[0], 245760)
p3, p2 = torch.from_numpy(p3d).to(dev), torch.from_numpy(p2d).to(dev)
for name, fn in (("pnp_ransac (H=500, 6 GN)", lambda: ops.pnp_ransac(p3, p2, K, H=500, reperr=2.0, seed=1, refine_iters=6)),):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/20*1e3:.1f} us per call")
