"""isr_zbuf_score_direct (the fused LDS z-buffer scorer) at the reference's size: ms per launch for B poses, both descriptor
grids.   python tools/time_zbuf.py [--B 224 500 1000]"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_est_surf as pes, synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, nargs="+", default=[224, 500, 1000])
args = ap.parse_args()
dev = torch.device("cuda:0")
s = synth.crop_scene()
ml, q_img = torch.from_numpy(s["mask_lgts"]).to(dev), torch.from_numpy(s["query"]).to(dev)
keys, pts = torch.from_numpy(s["keys"]).to(dev), torch.from_numpy(s["pts"]).to(dev)
mlp, nmlp, mprob, queries, res = pes.prepare(ml, q_img, 3, True)
Ks = pes._k_scaled(s["K"], 3)
rng = np.random.default_rng(0)
for name, grid in (("pooled queries (3 x 3 window)", pes.DescriptorGrid.pooled(queries, keys, res)),
                   ("per-pixel queries (9 x 9 window)", pes.DescriptorGrid.per_pixel(q_img, keys, 3))):
    for B in args.B:
        Rs, ts = [], []
        for _ in range(B):
            R, t = synth.perturb_pose(rng, s["R"], s["t"], 8.0, 6.0)
            Rs.append(R); ts.append(t)
        R_d = torch.from_numpy(np.stack(Rs)).float().to(dev)
        t_d = torch.from_numpy(np.stack(ts)).float().to(dev)
        pes.zbuf_score_direct(pts, R_d, t_d, Ks, res, mlp, nmlp, grid, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = pes.zbuf_score_direct(pts, R_d, t_d, Ks, res, mlp, nmlp, grid, True)
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: B = {B}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per launch, {e0.elapsed_time(e1) / 10 * 1e3 / B:.2f} us per pose; "
              f"best score {float(out[0].max()):.4f}")
