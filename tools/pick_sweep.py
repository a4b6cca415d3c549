"""Cell-size sweep of the block-cooperative grid search at the BENCH's Chamfer-pick shape (the reference's
rotation-only, convention-mixing pair clouds of verfication.py:83-101, 63 pairs of 20 000 points):
python tools/pick_sweep.py   (ISR_NN_TILE="target scale,query scale,threads")."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, registration, synth
dev = torch.device("cuda:0")
N, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 64)
pts = torch.from_numpy(synth.tless_like(np.random.default_rng(20240), N)).to(dev)
rng = np.random.default_rng(99)
Rg, tg = synth.random_poses(rng, n)
Rp = np.array([synth.perturb_pose(rng, Rg[i], tg[i], 0.05, 0.1)[0] for i in range(n)])
Rrel = np.einsum("nij,nkj->nik", Rg[1:], Rg[:-1])
def run():
    ch = registration.chamfer_pairs(pts, Rp, Rrel)
    return ch
ref = None
for plan in ("brute", "default", "6,11,64", "4,11,64", "3,11,64", "4,8,64", "3,8,64", "3,6,64", "2.5,8,64", "2,8,64", "2,6,64", "3,8,128", "4,16,64"):
    if plan == "brute":
        ops.set_tuning(nn_path=0)
    else:
        ops.set_tuning(nn_path=2)
        if plan == "default":
            ops.set_tile_plan(None)
        else:
            ops.set_tile_plan(plan)
    ch = run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ch = run()
    e1.record(); torch.cuda.synchronize()
    c = ch.cpu().numpy()
    if ref is None:
        ref = c
    print(f"{plan:10s} {e0.elapsed_time(e1) / 3:7.3f} ms  identical to brute force: {np.array_equal(c, ref)}  mean Chamfer {c.mean():.3f}", flush=True)
