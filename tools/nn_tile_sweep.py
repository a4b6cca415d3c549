"""Cell-size / workgroup-size sweep of the block-cooperative grid search
(ISR_NN_TILE="target scale,query scale,threads per workgroup")."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, synth
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def poses(B, near=None, rot=3.0):
    if near is None:
        R, t = synth.random_poses(rng, B)
    else:
        R, t = np.empty((B, 3, 3)), np.empty((B, 3))
        for i in range(B):
            R[i], t[i] = synth.perturb_pose(rng, near[i, :, :3], near[i, :, 3], rot, 2.0)
    return np.concatenate([R, t[:, :, None]], 2)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cloud = torch.from_numpy(np.ascontiguousarray(synth.tless_like(rng, N))).to(dev)
Ta = poses(63)
cases = {"near3deg": poses(63, Ta, 3.0), "mid15deg": poses(63, Ta, 15.0), "random": poses(63)}
ops.set_tuning(nn_path=0)
for name, Tb in cases.items():
    Tq, Tt = torch.from_numpy(Ta).to(dev), torch.from_numpy(Tb).to(dev)
    ops.nn_batched(cloud, cloud, Tq, Tt); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [ops.nn_batched(cloud, cloud, Tq, Tt) for _ in range(3)]; e1.record(); torch.cuda.synchronize()
    print(f"N={N} {name:9s} brute {e0.elapsed_time(e1)/3:7.3f} ms", flush=True)
ops.set_tuning(nn_path=2)
for plan in ("6,16,256", "6,11,64", "6,16,64", "8,11,64", "6,13,64", "5,11,64", "6,16,128", "8,16,64", "6,22,64"):
    ops.set_tile_plan(plan)
    out = []
    for name, Tb in cases.items():
        Tq, Tt = torch.from_numpy(Ta).to(dev), torch.from_numpy(Tb).to(dev)
        ops.nn_batched(cloud, cloud, Tq, Tt); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [ops.nn_batched(cloud, cloud, Tq, Tt) for _ in range(3)]; e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1)/3:7.3f}")
    print(f"N={N} tile {plan:10s}: " + " | ".join(out), flush=True)
