"""estimate_pose at the reference's size (r = 224, e = 12, m = 80 000, 10 000 samples, <= 1 000 scored poses) timed per call
and — for isr_corr_logsoftmax, the HBM-bound stage — per launch with HIP events.
    python tools/time_estimate_pose.py [--reps 5] [--avg-queries 0|1]
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table (profiles/r03_estimate_pose_ref_size.csv)."""
import argparse, sys, time
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, pose_est_surf as pes
from test_gpu_estimate_pose import _scene_ref

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--avg-queries", type=int, default=1)
ap.add_argument("--materialize", type=int, default=0, help="1: the route through the (n, m) arrays; 0: matrix-free (the default of estimate_pose)")
ap.add_argument("--only-call", action="store_true", help="skip the per-stage matrix timings")
ap.add_argument("--block", type=int, default=0, help="> 0: also time estimate_poses on a block of that many crops")
ap.add_argument("--streams", type=int, default=4)
args = ap.parse_args()
dev = torch.device("cuda:0")
s = _scene_ref()
a = [torch.from_numpy(s["mask_lgts"]).to(dev), torch.from_numpy(s["query"]).to(dev), torch.from_numpy(s["pts"]).to(dev),
     torch.from_numpy(s["normals"]).to(dev), torch.from_numpy(s["keys"]).to(dev), s["diameter"], s["K"]]
kw = dict(max_poses=10000, max_pose_evaluations=1000, avg_queries=bool(args.avg_queries), seed=3, materialize=bool(args.materialize))
out = pes.estimate_pose(*a, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    out = pes.estimate_pose(*a, **kw)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.reps * 1e3
print(f"estimate_pose(avg_queries={bool(args.avg_queries)}, materialize={bool(args.materialize)}) r=224 e=12 m=80000 max_poses=10000: {ms:.2f} ms per call; "
      f"{out[0].shape[0]} poses scored, best score {float(out[2].max()):.4f}")
if args.block:
    B = args.block
    ml, q = a[0][None].expand(B, -1, -1).contiguous(), a[1][None].expand(B, -1, -1, -1).contiguous()
    kwb = {k: v for k, v in kw.items() if k != "seed"}
    outs = pes.estimate_poses(ml, q, a[2], a[3], a[4], a[5], a[6], n_streams=args.streams, **kwb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        outs = pes.estimate_poses(ml, q, a[2], a[3], a[4], a[5], a[6], n_streams=args.streams, **kwb)
    torch.cuda.synchronize()
    msb = (time.perf_counter() - t0) / args.reps * 1e3 / B
    print(f"estimate_poses(block of {B}, {args.streams} streams, avg_queries={bool(args.avg_queries)}): {msb:.3f} ms per image = "
          f"{1e3 / msb:.0f} images/s ({ms / msb:.2f}x the single call); poses scored per image {sorted(set(o[0].shape[0] for o in outs))}")
if args.only_call:
    sys.exit(0)
# the materialised log-softmax matrix alone: 4 n m bytes written (the reference keeps the same matrix resident)
mlp, nmlp, mprob, q, res = pes.prepare(a[0], a[1], 3, True)
n, m = q.shape[0], a[4].shape[0]
ops.corr_logsoftmax(q, a[4]); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    c = ops.corr_logsoftmax(q, a[4])
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
byts = 4.0 * n * m
print(f"isr_corr_logsoftmax n={n} m={m}: {t * 1e3:.3f} ms, {byts / t * 1e-12:.2f} TB/s written = {byts / t / 8e12:.3f} of 8 TB/s "
      f"({byts * 1e-9:.2f} GB algorithmic)")
e0.record()
for _ in range(10):
    p = pes.pool_corr(c, res)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(f"isr_ep_pool_corr (3x3 spatial max-pool of the matrix, the two-step route): {t * 1e3:.3f} ms, {2 * byts / t * 1e-12:.2f} TB/s "
      f"(read + write) = {2 * byts / t / 8e12:.3f} of 8 TB/s")
pes.corr_matrices(q, a[4], res, True); torch.cuda.synchronize()
e0.record()
for _ in range(10):
    raw, pooled = pes.corr_matrices(q, a[4], res, True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(f"isr_ep_corr_matrices (matrix + pooled twin in one pass, incl. the K1 lse launch): {t * 1e3:.3f} ms, {2 * byts / t * 1e-12:.2f} TB/s "
      f"written = {2 * byts / t / 8e12:.3f} of 8 TB/s ({2 * byts * 1e-9:.2f} GB algorithmic)")
