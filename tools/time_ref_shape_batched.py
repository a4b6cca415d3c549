"""The reference's own per-image loop shape (inference.py:163, 248-293; genFeat.py:201), BATCHED:
1 280 images of a 224 x 224 crop -> 75 x 75 lattice (P_cap = 5 625), D = 12, N = 80 000 keys, through
sequence.register_crops (isr_prep_queries_batch + ONE K1 launch per group + one filter / RANSAC chain per group)
against the single-image chain sequence.register_crop.

    python tools/time_ref_shape_batched.py [--batch 64] [--images 1280] [--distinct 128] [--dtype f32|bf16] [--once]

--once: one pass over the distinct crops and nothing else (for rocprofv3 --kernel-trace: launches per image)."""
import argparse, sys, time
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, sequence, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--images", type=int, default=1280)
ap.add_argument("--distinct", type=int, default=128)
ap.add_argument("--dtype", default="both")
ap.add_argument("--itr", type=int, default=500)
ap.add_argument("--once", action="store_true")
ap.add_argument("--streams", type=int, default=1, help="1: K1 and the chains on one stream; 3: chains on side streams beside the next K1")
args = ap.parse_args()
dev = torch.device("cuda:0")
n = args.distinct
cb = synth.crop_batch(dev, n)
N, D, ds, S1, Kc, R, counts = cb["keys"].shape[0], cb["D"], cb["ds"], cb["S1"], cb["Kc"], cb["R"], cb["counts"]
feats, masks, pts_d, keys_d = cb["feats"], cb["masks"], cb["pts"], cb["keys"]
torch.cuda.synchronize()
print(f"{n} distinct crops, masked lattice pixels per crop: mean {np.mean(counts):.0f} (min {min(counts)}, max {max(counts)}) of {S1 * S1}; "
      f"N = {N}, D = {D}, itr = {args.itr}", flush=True)
cams = np.broadcast_to(Kc, (n, 3, 3))
reps = max(1, args.images // n)


def run(model, label):
    seeds = list(range(n))
    if args.once:
        res, nd = sequence.register_crops(model, feats, masks, cams, n_feat=D, down_sample=ds, itr=args.itr, seeds=seeds,
                                          refine_iters=6, group=args.batch, n_streams=args.streams)
        torch.cuda.synchronize()
        return
    # single-image chain (round 2's path) on a subset
    m1 = min(n, 64)
    for i in range(4):
        sequence.register_crop(model, feats[i:i + 1], masks[i], Kc, n_feat=D, down_sample=ds, itr=args.itr, seed=i, refine_iters=6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    single = [sequence.register_crop(model, feats[i:i + 1], masks[i], Kc, n_feat=D, down_sample=ds, itr=args.itr, seed=i,
                                     refine_iters=6)[0] for i in range(m1)]
    torch.cuda.synchronize()
    t_single = (time.perf_counter() - t0) / m1
    # batched
    ops.enable_timing(True)
    res, nd = sequence.register_crops(model, feats, masks, cams, n_feat=D, down_sample=ds, itr=args.itr, seeds=seeds,
                                      refine_iters=6, group=args.batch, n_streams=args.streams)
    torch.cuda.synchronize()
    ops.drain_timing()
    t0 = time.perf_counter()
    for _ in range(reps):
        res, nd = sequence.register_crops(model, feats, masks, cams, n_feat=D, down_sample=ds, itr=args.itr, seeds=seeds,
                                          refine_iters=6, group=args.batch, n_streams=args.streams)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = ops.drain_timing()
    ops.enable_timing(False)
    calls, ms, flop = tm["corr_argmax"]
    same = all(torch.equal(res[i].pose, single[i].pose) and torch.equal(res[i].idx[:counts[i]], single[i].idx[:counts[i]])
               for i in range(m1))
    ok = sum(int(r.status.item()) for r in res)
    errs = [synth.rot_angle(res[i].pose.cpu().numpy()[:, :3], R[i]) for i in range(n) if int(res[i].status.item())]
    print(f"[{label}] single-image chain: {t_single * 1e3:.3f} ms/image = {1 / t_single:.0f} images/s")
    print(f"[{label}] batched, {args.batch} images per group, {reps * n} images: {dt / (reps * n) * 1e3:.4f} ms/image = "
          f"{reps * n / dt:.0f} images/s ({1 / t_single and (reps * n / dt) * t_single:.1f}x the single-image rate); "
          f"K1 {ms / calls:.3f} ms per launch of {args.batch} x {S1 * S1} rows = {flop / (ms * 1e-3) * 1e-12:.1f} TFLOP/s "
          f"(2 P N D with D as padded); registered {ok}/{n}, median rot err {np.median(errs):.2e} rad; "
          f"per-image results identical to the single-image chain: {same}", flush=True)


if args.dtype in ("f32", "both"):
    run(sequence.SequenceModel(keys=keys_d, pts=pts_d), "f32 exact, D = 12")
if args.dtype in ("bf16", "both"):
    run(sequence.SequenceModel(keys=keys_d.bfloat16(), pts=pts_d, log2_queries=True), "bf16 log2, D = 12 padded to 16")
