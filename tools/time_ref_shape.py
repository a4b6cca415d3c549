"""The reference's own shape (75x75 masked crop, D = 12, N = 80 000) through the exact f32 path
and the whole per-image chain: python tools/time_ref_shape.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, sequence, synth
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
P, N, D = 5625, 80000, 12
pts = synth.tless_like(rng, N); keys = synth.unit_keys(rng, N, D, tau=6.0)
K = synth.camera(75, 75, f=120.0)
R, t = synth.random_poses(rng, 1, tz=300.0, t_sigma=3.0)
Q, pix, gm, gg = synth.image_case(rng, keys, pts, K, R[0], t[0], P, sigma=0.25)
q, k = torch.from_numpy(Q).to(dev), torch.from_numpy(keys).to(dev)
for name, qq, kk in (("f32 exact", q, k), ("bf16", q.bfloat16(), k.bfloat16())):
    idx, logp = ops.corr_argmax(qq, kk); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.corr_argmax(qq, kk)
    e1.record(); torch.cuda.synchronize()
    print(f"getCors {name}: P={P} N={N} D={D}: {e0.elapsed_time(e1)/20*1e3:.1f} us, planted recovered {(idx.cpu().numpy()==gm).mean():.3f}")
model = sequence.SequenceModel(keys=k, pts=torch.from_numpy(pts).to(dev))
pixd = torch.from_numpy(pix).to(dev)
r = sequence.register_image(model, q, pixd, K, itr=500, reperr=2.0, seed=1); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): r = sequence.register_image(model, q, pixd, K, itr=500, reperr=2.0, seed=i)
torch.cuda.synchronize()
print(f"whole per-image chain (getCors f32 + filter + PnP-RANSAC 500 + refit): {(time.perf_counter()-t0)/20*1e3:.3f} ms/image; status {int(r.status.item())}, rot err {synth.rot_angle(r.pose.cpu().numpy()[:, :3], R[0]):.2e} rad")

# the same chain captured once in a HIP graph (no host launches on replay)
try:
    qs, ps = q.clone(), pixd.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            sequence.register_image(model, qs, ps, K, itr=500, reperr=2.0, seed=1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rg = sequence.register_image(model, qs, ps, K, itr=500, reperr=2.0, seed=1)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): g.replay()
    torch.cuda.synchronize()
    ref = sequence.register_image(model, q, pixd, K, itr=500, reperr=2.0, seed=1)
    torch.cuda.synchronize()
    print(f"the chain as one HIP graph replay: {(time.perf_counter()-t0)/20*1e3:.3f} ms/image; pose identical to eager: {torch.equal(rg.pose, ref.pose)}")
except Exception as e:  # report, do not hide
    print("HIP graph capture failed:", repr(e))
