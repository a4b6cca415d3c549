"""logp / lse of the exact-f32 K1 routes against an f64 log-softmax on the bench's data (one 640 x 480 x 64-D image, 20 000 keys):
python tools/diag_f32_routes_accuracy.py"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, synth
dev = torch.device("cuda:0")
P, N, D = 307200, 20000, 64
keys_f32, pts, *_ = bench.make_model(dev, N, D, 5.0)
R, t = synth.random_poses(np.random.default_rng(99), 1)
Q, pix = bench.make_image(dev, keys_f32, pts, synth.camera(), R[0], t[0], P, 0, True)
q = Q.float() / ops.LOG2E
rows = torch.arange(0, P, 97, device=dev)[:2048]
ref = torch.log_softmax(q[rows].double() @ keys_f32.double().T, dim=-1)
rv, ri = ref.max(dim=1)
rl = torch.logsumexp(q[rows].double() @ keys_f32.double().T, dim=-1)
for chain in (0, 2, 1):
    with ops.tuning(k1_f32_chain=chain):
        idx, logp, lse = ops.corr_argmax(q, keys_f32, want_lse=True)
    dl = (logp[rows].double() - rv).abs()
    ds = (lse[rows].double() - rl).abs()
    w = int(dl.argmax())
    print(f"K1_F32_CHAIN={chain}: idx equal f64 argmax {int((idx[rows].long() == ri).sum())}/{len(rows)}  max |logp - f64| {float(dl.max()):.3e} (row {int(rows[w])}: logp {float(logp[rows][w]):.6f} f64 {float(rv[w]):.6f})  "
          f"mean {float(dl.mean()):.3e}  max |lse - f64| {float(ds.max()):.3e}  max |logp| {float(rv.abs().max()):.3f}", flush=True)

# the rows on which the routes differ most from the chain kernel, against f64
outs = {}
for chain in (0, 2, 1):
    with ops.tuning(k1_f32_chain=chain):
        outs[chain] = ops.corr_argmax(q, keys_f32, want_lse=True)
for chain in (0, 2):
    d = (outs[chain][1] - outs[1][1]).abs()
    worst = torch.topk(d, 5).indices
    ref = torch.log_softmax(q[worst].double() @ keys_f32.double().T, dim=-1).max(dim=1).values
    print(f"K1_F32_CHAIN={chain} vs 1: max |logp diff| {float(d.max()):.3e}; worst rows {worst.tolist()}")
    for j, w in enumerate(worst.tolist()):
        print(f"   row {w}: route {chain} logp {float(outs[chain][1][w]):.7f}  chain kernel {float(outs[1][1][w]):.7f}  f64 {float(ref[j]):.7f}  |q| {float(q[w].norm()):.3f}")

# determinism of the plane routes, and the bench's own measurement function on the same data
for chain in (0, 2):
    with ops.tuning(k1_f32_chain=chain):
        a = ops.corr_argmax(q, keys_f32, want_lse=True)
        same = True
        for _ in range(6):
            b = ops.corr_argmax(q, keys_f32, want_lse=True)
            same &= all(torch.equal(x, y) for x, y in zip(a, b))
    print(f"K1_F32_CHAIN={chain}: seven calls bit-identical: {same}")
print(bench.measure_k1_f32(Q, keys_f32, dev, True))
g = Q[None].expand(8, -1, -1).reshape(-1, D).contiguous()
kb = keys_f32.bfloat16().contiguous()
for _ in range(3):
    ops.corr_argmax(g, kb, log2_prescaled=True)
print(bench.measure_k1_f32(Q, keys_f32, dev, True))
