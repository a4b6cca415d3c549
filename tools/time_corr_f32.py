"""K1 exact-f32 path alone: python tools/time_corr_f32.py [P N D [chain]]  (default: the crop-batch shape, 128 x 2 195 rows,
80 000 keys, D = 12).  chain = ISR_TUNE_K1_F32_CHAIN for the run (0 default = f16 planes, 1 f32-MFMA chain kernel, 2 bf16 planes,
4 round 3's 96-wide rows at D <= 16); without it every route that applies to D is timed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
P, N, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (280960, 80000, 12)
routes = [int(sys.argv[4])] if len(sys.argv) > 4 else ([0, 2, 4, 1] if D <= 16 else [0, 2, 1] if D <= 64 else [0, 1])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
K = torch.randn(N, D, device=dev, generator=g)
K = 6.0 * K / K.norm(dim=1, keepdim=True)
gt = torch.randint(N, (P,), device=dev, generator=g)
Q = K[gt] + 0.25 * (12.0 / max(D, 12)) ** 0.5 * torch.randn(P, D, device=dev, generator=g)
ref = None
for chain in routes:
    with ops.tuning(k1_f32_chain=chain):
        idx, logp = ops.corr_argmax(Q, K); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.corr_argmax(Q, K)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        rc = ops.corr_recheck_count_f32(D)
    same = "" if ref is None else f"  idx equal to the first route: {bool(torch.equal(idx, ref))}"
    ref = idx if ref is None else ref
    print(f"f32 exact (K1_F32_CHAIN={chain}): P={P} N={N} D={D}  {ms:.3f} ms  {2.0*P*N*D/ms*1e-9:.1f} TFLOP/s  {P*N/ms*1e-9:.2f} Texp/s  "
          f"rechecked {rc}  recovered {(idx.long()==gt).float().mean().item():.3f}{same}", flush=True)
