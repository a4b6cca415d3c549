"""Randomised stress of the exact-f32 K1 routes (f16 planes, bf16 planes) against the f32-MFMA chain kernel: shapes, descriptor
scales over six decades, keys that are copies of other keys perturbed in their last bits (the margin test must send every such
query to the f32-chain recheck, and the recheck must pick the chain kernel's key), queries on top of those keys, zero rows, a few
elements beyond f16's range (the gate).  python tools/stress_corr_f32.py [cases] [seed]"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
bad = 0
worst_lp = 0.0
total_recheck = {0: 0, 2: 0}
for c in range(cases):
    D = int(rng.integers(1, 129))
    N = int(rng.choice([1, 7, 33, 500, 4097, 9000, 20000, 33000]))
    P = int(rng.choice([1, 64, 300, 2000, 9000, 20000]))
    scale_k = float(10.0 ** rng.uniform(-3, 2.3))
    scale_q = float(10.0 ** rng.uniform(-3, 1.0)) / max(scale_k, 1e-3) * float(rng.uniform(2.0, 60.0))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 31)))
    K = torch.randn(N, D, device=dev, generator=g) * scale_k
    # near copies: a tenth of the keys repeat an earlier key with its last bits changed
    if N > 10:
        m = N // 10
        src = torch.randint(0, N // 2, (m,), device=dev, generator=g)
        dst = torch.randint(N // 2, N, (m,), device=dev, generator=g)
        ulps = torch.randint(-3, 4, (m, D), device=dev, generator=g)
        K[dst] = K[src] * (1.0 + ulps.float() * 1.1920929e-07)      # the last bits of every element, finite by construction
    gt = torch.randint(N, (P,), device=dev, generator=g)
    Q = K[gt] / max(scale_k, 1e-30) * scale_q * (1.0 + 0.0 * gt[:, None]) + 0.1 * scale_q * torch.randn(P, D, device=dev, generator=g) * float(rng.uniform(0, 1))
    Q[:: max(2, P // 50)] = 0.0
    if c % 7 == 3 and P > 5:
        Q[5, 0] = 9.0e4                                    # beyond f16: the f16 route must fall through
    with ops.tuning(k1_f32_chain=1):
        ref = ops.corr_argmax(Q, K, want_lse=True)
    for route in ((0, 2) if D <= 64 else (0,)):
        with ops.tuning(k1_f32_chain=route):
            got = ops.corr_argmax(Q, K, want_lse=True)
            rc = ops.corr_recheck_count_f32(D)
        total_recheck[route] += rc if 0 <= rc <= P else 0      # (a gated f16 call leaves its counter unset)
        neq = int((got[0] != ref[0]).sum())
        fin = torch.isfinite(ref[1]) & torch.isfinite(got[1])
        dlp = float((got[1][fin] - ref[1][fin]).abs().max()) if bool(fin.any()) else 0.0
        sc = 1.0 + float(ref[2][torch.isfinite(ref[2])].abs().max()) if bool(torch.isfinite(ref[2]).any()) else 1.0
        worst_lp = max(worst_lp, dlp / sc)
        if neq or dlp > 3e-6 * sc:
            bad += 1
            print(f"case {c}: route {route} P={P} N={N} D={D} |k|~{scale_k:.3g} |q|~{scale_q:.3g}: {neq} indices differ, logp diff {dlp:.3e} (scale {sc:.3g}), rechecked {rc}", flush=True)
            w = int((got[1] - ref[1]).abs().nan_to_num(1e30).argmax())
            print(f"   worst row {w}: got idx {int(got[0][w])} logp {float(got[1][w]):.6g} lse {float(got[2][w]):.6g} | chain idx {int(ref[0][w])} logp {float(ref[1][w]):.6g} lse {float(ref[2][w]):.6g}"
                  f" | |q| {float(Q[w].norm()):.4g} max|k| {float(K.norm(dim=1).max()):.4g} | non-finite lse: got {int((~torch.isfinite(got[2])).sum())} chain {int((~torch.isfinite(ref[2])).sum())}")
            s64 = (Q[w].double() @ K.double().T)
            print(f"   f64: argmax {int(s64.argmax())} max logit {float(s64.max()):.6g} lse {float(torch.logsumexp(s64, 0)):.6g} logp {float(s64.max() - torch.logsumexp(s64, 0)):.6g}")
print(f"{cases} cases, seed {seed}: {bad} failures; worst |logp - chain| / (1 + max|lse|) = {worst_lp:.3e}; queries sent to the f32-chain recheck: "
      f"f16 planes {total_recheck[0]}, bf16 planes {total_recheck[2]}")
sys.exit(1 if bad else 0)
