"""Randomised stress of ISR_DTYPE_BF16_LOG2_SCREENED against the unscreened log2-domain kernel (round 5).
    python tools/stress_corr_screened.py [seed] [cases]
Per case: random P, N (ragged), descriptor norms from flat to extremely peaked, a random mix of planted / unplanted / zero / huge /
tiny / duplicated rows, near-copy keys; checks
  * indices torch.equal to the unscreened kernel's (both are the exact arg-max);
  * |d logp|, |d lse| <= 2e-6 + 3e-7 (|lse| + |logp|)  (the left-out pieces: < 5e-7 of a sum; f32 outputs, logp = max - lse);
  * a random slice of the queries recomputed in a launch of its own: torch.equal (a result depends on (query, keys) only)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
D = 64
worst = [0.0, 0.0]
handed_total = redone_total = 0
for c in range(cases):
    P = int(rng.integers(1, 40000))
    N = int(rng.choice([rng.integers(1, 300), rng.integers(300, 5000), rng.integers(5000, 60000)]))
    tau = float(rng.choice([0.5, 2.0, 5.0, 8.0, 12.0, 20.0]))
    K = rng.normal(0, 1, (N, D)).astype(np.float32)
    K *= tau / np.linalg.norm(K, axis=1, keepdims=True)
    if N > 10 and rng.random() < 0.5:                       # near copies and exact duplicates of other keys
        j = rng.integers(N, size=N // 10); i = rng.integers(N, size=N // 10)
        K[j] = K[i] * (1.0 + rng.choice([0.0, 1e-3, 1e-2], size=(N // 10, 1)))
    gt = rng.integers(N, size=P)
    Q = (K[gt] + float(rng.choice([0.05, 0.35, 1.0])) * rng.normal(0, 1, (P, D))).astype(np.float32)
    kind = rng.random(P)
    Q[kind < 0.15] = rng.normal(0, 1, (int((kind < 0.15).sum()), D))            # unplanted
    Q[(kind > 0.15) & (kind < 0.20)] = 0.0                                        # padding rows
    Q[(kind > 0.20) & (kind < 0.23)] *= 30.0                                      # far outside the direct sum's range
    Q[(kind > 0.23) & (kind < 0.26)] *= 1e-3
    if rng.random() < 0.3:
        Q[: P // 2] = 0.0                                                         # whole workgroups of padding
    qb = ops.prescale_queries_log2(torch.from_numpy(Q)).to(dev)
    kb = torch.from_numpy(K).bfloat16().to(dev)
    a = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True)
    b = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True, screened=True)
    red, handed = ops.corr_screen_redone()
    redone_total += red; handed_total += handed
    assert torch.equal(a[0], b[0]), f"case {c}: {(a[0] != b[0]).sum().item()} index mismatches (P={P} N={N} tau={tau})"
    for k, name in ((1, "logp"), (2, "lse")):
        d = (a[k] - b[k]).abs()
        lim = 2e-6 + 3e-7 * (a[k].abs() + a[2].abs())       # logp = max - lse in f32: its spacing is lse's
        assert bool((d <= lim).all()), f"case {c}: {name} differs by {d.max().item():.3g} (P={P} N={N} tau={tau})"
        worst[k - 1] = max(worst[k - 1], float(d.max()))
    lo = int(rng.integers(0, P)); hi = min(P, lo + int(rng.integers(1, 3000)))
    s = ops.corr_argmax(qb[lo:hi].contiguous(), kb, want_lse=True, log2_prescaled=True, screened=True)
    assert all(torch.equal(x[lo:hi], y) for x, y in zip(b, s)), f"case {c}: slice [{lo}, {hi}) differs (P={P} N={N} tau={tau})"
print(f"seed {seed}: {cases} cases ok; worst |d logp| {worst[0]:.3g}, worst |d lse| {worst[1]:.3g}; "
      f"{redone_total} tile items fetched and redone, {handed_total} blocks handed to the dense kernel")
