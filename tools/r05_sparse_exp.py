"""Round-5: the screened K1 route (ISR_TUNE_K1_SKIP = 5) against the shipped kernel.
    python tools/r05_sparse_exp.py [P N]"""
import sys
import torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops

D = 64
dev = torch.device("cuda:0")


def data(P, N, seed, planted=True, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    K = torch.randn(N, D, device=dev, generator=g)
    K = 8.0 * K / K.norm(dim=1, keepdim=True)
    if planted:
        gt = torch.randint(N, (P,), device=dev, generator=g)
        Q = scale * (K[gt] + 0.35 * torch.randn(P, D, device=dev, generator=g))
    else:
        Q = scale * torch.randn(P, D, device=dev, generator=g)
    return ops.prescale_queries_log2(Q), K.bfloat16()


def run(q, k, mode, reps=0):
    if True:
        idx, logp, lse = ops.corr_argmax(q, k, want_lse=True, log2_prescaled=True, screened=(mode == 5))
        torch.cuda.synchronize()
        ms = 0.0
        if reps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.corr_argmax(q, k, want_lse=True, log2_prescaled=True, screened=(mode == 5))
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
        red = ops.corr_screen_redone()[0]
        rc = ops.corr_recheck_count()
        clk = ops.corr_clock_mhz()
    return idx, logp, lse, ms, red, rc, clk


# correctness on small and ragged shapes first
for (P, N, planted, scale) in [(300, 999, True, 1.0), (1000, 20000, True, 1.0), (2048, 4097, False, 1.0), (5000, 20000, False, 1.0),
                               (777, 300, True, 0.3), (4096, 33, True, 1.0), (70000, 20000, True, 1.0)]:
    q, k = data(P, N, P + N, planted, scale)
    a = run(q, k, 0)
    b = run(q, k, 5)
    tiles = ((P + 31) // 32) * ((N + 31) // 32)
    print(f"P={P} N={N} planted={planted} scale={scale}: idx equal {bool((a[0] == b[0]).all())}  max|dlogp| {(a[1] - b[1]).abs().max().item():.3g}"
          f"  max|dlse| {(a[2] - b[2]).abs().max().item():.3g}  redone {b[4]} of {tiles} tile items ({b[4] / tiles:.3f})  recheck {a[5]} / {b[5]}", flush=True)

P, N = (int(x) for x in sys.argv[1:3]) if len(sys.argv) > 2 else (4915200, 20000)
for name, planted in (("planted", True), ("random", False)):
    q, k = data(P, N, 0, planted)
    for rep in range(2):
        for mode in (0, 5):
            idx, logp, lse, ms, red, rc, clk = run(q, k, mode, reps=10)
            if mode == 0:
                ref = (idx.clone(), logp.clone(), lse.clone())
                extra = ""
            else:
                tiles = (P // 32) * ((N + 31) // 32)
                extra = (f"  idx equal {bool((idx == ref[0]).all())}  max|dlogp| {(logp - ref[1]).abs().max().item():.3g}  max|dlse| {(lse - ref[2]).abs().max().item():.3g}"
                         f"  redone {red / tiles:.4f} of the tile items")
            print(f"{name} mode={mode}: {ms:.3f} ms  {2.0*P*N*D/ms*1e-9:.1f} TFLOP/s  clock {clk:.0f} MHz  recheck {rc}{extra}", flush=True)
