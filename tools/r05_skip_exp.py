"""Round-5 experiment: the tile-skip variant of K1 (ISR_TUNE_K1_SKIP) against the shipped kernel, bench data.
    python tools/r05_skip_exp.py [P N D]"""
import sys
import torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops

P, N, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4915200, 20000, 64)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
K = torch.randn(N, D, device=dev, generator=g)
K = 8.0 * K / K.norm(dim=1, keepdim=True)
gt = torch.randint(N, (P,), device=dev, generator=g)
Q = K[gt] + 0.35 * torch.randn(P, D, device=dev, generator=g)
Kb = K.bfloat16()
cases = [("planted", ops.prescale_queries_log2(Q)), ("random", ops.prescale_queries_log2(torch.randn(P, D, device=dev, generator=g)))]
for name, q in cases:
    ref = None
    for rep in range(2):
        for mode in (0, 3, 1, 2):
            with ops.tuning(k1_skip=mode):
                idx, logp = ops.corr_argmax(q, Kb, log2_prescaled=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.corr_argmax(q, Kb, log2_prescaled=True)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                clk = ops.corr_clock_mhz()
            if mode == 0:
                ref = (idx.clone(), logp.clone())
                extra = ""
            else:
                extra = f" idx equal {bool((idx == ref[0]).all())}  max|dlogp| {(logp - ref[1]).abs().max().item():.3g}"
            print(f"{name} skip={mode}: {ms:.3f} ms  {2.0*P*N*D/ms*1e-9:.1f} TFLOP/s  clock {clk:.0f} MHz{extra}", flush=True)
