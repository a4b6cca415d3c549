"""Quick K1 timing on the GPU box: python tools/time_corr.py [P N D]."""
import sys
import torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops

P, N, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (307200, 20000, 64)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
K = torch.randn(N, D, device=dev, generator=g)
K = 8.0 * K / K.norm(dim=1, keepdim=True)
gt = torch.randint(N, (P,), device=dev, generator=g)
Q = K[gt] + 0.35 * torch.randn(P, D, device=dev, generator=g)
Kb = K.bfloat16()
cases = [("planted bf16", Q.bfloat16(), False), ("planted bf16-log2", ops.prescale_queries_log2(Q), True),
         ("random bf16", torch.randn(P, D, device=dev, generator=g).bfloat16(), False),
         ("random bf16-log2", ops.prescale_queries_log2(torch.randn(P, D, device=dev, generator=g)), True),
         ("planted(0.6x) bf16-log2", ops.prescale_queries_log2(0.6 * Q), True)]
for name, q, l2 in cases:
    idx, logp = ops.corr_argmax(q, Kb, log2_prescaled=l2)
    torch.cuda.synchronize()
    if name.startswith("planted"):
        print(name, "recovered planted:", (idx.long() == gt).float().mean().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.corr_argmax(q, Kb, log2_prescaled=l2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: P={P} N={N} D={D}  {ms:.3f} ms  {2.0*P*N*D/ms*1e-9:.1f} TFLOP/s  {P*N/ms*1e-9:.2f} Texp/s"
          f"  rechecked {ops.corr_recheck_count()} of {P}  shader clock {ops.corr_clock_mhz():.0f} MHz")
