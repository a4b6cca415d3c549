"""K1 screened route alone, for rocprofv3 --kernel-trace --stats: python tools/r05_sparse_prof.py [mode] [P]"""
import sys
import torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 5
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4915200
N, D = 20000, 64
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
K = torch.randn(N, D, device=dev, generator=g)
K = 8.0 * K / K.norm(dim=1, keepdim=True)
gt = torch.randint(N, (P,), device=dev, generator=g)
Q = K[gt] + 0.35 * torch.randn(P, D, device=dev, generator=g)
q, k = ops.prescale_queries_log2(Q), K.bfloat16()
if True:
    for _ in range(12):
        ops.corr_argmax(q, k, want_lse=True, log2_prescaled=True, screened=(mode == 5))
    torch.cuda.synchronize()
print("done")
