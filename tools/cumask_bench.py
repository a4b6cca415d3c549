"""Experiment: bench.py with the side streams (per-group chains, verification) confined to a subset of the CUs
(hipExtStreamCreateWithCUMask): SIDE_CUS=<n> [SIDE_SPREAD=1] python tools/cumask_bench.py <bench args>."""
import ctypes, os, runpy, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
orig = torch.cuda.Stream
N = int(os.environ.get("SIDE_CUS", "0"))
spread = os.environ.get("SIDE_SPREAD", "0") == "1"
keep = []
def make(device=None, priority=0, **kw):
    if priority != -1 or N <= 0:
        return orig(device=device, priority=priority, **kw)
    words = (ctypes.c_uint32 * 8)()
    bits = [i * (256 // N) for i in range(N)] if spread else list(range(N))
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    keep.append(s)
    return torch.cuda.ExternalStream(s.value, device=device)
torch.cuda.Stream = make
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
