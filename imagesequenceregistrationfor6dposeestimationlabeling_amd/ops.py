"""Tensor-level wrappers over the C ABI: allocate outputs / workspace with torch (plumbing only)
and enqueue the HIP kernels on torch's current stream.  No arithmetic happens here."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _capi
from ._capi import check, current_stream, lib, ptr, require_cuda

_workspaces: dict[tuple, torch.Tensor] = {}


def workspace(device: torch.device, nbytes: int, tag: str = "default") -> torch.Tensor:
    """A cached, grow-only scratch buffer per (device, stream, tag)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


def _f64c(t: torch.Tensor | None) -> torch.Tensor | None:
    return None if t is None else t.to(torch.float64).contiguous()


@dataclass
class NNResult:
    sum_d: torch.Tensor          # (B,) f64
    sum_d2: torch.Tensor         # (B,) f64
    n_in: torch.Tensor           # (B,) i32
    nn_idx: torch.Tensor | None  # (B, Nq) i32
    nn_d: torch.Tensor | None    # (B, Nq) f64
    cov: torch.Tensor | None     # (B, 16) f64


def nn_batched(qry: torch.Tensor, tgt: torch.Tensor, Tq: torch.Tensor | None = None,
               Tt: torch.Tensor | None = None, radius: float = -1.0, want_idx: bool = False,
               want_dist: bool = False, want_cov: bool = False) -> NNResult:
    """isr_nn_batched: qry (Nq,3) f32, tgt (Nt,3) f32, Tq/Tt (B,3,4) or (B,12) f64 or None."""
    dev = require_cuda(qry, tgt, Tq, Tt)
    qry, tgt, Tq, Tt = _f32c(qry), _f32c(tgt), _f64c(Tq), _f64c(Tt)
    if qry.ndim != 2 or qry.shape[1] != 3 or tgt.ndim != 2 or tgt.shape[1] != 3:
        raise ValueError(f"clouds must be (N,3): got {tuple(qry.shape)} and {tuple(tgt.shape)}")
    B = 1
    for T in (Tq, Tt):
        if T is not None:
            if T.numel() % 12:
                raise ValueError("transforms must be (B,3,4)")
            B = max(B, T.numel() // 12)
    for T in (Tq, Tt):
        if T is not None and T.numel() // 12 != B:
            raise ValueError("Tq and Tt must have the same batch size")
    Nq, Nt = qry.shape[0], tgt.shape[0]
    L = lib()
    sum_d = torch.empty(B, dtype=torch.float64, device=dev)
    sum_d2 = torch.empty(B, dtype=torch.float64, device=dev)
    n_in = torch.empty(B, dtype=torch.int32, device=dev)
    nn_idx = torch.empty((B, Nq), dtype=torch.int32, device=dev) if want_idx else None
    nn_d = torch.empty((B, Nq), dtype=torch.float64, device=dev) if want_dist else None
    cov = torch.empty((B, 16), dtype=torch.float64, device=dev) if want_cov else None
    nbytes = L.isr_nn_batched_workspace_bytes(Nq, Nt, B)
    ws = workspace(dev, nbytes, "nn")
    with torch.cuda.device(dev):
        rc = L.isr_nn_batched(ptr(qry), Nq, ptr(tgt), Nt, ptr(Tq), ptr(Tt), B, float(radius),
                              ptr(sum_d), ptr(sum_d2), ptr(n_in), ptr(nn_idx), ptr(nn_d), ptr(cov),
                              ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_nn_batched")
    return NNResult(sum_d, sum_d2, n_in, nn_idx, nn_d, cov)


def rel_pose_table(R: torch.Tensor, t: torch.Tensor, mode: int, i0: int = 0,
                   i1: int | None = None) -> torch.Tensor:
    """isr_rel_pose_table: rows [i0,i1) of the n x n relative-pose table as (rows, n, 3, 4) f64."""
    dev = require_cuda(R, t)
    R, t = _f64c(R).reshape(-1, 9), _f64c(t).reshape(-1, 3)
    n = R.shape[0]
    i1 = n if i1 is None else i1
    out = torch.empty((i1 - i0, n, 3, 4), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_rel_pose_table(ptr(R), ptr(t), n, i0, i1, mode, ptr(out), current_stream(dev))
    check(rc, "isr_rel_pose_table")
    return out


def _pad_cols(t: torch.Tensor, D: int) -> torch.Tensor:
    if t.shape[1] == D:
        return t.contiguous()
    out = torch.zeros((t.shape[0], D), dtype=t.dtype, device=t.device)
    out[:, : t.shape[1]] = t
    return out


def corr_argmax(queries: torch.Tensor, keys: torch.Tensor, want_lse: bool = False):
    """isr_corr_argmax.  queries (P,D), keys (N,D); bf16 tensors take the bf16 MFMA path, f32
    tensors the exact f32 MFMA path (f16/f64 are converted to f32).  Zero columns are appended
    where the kernel needs a padded D (exact: they add 0 to every logit).
    Returns idx (P,) i32, logp (P,) f32[, lse (P,) f32] on the device."""
    dev = require_cuda(queries, keys)
    if queries.ndim != 2 or keys.ndim != 2 or queries.shape[1] != keys.shape[1]:
        raise ValueError(f"queries {tuple(queries.shape)} / keys {tuple(keys.shape)} must be (P,D),(N,D)")
    P, D = queries.shape
    N = keys.shape[0]
    if P == 0 or N == 0:
        raise ValueError("empty queries or keys")
    if queries.dtype == torch.bfloat16 and keys.dtype == torch.bfloat16:
        dtype = _capi.DTYPE_BF16
        Dp = next((d for d in (16, 32, 64, 128) if d >= D), None)
        if Dp is None:
            raise ValueError(f"bf16 path supports D <= 128, got {D}")
    else:
        dtype = _capi.DTYPE_F32
        queries, keys = queries.to(torch.float32), keys.to(torch.float32)
        Dp = D
        if D > 64:
            raise ValueError(f"f32 path supports D <= 64, got {D}")
    q, k = _pad_cols(queries, Dp), _pad_cols(keys, Dp)
    idx = torch.empty(P, dtype=torch.int32, device=dev)
    logp = torch.empty(P, dtype=torch.float32, device=dev)
    lse = torch.empty(P, dtype=torch.float32, device=dev) if want_lse else None
    L = lib()
    nbytes = L.isr_corr_argmax_workspace_bytes(P, N, Dp, dtype)
    ws = workspace(dev, nbytes, "corr")
    with torch.cuda.device(dev):
        rc = L.isr_corr_argmax(ptr(q), ptr(k), P, N, Dp, Dp, Dp, dtype, ptr(idx), ptr(logp), ptr(lse),
                               ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_corr_argmax")
    return (idx, logp, lse) if want_lse else (idx, logp)
