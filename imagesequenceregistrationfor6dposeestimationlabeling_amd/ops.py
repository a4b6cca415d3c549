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
