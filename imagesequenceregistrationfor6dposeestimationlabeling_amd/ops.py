"""Tensor-level wrappers over the C ABI: allocate outputs / workspace with torch (plumbing only)
and enqueue the HIP kernels on torch's current stream.  No arithmetic happens here."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

import torch

from . import _capi
from ._capi import check, current_stream, lib, ptr, require_cuda

_workspaces: dict[tuple, torch.Tensor] = {}

# Optional per-entry-point HIP-event timing (bench.py's live roofline measurement): when enabled,
# every wrapper brackets its C-ABI call with events on the stream it launches on and records
# (start, end, algorithmic work).  Off by default: no events, no overhead.
_timers: dict[str, list] | None = None


def enable_timing(on: bool = True) -> None:
    global _timers
    _timers = {} if on else None


def drain_timing() -> dict[str, tuple[int, float, float]]:
    """-> {name: (calls, total_ms, total_work)}; synchronises.  Clears the records."""
    out = {}
    if _timers is None:
        return out
    torch.cuda.synchronize()
    for name, recs in _timers.items():
        ms = sum(a.elapsed_time(b) for a, b, _ in recs)
        out[name] = (len(recs), ms, float(sum(w for _, _, w in recs)))
    _timers.clear()
    return out


class _timed:
    def __init__(self, name: str, work: float):
        self.name, self.work = name, work

    def __enter__(self):
        if _timers is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _timers is not None:
            self.e1.record()
            _timers.setdefault(self.name, []).append((self.e0, self.e1, self.work))
        return False


class tuning:
    """Context manager over the library's tuning knobs (isr_tuning_set; experiments and tests only):
        with ops.tuning(nn_path=0): ...      # brute-force NN inside the block
    Knobs: nn_path (-1 auto, 0 brute, 1 per-lane grid, 2 block-cooperative grid), nn_filter (-1 auto, 0, 1),
    icp_warm (1, 0), nn_plan_rq, nn_plan_blocks, nn_tile_st / nn_tile_sq (x 1000), nn_tile_tb.  Process-wide:
    do not flip a knob while another thread is inside an entry point that reads it."""

    def __init__(self, **knobs):
        self.knobs = {_capi.TUNE[k]: int(v) for k, v in knobs.items()}

    def __enter__(self):
        L = lib()
        self.old = {k: L.isr_tuning_get(k) for k in self.knobs}
        for k, v in self.knobs.items():
            check(L.isr_tuning_set(k, v), "isr_tuning_set")
        return self

    def __exit__(self, *exc):
        L = lib()
        for k, v in self.old.items():
            L.isr_tuning_set(k, v)
        return False


def set_tuning(**knobs) -> None:
    """Set tuning knobs for the rest of the process (tools/ sweeps); see `tuning`."""
    for k, v in knobs.items():
        check(lib().isr_tuning_set(_capi.TUNE[k], int(v)), "isr_tuning_set")


def set_tile_plan(plan: str | None) -> None:
    """'st,sq,tb' (the old ISR_NN_TILE syntax) or None for the defaults."""
    if plan is None:
        set_tuning(nn_tile_st=0, nn_tile_sq=0, nn_tile_tb=0)
    else:
        st, sq, tb = plan.split(",")
        set_tuning(nn_tile_st=round(float(st) * 1000), nn_tile_sq=round(float(sq) * 1000), nn_tile_tb=int(tb))


def workspace(device: torch.device, nbytes: int, tag: str = "default") -> torch.Tensor:
    """A cached, grow-only scratch buffer per (device, stream, tag).
    While the current stream is being captured into a HIP graph nothing is cached: a buffer allocated during
    capture lives in the graph's private pool, and handing it to later eager calls (or to another capture)
    would share that memory with no ordering — a capture gets a fresh buffer per call, which the graph owns.
    Pre-warm (run the call once eagerly) if the capture should reuse the cached buffer instead."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is not None and buf.numel() >= nbytes:
        return buf
    fresh = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    if not torch.cuda.is_current_stream_capturing():
        _workspaces[key] = fresh
    return fresh


def clear_workspaces() -> None:
    """Drop every cached scratch buffer (e.g. after destroying streams: stream handles are reused as keys)."""
    _workspaces.clear()


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
    with torch.cuda.device(dev), _timed("nn_batched", float(B) * Nq * Nt):
        rc = L.isr_nn_batched(ptr(qry), Nq, ptr(tgt), Nt, ptr(Tq), ptr(Tt), B, float(radius),
                              ptr(sum_d), ptr(sum_d2), ptr(n_in), ptr(nn_idx), ptr(nn_d), ptr(cov),
                              ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_nn_batched")
    return NNResult(sum_d, sum_d2, n_in, nn_idx, nn_d, cov)


@dataclass
class DistField:
    """Distances from the cell centres of a uniform grid to a (static) cloud: what isr_adds_bounds reads."""
    field: torch.Tensor       # (nz, ny, nx) f32 device
    grid_min: np.ndarray      # (3,) f64 host: the corner of cell (0, 0, 0)
    h: float                  # cell edge
    dims: tuple               # (nx, ny, nz)
    bbox: np.ndarray          # (6,) f32 host: the cloud's bounding box, lo xyz then hi xyz
    n_points: int


def dist_field(cloud: torch.Tensor, cells: int = 128, margin: float | None = None) -> DistField:
    """Exact distances from the centres of a cells^3-ish grid around `cloud` (N,3) to the cloud, through isr_nn_batched.
    margin: how far beyond the cloud's bounding box the grid reaches (default: a quarter of its longest extent)."""
    dev = require_cuda(cloud)
    c = _f32c(cloud)
    lo, hi = c.min(dim=0).values.double().cpu().numpy(), c.max(dim=0).values.double().cpu().numpy()
    ext = hi - lo
    m = float(0.25 * ext.max()) if margin is None else float(margin)
    h = float((ext.max() + 2.0 * m) / cells)
    gmin = lo - m
    dims = tuple(int(np.ceil((ext[a] + 2.0 * m) / h)) for a in range(3))
    ax = [gmin[a] + h * (torch.arange(dims[a], device=dev, dtype=torch.float64) + 0.5) for a in range(3)]
    zz, yy, xx = torch.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    centres = torch.stack([xx, yy, zz], dim=-1).reshape(-1, 3).to(torch.float32).contiguous()
    # the centres as f32 are what the field is exact for: their rounding (<= 4e-6 mm at object scale) sits inside the slack
    # every user of the bounds keeps around the threshold
    parts = [nn_batched(centres[s0:s0 + (1 << 20)], c, want_dist=True).nn_d[0] for s0 in range(0, centres.shape[0], 1 << 20)]
    field = torch.cat(parts).to(torch.float32).reshape(dims[2], dims[1], dims[0]).contiguous()
    return DistField(field, np.asarray(gmin, np.float64), h, dims, np.concatenate([lo, hi]).astype(np.float32), int(c.shape[0]))


def adds_bounds(verts: torch.Tensor, Tq: torch.Tensor, Tt: torch.Tensor | None, fld: DistField):
    """isr_adds_bounds: per batch item b bounds of sum_v dist(Tt[b]^-1 Tq[b] v, cloud) — isr_nn_batched's sum_d for queries
    `verts` against the field's cloud — as (lb_sum, ub_sum) f64 device tensors (wider, still finite, for vertices off the grid)."""
    import ctypes
    dev = require_cuda(verts, Tq, Tt, fld.field)
    v, Tq, Tt = _f32c(verts), _f64c(Tq), _f64c(Tt)
    if v.ndim != 2 or v.shape[1] != 3 or Tq.numel() % 12 or (Tt is not None and Tt.numel() != Tq.numel()):
        raise ValueError(f"adds_bounds: verts {tuple(v.shape)}, Tq {tuple(Tq.shape)}, Tt {None if Tt is None else tuple(Tt.shape)}")
    B = Tq.numel() // 12
    lb = torch.empty(B, dtype=torch.float64, device=dev)
    ub = torch.empty(B, dtype=torch.float64, device=dev)
    if B == 0:
        return lb, ub
    gmin = (ctypes.c_double * 3)(*[float(x) for x in fld.grid_min])
    bbox = (ctypes.c_float * 6)(*[float(x) for x in fld.bbox])
    with torch.cuda.device(dev), _timed("adds_bounds", float(B) * v.shape[0]):
        rc = lib().isr_adds_bounds(ptr(v), v.shape[0], ptr(Tq), ptr(Tt), B, ptr(fld.field), ctypes.cast(gmin, ctypes.c_void_p),
                                   float(fld.h), fld.dims[0], fld.dims[1], fld.dims[2], ctypes.cast(bbox, ctypes.c_void_p),
                                   ptr(lb), ptr(ub), current_stream(dev))
    check(rc, "isr_adds_bounds")
    return lb, ub


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


LOG2E = 1.4426950408889634


def prescale_queries_log2(queries: torch.Tensor) -> torch.Tensor:
    """f32 descriptors -> bf16(queries * log2 e): the input of corr_argmax(..., log2_prescaled=True).
    One rounding, as for a plain .bfloat16(); a conversion, not part of the measured path."""
    return (queries.to(torch.float32) * LOG2E).to(torch.bfloat16)


@dataclass
class CorrCall:
    """An isr_corr_argmax call whose opening half has been enqueued (corr_argmax_open): what corr_argmax_close needs."""
    q: torch.Tensor
    k: torch.Tensor
    P: int
    N: int
    Dp: int
    dtype: int
    idx: torch.Tensor
    logp: torch.Tensor
    lse: torch.Tensor | None
    rows_per_image: int
    n_rows: torch.Tensor | None
    hist: torch.Tensor | None
    ws: torch.Tensor
    dev: torch.device

    def outputs(self):
        out = (self.idx, self.logp) if self.lse is None else (self.idx, self.logp, self.lse)
        return out if self.hist is None else out + (self.hist,)


def _corr_prepare(queries, keys, want_lse, log2_prescaled, screened, rows_per_image, n_rows, ws_tag) -> CorrCall:
    dev = require_cuda(queries, keys, n_rows)
    if queries.ndim != 2 or keys.ndim != 2 or queries.shape[1] != keys.shape[1]:
        raise ValueError(f"queries {tuple(queries.shape)} / keys {tuple(keys.shape)} must be (P,D),(N,D)")
    P, D = queries.shape
    N = keys.shape[0]
    if P == 0 or N == 0:
        raise ValueError("empty queries or keys")
    if screened and not log2_prescaled:
        raise ValueError("screened needs log2-prescaled bf16 queries")
    if queries.dtype == torch.bfloat16 and keys.dtype == torch.bfloat16:
        dtype = (_capi.DTYPE_BF16_LOG2_SCREENED if screened else _capi.DTYPE_BF16_LOG2) if log2_prescaled else _capi.DTYPE_BF16
        Dp = next((d for d in (16, 32, 64, 128) if d >= D), None)
        if Dp is None:
            raise ValueError(f"bf16 path supports D <= 128, got {D}")
    else:
        if log2_prescaled:
            raise ValueError("log2_prescaled needs bf16 queries and keys")
        dtype = _capi.DTYPE_F32
        queries, keys = queries.to(torch.float32), keys.to(torch.float32)
        Dp = D
        if D > 128:
            raise ValueError(f"f32 path supports D <= 128, got {D}")
    q, k = _pad_cols(queries, Dp), _pad_cols(keys, Dp)
    idx = torch.empty(P, dtype=torch.int32, device=dev)
    logp = torch.empty(P, dtype=torch.float32, device=dev)
    lse = torch.empty(P, dtype=torch.float32, device=dev) if want_lse else None
    ws = workspace(dev, lib().isr_corr_argmax_workspace_bytes(P, N, Dp, dtype), ws_tag)
    hist = None
    if rows_per_image is not None:
        if rows_per_image <= 0 or P % rows_per_image:
            raise ValueError(f"corr_argmax: P={P} is not a whole number of images of {rows_per_image} rows")
        if n_rows is not None and (n_rows.dtype != torch.int32 or n_rows.numel() != P // rows_per_image):
            raise ValueError("corr_argmax: n_rows must be (images,) int32")
        hist = torch.empty((P // rows_per_image, 2048), dtype=torch.int32, device=dev)
    return CorrCall(q, k, P, N, Dp, dtype, idx, logp, lse, int(rows_per_image or 1), n_rows, hist, ws, dev)


def _corr_launch(c: CorrCall, phase: int, what: str, work: float) -> None:
    L = lib()
    with torch.cuda.device(c.dev), _timed(what, work):
        if phase == 3 and c.hist is None:
            rc = L.isr_corr_argmax(ptr(c.q), ptr(c.k), c.P, c.N, c.Dp, c.Dp, c.Dp, c.dtype, ptr(c.idx), ptr(c.logp), ptr(c.lse),
                                   ptr(c.ws), c.ws.numel(), current_stream(c.dev))
        elif phase == 3:
            rc = L.isr_corr_argmax_digits(ptr(c.q), ptr(c.k), c.P, c.N, c.Dp, c.Dp, c.Dp, c.dtype, ptr(c.idx), ptr(c.logp), ptr(c.lse),
                                          c.rows_per_image, ptr(c.n_rows), ptr(c.hist), ptr(c.ws), c.ws.numel(), current_stream(c.dev))
        else:
            rc = L.isr_corr_argmax_phase(ptr(c.q), ptr(c.k), c.P, c.N, c.Dp, c.Dp, c.Dp, c.dtype, ptr(c.idx), ptr(c.logp), ptr(c.lse),
                                         c.rows_per_image, ptr(c.n_rows), ptr(c.hist), phase, ptr(c.ws), c.ws.numel(),
                                         current_stream(c.dev))
    check(rc, "isr_corr_argmax")
    global _last_corr
    _last_corr = (c.ws, c.P, c.N, c.dtype, c.dev)


def corr_argmax(queries: torch.Tensor, keys: torch.Tensor, want_lse: bool = False,
                log2_prescaled: bool = False, screened: bool = False, rows_per_image: int | None = None,
                n_rows: torch.Tensor | None = None):
    """isr_corr_argmax.  queries (P,D), keys (N,D); bf16 tensors take the bf16 MFMA path, f32
    tensors the exact f32 MFMA path (f16/f64 are converted to f32).  Zero columns are appended
    where the kernel needs a padded D (exact: they add 0 to every logit).
    log2_prescaled: the bf16 queries already carry a factor log2(e) (prescale_queries_log2): the
    kernel works in log2 units with the -M2 reference folded into the MFMA contraction; outputs are
    still natural-log.
    screened (with log2_prescaled): ISR_DTYPE_BF16_LOG2_SCREENED — the rows also go through a block-scaled FP6 screen, and
    pieces of the log-sum-exp proven to lie more than T = 21 + ceil(log2 N) log2 units below the query's maximum are never
    formed (indices stay exact, lse moves by < 5e-7; D = 64 only, other shapes run unscreened).  For peaked softmaxes.
    rows_per_image: isr_corr_argmax_digits — the rows are P / rows_per_image images whose top-80 % cut follows; the call also
    returns digit_hist (images, 2048) i32, the first histogram of that cut's radix select over logp (of image b's first
    n_rows[b] rows; n_rows None: all), formed where logp is written: pass it to select_top_batch(..., digit_hist=...).
    Returns idx (P,) i32, logp (P,) f32[, lse (P,) f32][, digit_hist] on the device."""
    c = _corr_prepare(queries, keys, want_lse, log2_prescaled, screened, rows_per_image, n_rows, "corr")
    _corr_launch(c, 3, "corr_argmax", 2.0 * c.P * c.N * c.Dp)
    return c.outputs()


def corr_argmax_open(queries: torch.Tensor, keys: torch.Tensor, want_lse: bool = False, log2_prescaled: bool = False,
                     screened: bool = False, rows_per_image: int | None = None, n_rows: torch.Tensor | None = None,
                     ws_tag: str = "corr") -> CorrCall:
    """isr_corr_argmax_phase, phase 1: pre-processing, key norms and the chip-filling kernel(s) of a corr_argmax call, enqueued
    on the current stream.  corr_argmax_close(call) enqueues the closing kernels (fallback, finalize, recheck, merge) — on
    whatever stream is current then, ordered behind this one by the caller (an event) — and returns corr_argmax's outputs.
    The workspace (cached per stream and ws_tag) must not be opened again before its close has finished: alternate two tags."""
    c = _corr_prepare(queries, keys, want_lse, log2_prescaled, screened, rows_per_image, n_rows, ws_tag)
    _corr_launch(c, 1, "corr_argmax", 2.0 * c.P * c.N * c.Dp)
    return c


def corr_argmax_close(call: CorrCall):
    """isr_corr_argmax_phase, phase 2, for a call opened by corr_argmax_open: the outputs are complete when it has run."""
    _corr_launch(call, 2, "corr_close", 0.0)
    return call.outputs()


def corr_lse(queries: torch.Tensor, keys: torch.Tensor, log2_prescaled: bool = False, screened: bool = False) -> torch.Tensor:
    """The row log-sum-exps of queries @ keys.T alone — pose_refine.py:56's denominator image, estimate_pose's row sums
    (poseEstSurf.py:68-71) — as an lse-only call of isr_corr_argmax (idx = logp = NULL): no maxima are tracked, no index is
    certified, and the values are the bits corr_argmax(..., want_lse=True) returns."""
    dev = require_cuda(queries, keys)
    if queries.ndim != 2 or keys.ndim != 2 or queries.shape[1] != keys.shape[1]:
        raise ValueError(f"queries {tuple(queries.shape)} / keys {tuple(keys.shape)} must be (P,D),(N,D)")
    P, D = queries.shape
    N = keys.shape[0]
    if P == 0 or N == 0:
        raise ValueError("empty queries or keys")
    if queries.dtype == torch.bfloat16 and keys.dtype == torch.bfloat16:
        dtype = (_capi.DTYPE_BF16_LOG2_SCREENED if screened else _capi.DTYPE_BF16_LOG2) if log2_prescaled else _capi.DTYPE_BF16
        Dp = next((d for d in (16, 32, 64, 128) if d >= D), None)
        if Dp is None:
            raise ValueError(f"bf16 path supports D <= 128, got {D}")
    else:
        if log2_prescaled:
            raise ValueError("log2_prescaled needs bf16 queries and keys")
        dtype = _capi.DTYPE_F32
        queries, keys = queries.to(torch.float32), keys.to(torch.float32)
        Dp = D
        if D > 128:
            raise ValueError(f"f32 path supports D <= 128, got {D}")
    q, k = _pad_cols(queries, Dp), _pad_cols(keys, Dp)
    lse = torch.empty(P, dtype=torch.float32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_corr_argmax_workspace_bytes(P, N, Dp, dtype), "corr")
    with torch.cuda.device(dev), _timed("corr_lse", 2.0 * P * N * Dp):
        rc = L.isr_corr_argmax(ptr(q), ptr(k), P, N, Dp, Dp, Dp, dtype, None, None, ptr(lse), ptr(ws), ws.numel(),
                               current_stream(dev))
    check(rc, "isr_corr_argmax (lse only)")
    return lse


def corr_topk(queries: torch.Tensor, keys: torch.Tensor, k: int):
    """getCors with leaves = k > 1 (inference.py:145-149): per query the k largest entries of log_softmax(queries @ keys.T)
    and their keys — isr_corr_topk behind an lse-only K1 call; the (P, N) matrix is never formed.  Any float dtype (bf16 /
    f16 rows are widened exactly); 1 <= k <= 8.  Returns idx (P, k) int32, vals (P, k) f32 on the device, descending,
    equal values by ascending key."""
    dev = require_cuda(queries, keys)
    if queries.ndim != 2 or keys.ndim != 2 or queries.shape[1] != keys.shape[1]:
        raise ValueError(f"queries {tuple(queries.shape)} / keys {tuple(keys.shape)} must be (P,D),(N,D)")
    if not 1 <= int(k) <= 8:
        raise ValueError(f"leaves = {k}: 1 .. 8 are supported without materialising the matrix")
    q, kk = _f32c(queries), _f32c(keys)
    P, D = q.shape
    N = kk.shape[0]
    if P == 0 or N == 0:
        raise ValueError("empty queries or keys")
    if D > 128:
        raise ValueError(f"D <= 128, got {D}")
    lse = corr_lse(q, kk)
    idx = torch.empty((P, int(k)), dtype=torch.int32, device=dev)
    vals = torch.empty((P, int(k)), dtype=torch.float32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_corr_topk_workspace_bytes(P, N), "corr_topk")
    with torch.cuda.device(dev):
        rc = L.isr_corr_topk(ptr(q), ptr(kk), P, N, D, D, D, int(k), ptr(lse), ptr(idx), ptr(vals), ptr(ws), ws.numel(),
                             current_stream(dev))
    check(rc, "isr_corr_topk")
    return idx, vals


_last_corr = None


def corr_clock_mhz() -> float:
    """Diagnostics: the shader clock (MHz) the last bf16 corr_argmax launch held (0.0 for the f32 path)."""
    import ctypes
    if _last_corr is None:
        return 0.0
    ws, P, N, dtype, dev = _last_corr
    out = ctypes.c_double(0.0)
    with torch.cuda.device(dev):
        rc = lib().isr_corr_argmax_clock_mhz(ptr(ws), ws.numel(), P, N, dtype, ctypes.addressof(out), current_stream(dev))
    check(rc, "isr_corr_argmax_clock_mhz")
    return float(out.value)


def corr_recheck_count() -> int:
    """Diagnostics: how many queries of the last corr_argmax call were decided by the exact recheck
    (-1 on the f32 path).  Synchronises the current stream."""
    import ctypes
    if _last_corr is None:
        return -1
    ws, P, N, dtype, dev = _last_corr
    out = ctypes.c_int32(-1)
    with torch.cuda.device(dev):
        rc = lib().isr_corr_argmax_recheck_count(ptr(ws), ws.numel(), P, N, dtype, ctypes.addressof(out),
                                                 current_stream(dev))
    check(rc, "isr_corr_argmax_recheck_count")
    return int(out.value)


def corr_quantize_fp6(rows: torch.Tensor):
    """isr_corr_quantize_fp6 (parity hook of the screened K1 route): rows (R, 64) bf16 on the device ->
    (image (R, 64) uint8, norms (R, 2) f32 {|x|, |x - x~|}, maxima (2,) f32 {max |x - x~|^2, max |x~|^2})."""
    dev = require_cuda(rows)
    if rows.ndim != 2 or rows.shape[1] != 64 or rows.dtype != torch.bfloat16:
        raise ValueError(f"rows must be (R, 64) bf16, got {tuple(rows.shape)} {rows.dtype}")
    rows = rows.contiguous()
    R = rows.shape[0]
    out = torch.empty((R, 64), dtype=torch.uint8, device=dev)
    nrm = torch.empty((R, 2), dtype=torch.float32, device=dev)
    kmax = torch.empty(2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_corr_quantize_fp6(ptr(rows), R, 64, ptr(out), ptr(nrm), ptr(kmax), current_stream(dev))
    check(rc, "isr_corr_quantize_fp6")
    return out, nrm, kmax


def corr_screen_redone() -> tuple[int, int]:
    """Diagnostics of the last screened corr_argmax call: (tile items — 32 queries x 32 keys — fetched again and redone on the
    bf16 matrix cores behind the FP6 screen, 256-query blocks handed to the dense kernel); zeros on the unscreened routes.
    Synchronises the current stream."""
    import ctypes
    if _last_corr is None:
        return 0, 0
    ws, P, N, dtype, dev = _last_corr
    out = (ctypes.c_longlong * 2)(0, 0)
    with torch.cuda.device(dev):
        rc = lib().isr_corr_argmax_screen_redone(ptr(ws), ws.numel(), P, N, dtype, ctypes.addressof(out), current_stream(dev))
    check(rc, "isr_corr_argmax_screen_redone")
    return int(out[0]), int(out[1])


def corr_recheck_count_f32(D: int) -> int:
    """Diagnostics: the length of the f32-chain recheck list of the last f32 corr_argmax call with D columns, under the
    knob setting still in force (-1 when that call took the f32-MFMA chain kernel).  Synchronises the current stream."""
    import ctypes
    if _last_corr is None:
        return -1
    ws, P, N, dtype, dev = _last_corr
    if dtype != _capi.DTYPE_F32:
        return -1
    out = ctypes.c_int32(-1)
    with torch.cuda.device(dev):
        rc = lib().isr_corr_argmax_recheck_count_f32(ptr(ws), ws.numel(), P, N, int(D), ctypes.addressof(out),
                                                     current_stream(dev))
    check(rc, "isr_corr_argmax_recheck_count_f32")
    return int(out.value)


def select_top(logp: torch.Tensor, frac: float = 0.8, min_n: int = 500, n_dev: torch.Tensor | None = None):
    """isr_select_top: keep (P,) i32 (first M entries valid, ascending), M_dev (1,) i32, thr (1,) f32.
    n_dev: (1,) i32 on the device — only the first n_dev values are the input (isr_select_top_dev)."""
    dev = require_cuda(logp, n_dev)
    logp = _f32c(logp).reshape(-1)
    P = logp.numel()
    keep = torch.empty(P, dtype=torch.int32, device=dev)
    M_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    thr = torch.empty(1, dtype=torch.float32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_select_top_workspace_bytes(P), "select")
    with torch.cuda.device(dev), _timed("select_top", 4.0 * P):
        if n_dev is None:
            rc = L.isr_select_top(ptr(logp), P, float(frac), int(min_n), ptr(keep), ptr(M_dev), ptr(thr),
                                  ptr(ws), ws.numel(), current_stream(dev))
        else:
            rc = L.isr_select_top_dev(ptr(logp), P, ptr(n_dev), float(frac), int(min_n), ptr(keep), ptr(M_dev),
                                      ptr(thr), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_select_top")
    return keep, M_dev, thr


def prep_queries(feat: torch.Tensor, mask: torch.Tensor, c0: int = 0, D: int | None = None, step: int = 3,
                 dtype: str = "bf16_log2"):
    """isr_prep_queries: the network's channels-last feature map -> K1's query operand.
    feat (H, W, C) or (1, H, W, C) f32; mask (H, W) or (H, W, k) uint8 (channel 0 is used, as
    cropMask[:, :, 0]); dtype 'bf16' | 'bf16_log2' | 'f32'.
    Returns Q (S, Dpad) [S = capacity = ceil(H/step) * ceil(W/step)], pix_xy (S, 2) f32, n_dev (1,) i32."""
    dev = require_cuda(feat, mask)
    feat = feat.reshape(feat.shape[-3:]) if feat.ndim == 4 else feat
    feat = _f32c(feat)
    H, W, C = feat.shape
    D = C - c0 if D is None else D
    if mask.dtype != torch.uint8:
        mask = (mask != 0).to(torch.uint8)
    mask = mask.contiguous()
    stride = 1 if mask.ndim == 2 else mask.shape[2]
    if tuple(mask.shape[:2]) != (H, W):
        raise ValueError(f"mask {tuple(mask.shape)} does not match the feature map {(H, W)}")
    code = {"bf16": _capi.DTYPE_BF16, "bf16_log2": _capi.DTYPE_BF16_LOG2, "f32": _capi.DTYPE_F32}[dtype]
    Dpad = D if dtype == "f32" else (16 if D <= 16 else 32 if D <= 32 else 64 if D <= 64 else 128)
    if D > 128 or (dtype == "f32" and D > 64):
        raise ValueError(f"D={D} not supported by K1")
    S = ((H + step - 1) // step) * ((W + step - 1) // step)
    Q = torch.empty((S, Dpad), dtype=torch.float32 if dtype == "f32" else torch.bfloat16, device=dev)
    pix = torch.zeros((S, 2), dtype=torch.float32, device=dev)
    n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_prep_queries_workspace_bytes(H, W, step), "prep")
    with torch.cuda.device(dev):
        rc = L.isr_prep_queries(ptr(feat), H, W, C, int(c0), int(D), ptr(mask), int(stride), int(step), code, Dpad,
                                ptr(Q), ptr(pix), ptr(n_dev), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_prep_queries")
    return Q, pix, n_dev


def prep_queries_batch(feat: torch.Tensor, mask: torch.Tensor, c0: int = 0, D: int | None = None, step: int = 3,
                       dtype: str = "bf16_log2"):
    """isr_prep_queries_batch: a GROUP of crops in three launches.  feat (B, H, W, C) f32 channels-last; mask
    (B, H, W) or (B, H, W, k) uint8 (channel 0 is used).  Returns Q (B, S, Dpad) with zero rows past each image's
    count, pix_xy (B, S, 2) f32, n_dev (B,) i32 — S = ceil(H/step) * ceil(W/step)."""
    dev = require_cuda(feat, mask)
    if feat.ndim != 4:
        raise ValueError(f"prep_queries_batch: feat {tuple(feat.shape)} must be (B, H, W, C)")
    feat = _f32c(feat)
    B, H, W, C = feat.shape
    D = C - c0 if D is None else D
    if mask.dtype != torch.uint8:
        mask = (mask != 0).to(torch.uint8)
    mask = mask.contiguous()
    stride = 1 if mask.ndim == 3 else mask.shape[3]
    if tuple(mask.shape[:3]) != (B, H, W):
        raise ValueError(f"mask {tuple(mask.shape)} does not match the feature maps {(B, H, W)}")
    code = {"bf16": _capi.DTYPE_BF16, "bf16_log2": _capi.DTYPE_BF16_LOG2, "f32": _capi.DTYPE_F32}[dtype]
    Dpad = D if dtype == "f32" else (16 if D <= 16 else 32 if D <= 32 else 64 if D <= 64 else 128)
    if D > 128 or (dtype == "f32" and D > 64):
        raise ValueError(f"D={D} not supported by K1")
    S = ((H + step - 1) // step) * ((W + step - 1) // step)
    Q = torch.empty((B, S, Dpad), dtype=torch.float32 if dtype == "f32" else torch.bfloat16, device=dev)
    pix = torch.zeros((B, S, 2), dtype=torch.float32, device=dev)
    n_dev = torch.empty(B, dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_prep_queries_batch_workspace_bytes(H, W, step, B), "prep")
    with torch.cuda.device(dev):
        rc = L.isr_prep_queries_batch(ptr(feat), B, H, W, C, int(c0), int(D), ptr(mask), int(stride), int(step), code, Dpad,
                                      ptr(Q), ptr(pix), ptr(n_dev), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_prep_queries_batch")
    return Q, pix, n_dev


IMAGENET_MEAN = (0.485, 0.456, 0.406)      # normalize(), inference.py:135-141
IMAGENET_STD = (0.229, 0.224, 0.225)


def mask_bbox(mask: torch.Tensor) -> torch.Tensor:
    """isr_mask_bbox: mask (B, H, W[, C]) u8 on the device -> (B, 4) i32 {x, y, w, h} (cv2.boundingRect of channel 0)."""
    dev = require_cuda(mask)
    m = mask if mask.ndim == 4 else mask[..., None]
    m = m.contiguous()
    B, H, W, C = m.shape
    out = torch.empty((B, 4), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_mask_bbox(ptr(m), B, H, W, C, ptr(out), current_stream(dev))
    check(rc, "isr_mask_bbox")
    return out


def crop_normalize(rgb: torch.Tensor, mask: torch.Tensor, M, out_size: int = 224, use_mask: bool = True,
                   mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """isr_crop_normalize: rgb (B, H, W, 3) u8, mask (B, H, W[, C]) u8 on the device, M (B, 2, 3) host f64 (the
    reference's source -> crop affine) -> inputIM (B, 3, r, r) f32, cropMask (B, r, r) u8."""
    import ctypes
    import numpy as np
    dev = require_cuda(rgb, mask)
    rgb = rgb.contiguous()
    m = (mask if mask.ndim == 4 else mask[..., None]).contiguous()
    B, H, W, _ = rgb.shape
    if rgb.dtype != torch.uint8 or m.dtype != torch.uint8 or tuple(m.shape[:3]) != (B, H, W) or rgb.shape[3] != 3:
        raise ValueError(f"crop_normalize: rgb {tuple(rgb.shape)} {rgb.dtype} / mask {tuple(m.shape)} {m.dtype}")
    Mh = np.ascontiguousarray(np.asarray(M, np.float64).reshape(B, 6))
    mu = (ctypes.c_double * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_double * 3)(*[float(v) for v in std])
    out = torch.empty((B, 3, out_size, out_size), dtype=torch.float32, device=dev)
    cm = torch.empty((B, out_size, out_size), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_crop_normalize(ptr(rgb), ptr(m), B, H, W, m.shape[3], Mh.ctypes.data_as(ctypes.c_void_p), int(out_size),
                                      int(bool(use_mask)), ctypes.cast(mu, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p),
                                      ptr(out), ptr(cm), current_stream(dev))
    check(rc, "isr_crop_normalize")
    return out, cm


def gather_corr(idx, keep, M_dev, pts, pix_xy):
    """isr_gather_corr: p3d (P,3) f32, p2d (P,2) f32 (first M rows valid)."""
    dev = require_cuda(idx, keep, M_dev, pts, pix_xy)
    pts, pix_xy = _f32c(pts), _f32c(pix_xy)
    P = idx.numel()
    p3d = torch.empty((P, 3), dtype=torch.float32, device=dev)
    p2d = torch.empty((P, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_gather_corr(ptr(idx), ptr(keep), ptr(M_dev), P, ptr(pts), pts.shape[0],
                                   ptr(pix_xy), ptr(p3d), ptr(p2d), current_stream(dev))
    check(rc, "isr_gather_corr")
    return p3d, p2d


def _kcam(K) -> "ctypes.Array":
    import ctypes
    import numpy as np
    k = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(9))
    return (ctypes.c_double * 9)(*k.tolist())


def _m_dev(M, dev, cap):
    if isinstance(M, torch.Tensor):
        return M
    return torch.tensor([cap if M is None else int(M)], dtype=torch.int32, device=dev)


def p3p_hypotheses(p3d, p2d, Kcam, H: int, seed: int, M_dev=None, want_samples: bool = False):
    import ctypes
    dev = require_cuda(p3d, p2d)
    p3d, p2d = _f32c(p3d), _f32c(p2d)
    cap = p3d.shape[0]
    M_dev = _m_dev(M_dev, dev, cap)
    Rt = torch.empty((H, 3, 4), dtype=torch.float64, device=dev)
    ok = torch.empty(H, dtype=torch.uint8, device=dev)
    smp = torch.empty((H, 4), dtype=torch.int32, device=dev) if want_samples else None
    k = _kcam(Kcam)
    with torch.cuda.device(dev):
        rc = lib().isr_p3p_hypotheses(ptr(p3d), ptr(p2d), ptr(M_dev), cap, ctypes.cast(k, ctypes.c_void_p),
                                      H, seed & 0xFFFFFFFFFFFFFFFF, ptr(Rt), ptr(ok), ptr(smp),
                                      current_stream(dev))
    check(rc, "isr_p3p_hypotheses")
    return (Rt, ok, smp) if want_samples else (Rt, ok)


def p3p_all_roots(X, uv, Kcam):
    """isr_p3p_all_roots (diagnostics): X (S,3,3), uv (S,3,2) f64 on the device -> poses (S,4,3,4), n (S,)."""
    import ctypes
    dev = require_cuda(X, uv)
    X, uv = _f64c(X), _f64c(uv)
    S = X.shape[0]
    poses = torch.zeros((S, 4, 3, 4), dtype=torch.float64, device=dev)
    n = torch.zeros(S, dtype=torch.int32, device=dev)
    k = _kcam(Kcam)
    with torch.cuda.device(dev):
        rc = lib().isr_p3p_all_roots(ptr(X), ptr(uv), ctypes.cast(k, ctypes.c_void_p), S, ptr(poses), ptr(n),
                                     current_stream(dev))
    check(rc, "isr_p3p_all_roots")
    return poses, n


def ransac_score(p3d, p2d, Kcam, Rt, ok, reperr: float, M_dev=None):
    import ctypes
    dev = require_cuda(p3d, p2d, Rt, ok)
    p3d, p2d, Rt = _f32c(p3d), _f32c(p2d), _f64c(Rt)
    cap, H = p3d.shape[0], Rt.numel() // 12
    M_dev = _m_dev(M_dev, dev, cap)
    n_inl = torch.empty(H, dtype=torch.int32, device=dev)
    best = torch.empty(1, dtype=torch.int32, device=dev)
    mask = torch.zeros((cap + 31) // 32, dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_pnp_ransac_workspace_bytes(cap, H), "ransac")
    k = _kcam(Kcam)
    with torch.cuda.device(dev):
        rc = L.isr_ransac_score(ptr(p3d), ptr(p2d), ptr(M_dev), cap, ctypes.cast(k, ctypes.c_void_p),
                                ptr(Rt), ptr(ok.contiguous()), H, float(reperr), ptr(n_inl), ptr(best),
                                ptr(mask), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_ransac_score")
    return n_inl, best, mask


def pnp_refine(p3d, p2d, Kcam, Rt0, mask=None, iters: int = 10, M_dev=None):
    import ctypes
    dev = require_cuda(p3d, p2d, Rt0)
    p3d, p2d = _f32c(p3d), _f32c(p2d)
    cap = p3d.shape[0]
    M_dev = _m_dev(M_dev, dev, cap)
    Rt = _f64c(Rt0).clone().reshape(3, 4)
    L = lib()
    ws = workspace(dev, 1 << 20, "refine")
    k = _kcam(Kcam)
    with torch.cuda.device(dev):
        rc = L.isr_pnp_refine(ptr(p3d), ptr(p2d), ptr(M_dev), cap, ptr(mask), ctypes.cast(k, ctypes.c_void_p),
                              int(iters), ptr(Rt), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_pnp_refine")
    return Rt


@dataclass
class PnPResult:
    pose: torch.Tensor      # (3,4) f64 device
    inl_idx: torch.Tensor   # (cap,) i32 device, first n_inl valid
    n_inl: torch.Tensor     # (1,) i32 device
    status: torch.Tensor    # (1,) i32 device
    n_eval: torch.Tensor | None = None   # (1,) i32 device: hypotheses the staged loop scored


def pnp_ransac(p3d, p2d, Kcam, H: int = 500, reperr: float = 2.0, seed: int = 0,
               refine_iters: int = 10, M_dev=None, confidence: float = 0.99) -> PnPResult:
    """isr_pnp_ransac, fully asynchronous: every output stays on the device.  confidence: cv2's
    solvePnPRansac parameter (default 0.99, what the reference's call uses); >= 1 scores every hypothesis."""
    import ctypes
    dev = require_cuda(p3d, p2d)
    p3d, p2d = _f32c(p3d), _f32c(p2d)
    cap = p3d.shape[0]
    M_dev = _m_dev(M_dev, dev, cap)
    pose = torch.empty((3, 4), dtype=torch.float64, device=dev)
    inl = torch.empty(cap, dtype=torch.int32, device=dev)
    n_inl = torch.zeros(1, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    n_eval = torch.zeros(1, dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_pnp_ransac_workspace_bytes(cap, H), "ransac")
    k = _kcam(Kcam)
    with torch.cuda.device(dev), _timed("pnp_ransac", 30.0 * H * cap):
        rc = L.isr_pnp_ransac(ptr(p3d), ptr(p2d), ptr(M_dev), cap, ctypes.cast(k, ctypes.c_void_p), int(H),
                              seed & 0xFFFFFFFFFFFFFFFF, float(reperr), float(confidence), int(refine_iters), ptr(pose),
                              ptr(inl), ptr(n_inl), ptr(status), ptr(n_eval), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_pnp_ransac")
    return PnPResult(pose, inl, n_inl, status, n_eval)


# ------------------------------------------------------------------ the per-group (batched) chain
def select_top_batch(logp: torch.Tensor, frac: float = 0.8, min_n: int = 500, n_dev: torch.Tensor | None = None,
                     digit_hist: torch.Tensor | None = None):
    """isr_select_top_batch: logp (B, P) -> keep (B, P) i32 (first M[b] valid, ascending), M (B,) i32,
    thr (B,) f32.  One chain of ten launches for the whole group; outputs are not pre-filled.
    digit_hist (B, 2048) i32: the first histogram as corr_argmax(..., rows_per_image=P, n_rows=n_dev) left it for these
    values (isr_select_top_batch_digits: nine launches, one read of logp less, the same results)."""
    dev = require_cuda(logp, n_dev, digit_hist)
    logp = _f32c(logp)
    B, P = logp.shape
    keep = torch.empty((B, P), dtype=torch.int32, device=dev)
    M_dev = torch.empty(B, dtype=torch.int32, device=dev)
    thr = torch.empty(B, dtype=torch.float32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_select_top_batch_workspace_bytes(P, B), "select")
    if digit_hist is not None and (digit_hist.dtype != torch.int32 or tuple(digit_hist.shape) != (B, 2048) or not digit_hist.is_contiguous()):
        raise ValueError(f"select_top_batch: digit_hist must be a contiguous ({B}, 2048) int32 tensor")
    with torch.cuda.device(dev), _timed("select_top", 4.0 * P * B):
        if digit_hist is None:
            rc = L.isr_select_top_batch(ptr(logp), P, B, ptr(n_dev), float(frac), int(min_n), ptr(keep), ptr(M_dev),
                                        ptr(thr), ptr(ws), ws.numel(), current_stream(dev))
        else:
            rc = L.isr_select_top_batch_digits(ptr(logp), P, B, ptr(n_dev), float(frac), int(min_n), ptr(digit_hist), ptr(keep),
                                               ptr(M_dev), ptr(thr), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_select_top_batch")
    return keep, M_dev, thr


def gather_corr_batch(idx, keep, M_dev, pts, pix_xy):
    """isr_gather_corr_batch: idx, keep (B, P); pix_xy (P, 2) shared or (B, P, 2) -> p3d (B, P, 3), p2d (B, P, 2)."""
    dev = require_cuda(idx, keep, M_dev, pts, pix_xy)
    pts, pix_xy = _f32c(pts), _f32c(pix_xy)
    B, P = keep.shape
    shared = pix_xy.ndim == 2
    if tuple(pix_xy.shape) != ((P, 2) if shared else (B, P, 2)) or tuple(idx.shape) != (B, P):
        raise ValueError(f"gather_corr_batch: idx {tuple(idx.shape)} / keep {tuple(keep.shape)} / pix_xy {tuple(pix_xy.shape)}")
    p3d = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
    p2d = torch.empty((B, P, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_gather_corr_batch(ptr(idx.contiguous()), ptr(keep), ptr(M_dev), P, B, ptr(pts), pts.shape[0],
                                         ptr(pix_xy), int(shared), ptr(p3d), ptr(p2d), current_stream(dev))
    check(rc, "isr_gather_corr_batch")
    return p3d, p2d


@dataclass
class PnPBatchResult:
    pose: torch.Tensor      # (B,3,4) f64 device
    inl_idx: torch.Tensor   # (B,cap) i32 device, first n_inl[b] valid
    n_inl: torch.Tensor     # (B,) i32 device
    status: torch.Tensor    # (B,) i32 device
    n_eval: torch.Tensor | None = None   # (B,) i32 device: hypotheses the staged loop scored per image


def pnp_ransac_batch(p3d, p2d, Kcams, M_dev, H: int = 500, reperr: float = 2.0, seeds=None,
                     refine_iters: int = 10, confidence: float = 0.99) -> PnPBatchResult:
    """isr_pnp_ransac_batch: p3d (B, cap, 3), p2d (B, cap, 2), M_dev (B,) i32; Kcams one 3x3 or (B, 3, 3)
    host array; seeds B ints.  Every output stays on the device, nothing is pre-filled."""
    import ctypes
    import numpy as np
    dev = require_cuda(p3d, p2d, M_dev)
    p3d, p2d = _f32c(p3d), _f32c(p2d)
    B, cap = p3d.shape[0], p3d.shape[1]
    K = np.asarray(Kcams, dtype=np.float64)
    K = np.ascontiguousarray(np.broadcast_to(K.reshape(-1, 3, 3), (B, 3, 3)) if K.size == 9 else K.reshape(B, 3, 3))
    sd = np.ascontiguousarray(np.asarray([0] * B if seeds is None else [int(x) & 0xFFFFFFFFFFFFFFFF for x in seeds],
                                         dtype=np.uint64))
    if sd.shape != (B,):
        raise ValueError("pnp_ransac_batch: one seed per image")
    pose = torch.empty((B, 3, 4), dtype=torch.float64, device=dev)
    inl = torch.empty((B, cap), dtype=torch.int32, device=dev)
    n_inl = torch.empty(B, dtype=torch.int32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    n_eval = torch.empty(B, dtype=torch.int32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_pnp_ransac_batch_workspace_bytes(cap, H, B), "ransac")
    with torch.cuda.device(dev), _timed("pnp_ransac", 30.0 * H * cap * B):
        rc = L.isr_pnp_ransac_batch(ptr(p3d), ptr(p2d), ptr(M_dev), cap, B, K.ctypes.data_as(ctypes.c_void_p), int(H),
                                    sd.ctypes.data_as(ctypes.c_void_p), float(reperr), float(confidence),
                                    int(refine_iters), ptr(pose),
                                    ptr(inl), ptr(n_inl), ptr(status), ptr(n_eval), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_pnp_ransac_batch")
    return PnPBatchResult(pose, inl, n_inl, status, n_eval)


def add_metric(verts: torch.Tensor, Ta: torch.Tensor | None, Tb: torch.Tensor | None) -> torch.Tensor:
    """isr_add_metric: (B,) f64 mean vertex distance between poses Ta[b] and Tb[b]."""
    dev = require_cuda(verts, Ta, Tb)
    verts, Ta, Tb = _f32c(verts), _f64c(Ta), _f64c(Tb)
    B = max([1] + [T.numel() // 12 for T in (Ta, Tb) if T is not None])
    out = torch.empty(B, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_add_metric(ptr(verts), verts.shape[0], ptr(Ta), ptr(Tb), B, ptr(out), current_stream(dev))
    check(rc, "isr_add_metric")
    return out


def corr_logsoftmax(queries: torch.Tensor, keys: torch.Tensor) -> torch.Tensor:
    """isr_corr_logsoftmax: the full (P,N) f32 log-softmax matrix (small P only: it is written out)."""
    dev = require_cuda(queries, keys)
    if queries.dtype == torch.bfloat16 and keys.dtype == torch.bfloat16:
        dtype = _capi.DTYPE_BF16
    else:
        dtype = _capi.DTYPE_F32
        queries, keys = queries.to(torch.float32), keys.to(torch.float32)
    q, k = queries.contiguous(), keys.contiguous()
    P, D = q.shape
    N = k.shape[0]
    out = torch.empty((P, N), dtype=torch.float32, device=dev)
    L = lib()
    ws = workspace(dev, L.isr_corr_logsoftmax_workspace_bytes(P, N, D, dtype), "corr_lsm")
    with torch.cuda.device(dev):
        rc = L.isr_corr_logsoftmax(ptr(q), ptr(k), P, N, D, D, D, dtype, ptr(out), N, ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_corr_logsoftmax")
    return out
