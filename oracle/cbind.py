"""TEST INFRASTRUCTURE, NOT PRODUCT — ctypes binding of oracle/libisr_oracle.so (isr_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = _DIR / "libisr_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    src = _DIR / "isr_oracle.c"
    if force or not _LIB.exists() or _LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "-B" if force else "-s"], check=True,
                       capture_output=True)
    return _LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB.exists():
            build()
        _lib = C.CDLL(str(_LIB))
    return _lib


def _p(a, dtype):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


def nn_batched(qry, tgt, Tq=None, Tt=None, radius=-1.0, want_idx=True, want_dist=True,
               want_cov=True):
    """orc_nn_batched.  qry (Nq,3) f32, tgt (Nt,3) f32, Tq/Tt (B,3,4) f64 or None."""
    qry = np.ascontiguousarray(qry, np.float32)
    tgt = np.ascontiguousarray(tgt, np.float32)
    B = 1
    if Tq is not None:
        Tq = np.ascontiguousarray(Tq, np.float64).reshape(-1, 12)
        B = Tq.shape[0]
    if Tt is not None:
        Tt = np.ascontiguousarray(Tt, np.float64).reshape(-1, 12)
        B = Tt.shape[0]
    Nq, Nt = len(qry), len(tgt)
    sum_d = np.zeros(B)
    sum_d2 = np.zeros(B)
    n_in = np.zeros(B, np.int32)
    nn_idx = np.zeros((B, Nq), np.int32) if want_idx else None
    nn_d = np.zeros((B, Nq)) if want_dist else None
    cov = np.zeros((B, 16)) if want_cov else None
    f = lib().orc_nn_batched
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                  C.c_double] + [C.c_void_p] * 6
    f(_p(qry, np.float32), Nq, _p(tgt, np.float32), Nt, _p(Tq, np.float64), _p(Tt, np.float64), B,
      float(radius), _p(sum_d, np.float64), _p(sum_d2, np.float64), _p(n_in, np.int32),
      _p(nn_idx, np.int32), _p(nn_d, np.float64), _p(cov, np.float64))
    return dict(sum_d=sum_d, sum_d2=sum_d2, n_in=n_in, nn_idx=nn_idx, nn_d=nn_d, cov=cov)


def corr_argmax_f32(Q, K):
    """orc_corr_argmax_f32 -> idx i32, maxlogit f32, lse f64, top2 f32."""
    Q = np.ascontiguousarray(Q, np.float32)
    K = np.ascontiguousarray(K, np.float32)
    P, D = Q.shape
    N = K.shape[0]
    idx = np.zeros(P, np.int32)
    mx = np.zeros(P, np.float32)
    lse = np.zeros(P)
    top2 = np.zeros(P, np.float32)
    f = lib().orc_corr_argmax_f32
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4
    f(_p(Q, np.float32), _p(K, np.float32), P, N, D, D, D, _p(idx, np.int32), _p(mx, np.float32),
      _p(lse, np.float64), _p(top2, np.float32))
    return dict(idx=idx, maxlogit=mx, lse=lse, top2=top2)


def corr_argmax_bf16(Qbits, Kbits, logit_scale=1.0):
    """orc_corr_argmax_bf16 on uint16 bf16 bit patterns -> idx i32, maxlogit/lse/top2 f64.
    logit_scale = ln 2 when the queries were pre-multiplied by log2(e) (ISR_DTYPE_BF16_LOG2)."""
    Q = np.ascontiguousarray(Qbits, np.uint16)
    K = np.ascontiguousarray(Kbits, np.uint16)
    P, D = Q.shape
    N = K.shape[0]
    idx = np.zeros(P, np.int32)
    mx = np.zeros(P)
    lse = np.zeros(P)
    top2 = np.zeros(P)
    f = lib().orc_corr_argmax_bf16
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_double] + [C.c_void_p] * 4
    f(_p(Q, np.uint16), _p(K, np.uint16), P, N, D, D, D, float(logit_scale), _p(idx, np.int32), _p(mx, np.float64),
      _p(lse, np.float64), _p(top2, np.float64))
    return dict(idx=idx, maxlogit=mx, lse=lse, top2=top2)


def ransac_score(p3d, p2d, Kcam, Rt, ok, reperr):
    """orc_ransac_score -> n_inl (H,) i32, best int, best_mask (ceil(M/32),) u32."""
    p3d = np.ascontiguousarray(p3d, np.float32)
    p2d = np.ascontiguousarray(p2d, np.float32)
    Kc = np.ascontiguousarray(Kcam, np.float64).reshape(9)
    Rt = np.ascontiguousarray(Rt, np.float64).reshape(-1, 12)
    H, M = Rt.shape[0], p3d.shape[0]
    okb = None if ok is None else np.ascontiguousarray(ok, np.uint8)
    n_inl = np.zeros(H, np.int32)
    best = np.zeros(1, np.int32)
    mask = np.zeros((M + 31) // 32, np.uint32)
    f = lib().orc_ransac_score
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                  C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    f(_p(p3d, np.float32), _p(p2d, np.float32), M, _p(Kc, np.float64), _p(Rt, np.float64),
      _p(okb, np.uint8), H, float(reperr), _p(n_inl, np.int32), _p(best, np.int32),
      _p(mask, np.uint32))
    return dict(n_inl=n_inl, best=int(best[0]), best_mask=mask)
