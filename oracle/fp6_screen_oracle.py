"""TEST INFRASTRUCTURE, NOT PRODUCT — the arithmetic of K1's FP6 screen (csrc/corr_sparse.hpp) restated in NumPy.

The screened route of getCors (inference.py:142-149; ISR_DTYPE_BF16_LOG2_SCREENED) skips a tile only when a block-scaled FP6 image
of the rows PROVES its logits lie far below the query's maximum.  What must hold for that proof:
    |<q, k> - <q~, k~>|  <=  |q| |k - k~| + |q - q~| |k~|          (q~, k~ the dequantised images; Cauchy-Schwarz)
This file restates the quantiser (corr_quant_fp6_kernel: e2m3 codes, E8M0 scale per 32 elements, the 64-byte row image and the
norms it leaves) and the bound (screen_error), so that CPU tests can check the inequality on random and adversarial rows and
GPU tests can compare the device's image with this one byte for byte.  parity: the instruction's arithmetic itself
(v_mfma_scale_f32_32x32x64_f8f6f4) is established on hardware by tools/mfma_scale_probe.py, not restated here.
"""
from __future__ import annotations

import numpy as np


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def quantize_e2m3(rows: np.ndarray):
    """rows (R, 64) float32 holding bf16 values -> dict(codes (R, 64) uint8 with the sign in bit 5, scale_exp (R, 2) int,
    deq (R, 64) float32 = the values the matrix instruction sees, image (R, 64) uint8 = the device's row layout,
    nrm (R, 2) float32 = {|x|, |x - x~|} as the device rounds and inflates them, d2 / t2 (R,) float32 = |x - x~|^2, |x~|^2)."""
    x = np.asarray(rows, np.float32)
    R = x.shape[0]
    assert x.shape[1] == 64
    xb = x.reshape(R, 2, 32)
    mx = np.abs(xb).max(-1)                                           # (R, 2)
    e = np.full((R, 2), -127, np.int64)
    pos = mx > 0
    t = (mx * np.float32(1.0 / 7.5)).astype(np.float32)
    f, p = np.frexp(np.where(pos, t, np.float32(1.0)))
    ee = np.where(f > 0.5, p, p - 1)
    ee = ee + (mx > np.float32(7.5) * np.ldexp(np.float32(1.0), ee).astype(np.float32))
    e = np.where(pos, np.clip(ee, -127, 127), e)
    inv = np.ldexp(np.float32(1.0), -e).astype(np.float32)[..., None]
    sc = np.ldexp(np.float32(1.0), e).astype(np.float32)[..., None]
    y = np.minimum(np.abs(xb) * inv, np.float32(7.5)).astype(np.float32)
    c = np.where(y < 2, np.rint(y * np.float32(8.0)),
                 np.where(y < 4, 16 + np.rint((y - np.float32(2.0)) * np.float32(4.0)),
                          np.minimum(24 + np.rint((y - np.float32(4.0)) * np.float32(2.0)), 31))).astype(np.int64)
    v = np.where(c < 16, c * 0.125, np.where(c < 24, 2.0 + (c - 16) * 0.25, 4.0 + (c - 24) * 0.5)).astype(np.float32)
    deq = np.copysign(v * sc, xb).astype(np.float32)
    codes = (c | ((np.signbit(xb)).astype(np.int64) << 5)).astype(np.uint8)
    # the device's f32 fma chains over the 32 elements of a half, then half 0 + half 1
    dx = (xb - deq).astype(np.float32)

    def chain(a):
        s = np.zeros(a.shape[:-1], np.float32)
        with np.errstate(over="ignore"):
            for j in range(a.shape[-1]):
                s = (a[..., j].astype(np.float64) * a[..., j].astype(np.float64) + s.astype(np.float64)).astype(np.float32)   # one rounding: fmaf
        return s

    n2 = chain(xb); d2 = chain(dx); t2 = chain(deq)
    with np.errstate(over="ignore"):
        n2, d2, t2 = (n2[:, 0] + n2[:, 1]).astype(np.float32), (d2[:, 0] + d2[:, 1]).astype(np.float32), (t2[:, 0] + t2[:, 1]).astype(np.float32)
    infl = np.float32(1.00001)
    nrm = np.stack([np.sqrt(n2).astype(np.float32) * infl, np.sqrt(d2).astype(np.float32) * infl], 1).astype(np.float32)
    # the 64-byte row image: per half 24 B of codes (element j at bits [6j, 6j + 6)), the scale byte, 7 zero bytes
    image = np.zeros((R, 64), np.uint8)
    for hlf in range(2):
        bits = np.zeros((R, 24), np.uint8)
        stream = np.zeros((R, 192), np.uint8)
        for b in range(6):
            stream[:, b::6] = (codes.reshape(R, 2, 32)[:, hlf, :] >> b) & 1
        for byte in range(24):
            bits[:, byte] = np.packbits(stream[:, 8 * byte:8 * byte + 8], axis=1, bitorder="little")[:, 0]
        image[:, 32 * hlf:32 * hlf + 24] = bits
        image[:, 32 * hlf + 24] = (e[:, hlf] + 127).astype(np.uint8)
    return {"codes": codes.reshape(R, 64), "scale_exp": e, "deq": deq.reshape(R, 64), "image": image, "nrm": nrm, "d2": d2, "t2": t2}


def screen_error(qn, dqn, dk2max, kt2max):
    """E_q of csrc/corr_sparse.hpp:screen_error, in float32 operation for operation."""
    f = np.float32
    dk = np.sqrt(f(dk2max)).astype(np.float32) * f(1.00001)
    kt = np.sqrt(f(kt2max)).astype(np.float32) * f(1.00001)
    qn, dqn = np.asarray(qn, np.float32), np.asarray(dqn, np.float32)
    return ((qn * dk + dqn * kt) * f(1.0001) + (qn + dqn) * kt * f(1.53e-5 + 66.0 * 1.1920929e-7 * 1.01) + f(1e-6)).astype(np.float32)


def screen_T(N: int) -> int:
    """T = 21 + ceil(log2 N) log2 units (csrc/corr_sparse.hpp:screen_T)."""
    tl = 0
    while (1 << tl) < N:
        tl += 1
    return 21 + tl
