"""Which register layout does v_mfma_scale_f32_32x32x64_f8f6f4 read?  (run on the GPU box)

Builds tools/mfma_scale_probe.hip into /tmp, writes register images of random exactly-representable matrices under each
layout hypothesis, runs one instruction per test and reports which hypotheses reproduce  D = (A sA) (B sB)  exactly.
    python tools/mfma_scale_probe.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/mfma_scale_probe"

FP4 = np.array([0, .5, 1, 1.5, 2, 3, 4, 6], dtype=np.float64)


def fp4_val(code):
    return np.where(code & 8, -1.0, 1.0) * FP4[code & 7]


def fp6_val(code):          # e2m3, bias 1
    s = np.where(code & 32, -1.0, 1.0)
    e = (code >> 3) & 3
    m = code & 7
    return s * np.where(e == 0, m * 0.125, (1 + m / 8.0) * 2.0 ** (e - 1))


def fp8_val(code):          # e4m3fn, bias 7
    s = np.where(code & 128, -1.0, 1.0)
    e = (code >> 3) & 15
    m = code & 7
    return s * np.where(e == 0, m * 2.0 ** -9, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))


KMAPS = {
    "H1 k=32h+j": lambda h, j: 32 * h + j,
    "H2 k=16h+j%16+32(j//16)": lambda h, j: 16 * h + (j % 16) + 32 * (j // 16),
    "H3 k=8h+j%8+16(j//8)": lambda h, j: 8 * h + (j % 8) + 16 * (j // 8),
}


def pack(codes, bits):
    """codes (64 lanes, 32 elements) -> (64, 8) uint32: element j in bits [bits*j, bits*(j+1)) of the lane's little-endian stream"""
    out = np.zeros((64, 8), dtype=np.uint64)
    for j in range(32):
        lo = bits * j
        w, o = lo // 32, lo % 32
        out[:, w] |= (codes[:, j].astype(np.uint64) << np.uint64(o)) & np.uint64(0xFFFFFFFF)
        if o + bits > 32:
            out[:, w + 1] |= codes[:, j].astype(np.uint64) >> np.uint64(32 - o)
    return out.astype(np.uint32)


def build(fmt, kmap, rng, scale_by_lane=True):
    bits = {0: 8, 2: 6, 4: 4}[fmt]
    val = {0: fp8_val, 2: fp6_val, 4: fp4_val}[fmt]
    if fmt == 0:     # moderate exponents only: products and sums stay exact in f32
        ca = (rng.integers(0, 2, (32, 64)) << 7) | (rng.integers(5, 10, (32, 64)) << 3) | rng.integers(0, 8, (32, 64))
        cb = (rng.integers(0, 2, (64, 32)) << 7) | (rng.integers(5, 10, (64, 32)) << 3) | rng.integers(0, 8, (64, 32))
    else:
        ca = rng.integers(0, 1 << bits, (32, 64))
        cb = rng.integers(0, 1 << bits, (64, 32))
    A, B = val(ca), val(cb)
    ea = rng.integers(125, 130, (32, 2))        # E8M0 scale per (row, 32-wide k block)
    eb = rng.integers(125, 130, (2, 32))
    lanes = np.arange(64)
    r, h = lanes & 31, lanes >> 5
    j = np.arange(32)
    k = kmap(h[:, None], j[None, :])            # (64, 32)
    a_codes = ca[r[:, None], k]
    b_codes = cb[k, r[:, None]]
    sa = ea[r, h].astype(np.uint32) * 0x01010101
    sb = eb[h, r].astype(np.uint32) * 0x01010101
    As = A * np.repeat(2.0 ** (ea - 127.0), 32, axis=1)
    Bs = B * np.repeat(2.0 ** (eb - 127.0), 32, axis=0)
    D = As @ Bs                                  # (32 rows, 32 cols)
    return pack(a_codes, bits), pack(b_codes, bits), sa, sb, D


def main():
    if subprocess.call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", os.path.join(ROOT, "tools/mfma_scale_probe.hip"),
                        "-o", EXE]) != 0:
        sys.exit("hipcc failed")
    rng = np.random.default_rng(0)
    tests, meta = [], []
    for fmt in (4, 2, 0):
        for name, km in KMAPS.items():
            for rep in range(2):
                a, b, sa, sb, D = build(fmt, km, rng)
                tests.append((fmt, a, b, sa, sb))
                meta.append((fmt, name, D))
    with open("/tmp/probe_in.bin", "wb") as f:
        f.write(np.int32(len(tests)).tobytes())
        for fmt, a, b, sa, sb in tests:
            f.write(np.int32(fmt).tobytes())
            f.write(a.tobytes()); f.write(b.tobytes()); f.write(sa.tobytes()); f.write(sb.tobytes())
    out = subprocess.run([EXE, "/tmp/probe_in.bin", "/tmp/probe_out.bin"], capture_output=True, text=True)
    print(out.stdout, out.stderr)
    res = np.fromfile("/tmp/probe_out.bin", dtype=np.float32).reshape(len(tests), 64, 16)
    lanes = np.arange(64)
    reg = np.arange(16)
    row = (reg[None, :] & 3) + 8 * (reg[None, :] >> 2) + 4 * (lanes[:, None] >> 5)
    col = np.broadcast_to(lanes[:, None] & 31, row.shape)
    for (fmt, name, D), got in zip(meta, res):
        exp = D[row, col]
        ok = np.array_equal(exp.astype(np.float32), got)
        print(f"fmt {fmt} ({ {0: 'fp8', 2: 'fp6', 4: 'fp4'}[fmt] }) {name}: {'EXACT' if ok else 'no'}  max|diff| {np.abs(exp - got).max():.4g}")


if __name__ == "__main__":
    main()
