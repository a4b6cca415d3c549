"""GPU parity: ISR_DTYPE_BF16_LOG2_SCREENED (csrc/corr_sparse.hpp) — K1 for 64-wide bf16 rows behind a rigorous FP6 screen.
Replaces getCors(queries, feats), inference.py:142-149.
  * indices: array_equal to the C oracle's exact arg-max (lowest key on ties), as on every bf16 route;
  * logp / lse: within 3e-5 of the oracle's full f64 sums (the screen leaves out < 2^-21 of a sum: 5e-7 in lse);
  * a query's outputs are a function of (query, keys) ONLY: torch.equal whatever the launch holds — including launches whose
    other query blocks are flat and get handed to the dense kernel, and blocks that are themselves handed over (the dense
    kernel applies the same rule to the same logits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D = 64


def _planted(rng, P, N, tau=8.0, sigma=0.35):
    K = rng.normal(0, 1, (N, D))
    K *= tau / np.linalg.norm(K, axis=1, keepdims=True)
    gt = rng.integers(N, size=P)
    Q = K[gt] + sigma * rng.normal(0, 1, (P, D))
    return Q.astype(np.float32), K.astype(np.float32), gt


def _bits(t):
    return t.view(torch.int16).numpy().view(np.uint16)


def _run(ops, dev, qb, kb, **kw):
    idx, logp, lse = ops.corr_argmax(qb.to(dev), kb.to(dev), want_lse=True, log2_prescaled=True, screened=True, **kw)
    torch.cuda.synchronize()
    return idx.cpu(), logp.cpu(), lse.cpu()


def _check(ops, oracle_lib, dev, Q, K, atol=3e-5):
    qb, kb = ops.prescale_queries_log2(torch.from_numpy(Q)), torch.from_numpy(K).bfloat16()
    idx, logp, lse = _run(ops, dev, qb, kb)
    o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb), logit_scale=np.log(2.0))
    bad = np.nonzero(idx.numpy() != o["idx"])[0]
    assert len(bad) == 0, f"{len(bad)} index mismatches, margins {(o['maxlogit'] - o['top2'])[bad][:5]}"
    scale = np.maximum(1.0, np.abs(o["lse"]))
    assert np.max(np.abs(logp.numpy() - (o["maxlogit"] - o["lse"]))) <= atol
    assert np.max(np.abs(lse.numpy() - o["lse"]) / scale) <= atol
    return (idx, logp, lse), o, (qb, kb)


@pytest.mark.parametrize("P,N,planted,scale", [
    (1, 1, True, 1.0), (3, 5, True, 1.0), (300, 999, True, 1.0), (1000, 20000, True, 1.0), (4096, 33, True, 1.0),
    (777, 300, True, 0.3), (2048, 4097, False, 1.0), (5000, 20000, False, 1.0), (513, 8192, True, 1.0), (700, 50000, True, 1.0),
])
def test_screened_against_the_oracle(cuda0, oracle_lib, P, N, planted, scale):
    """Planted winners (the screen skips almost everything), unplanted Gaussian queries (nothing can be skipped: the blocks are
    handed to the dense kernel when the key range is long enough, redone item by item otherwise), ragged P and N, one key."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(P * 7 + N)
    Q, K, _ = _planted(rng, P, N)
    if not planted:
        Q = rng.normal(0, 1, (P, D)).astype(np.float32)
    _check(ops, oracle_lib, cuda0, (Q * scale).astype(np.float32), K)
    redone, handed = ops.corr_screen_redone()
    tiles = ((P + 31) // 32) * ((N + 31) // 32)
    if planted and N >= 20000 and scale == 1.0:
        assert handed == 0 and redone < 0.05 * tiles
    if not planted and N >= 8 * 256:
        assert handed == (P + 255) // 256                     # every block of flat queries goes to the dense kernel


def test_screened_equals_the_unscreened_route_to_the_last_bits_that_matter(cuda0):
    """Same indices; logp / lse within 2e-6 absolute of the unscreened kernel's (the left-out pieces are < 5e-7 of a sum and the
    f32 outputs have a spacing of 4e-6 at lse = 64)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(5)
    Q, K, gt = _planted(rng, 40000, 20000)
    qb, kb = ops.prescale_queries_log2(torch.from_numpy(Q)).to(cuda0), torch.from_numpy(K).bfloat16().to(cuda0)
    a = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True)
    b = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True, screened=True)
    assert torch.equal(a[0], b[0]) and bool((a[0].cpu().long() == torch.from_numpy(gt)).all())
    assert float((a[1] - b[1]).abs().max()) <= 2e-6
    assert float((a[2] - b[2]).abs().max()) <= 8e-6 and float(((a[2] - b[2]).abs() / a[2].abs()).max()) <= 2e-7


def test_screened_result_is_a_function_of_query_and_keys_only(cuda0):
    """Peaked and flat queries in one launch: flat 256-query blocks are handed to the dense tile-skip kernel, peaked ones stay
    with the screen, blocks that mix both go either way — and every query's (idx, logp, lse) is bit for bit what it is in a
    launch of its own kind, alone, reordered, or sliced differently."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(17)
    N = 20000
    Qp, K, _ = _planted(rng, 3000, N)
    Qf = rng.normal(0, 1, (3000, D)).astype(np.float32)
    kb = torch.from_numpy(K).bfloat16()
    q_all = ops.prescale_queries_log2(torch.from_numpy(np.concatenate([Qp, Qf])))
    full = _run(ops, cuda0, q_all, kb)
    _, handed = ops.corr_screen_redone()
    assert 0 < handed < (6000 + 255) // 256                    # some blocks screened, some dense
    perm = torch.from_numpy(rng.permutation(6000))
    shuf = _run(ops, cuda0, q_all[perm], kb)                    # every block now mixes peaked and flat queries
    for a, b in zip(full, shuf):
        assert torch.equal(a[perm], b)
    for lo, hi in ((0, 3000), (3000, 6000), (100, 357), (2990, 3010), (5999, 6000)):
        part = _run(ops, cuda0, q_all[lo:hi].contiguous(), kb)
        for a, b in zip(full, part):
            assert torch.equal(a[lo:hi], b), (lo, hi)


def test_screened_lse_only_call_returns_the_full_call_s_bits(cuda0):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(3)
    Q, K, _ = _planted(rng, 5000, 9000)
    Q[1000:1300] = rng.normal(0, 1, (300, D))
    qb, kb = ops.prescale_queries_log2(torch.from_numpy(Q)).to(cuda0), torch.from_numpy(K).bfloat16().to(cuda0)
    full = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True, screened=True)
    only = ops.corr_lse(qb, kb, log2_prescaled=True, screened=True)
    assert torch.equal(full[2], only)


def test_screened_zero_rows_bad_queries_and_ties(cuda0, oracle_lib):
    """Padding rows (all zero: finished without touching the keys when a whole workgroup is zero, through the loop otherwise),
    queries outside the direct sum's range (maxima of several hundred log2 units, or far below zero: the per-query fallback
    kernel owns them, per QUERY), duplicated keys (the exact recheck picks the lowest) — next to ordinary neighbours."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(29)
    P, N = 2048, 6000
    Q, K, gt = _planted(rng, P, N, tau=5.0)
    Q[0:512] = 0.0                              # two whole 256-query blocks of padding
    Q[600:640] = 0.0                            # padding inside a live block
    Q[768:1024:2] *= 12.0                       # ~430 nats next to ~36
    Q[1024:1280] = -Q[1024:1280] - 6.0 * np.sign(K).mean(0)     # everything far below zero
    K[5000] = K[17]                             # a duplicate of an early key ...
    K[5900] = K[17]
    Q[1500] = K[17]                             # ... and a query whose winner is tied three ways
    (idx, _, _), o, _ = _check(ops, oracle_lib, cuda0, Q, K, atol=2e-4)
    assert int(idx[1500]) == 17 and int(idx[3]) == 0


def test_screened_near_ties_behind_the_screen(cuda0, oracle_lib):
    """A dominant key far down the stream, and one-ulp rivals of it (bf16 neighbours of the same row) in tiles whose FP6 image
    cannot tell them from the winner: every such tile must pass the screen and be formed exactly, so that the margin test sees
    the rival and the exact recheck decides (lowest key among exact ties, the larger exact logit otherwise)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(31)
    P, N = 1024, 12000
    K = (0.05 * rng.normal(0, 1, (N, D))).astype(np.float32)          # near-zero keys
    star = (8.0 * rng.normal(0, 1, D) / np.sqrt(D)).astype(np.float32)
    K[9000] = star
    kb = torch.from_numpy(K).bfloat16()
    kbits = _bits(kb).copy()
    for j, n in enumerate((40, 3333, 9001, 11999)):                     # one-ulp neighbours of the star row, one element each
        row = kbits[9000].copy()
        row[j] ^= 1
        kbits[n] = row
    kb = torch.from_numpy(kbits.view(np.int16)).view(torch.bfloat16)
    Q = (star[None, :] * (1.0 + 0.02 * rng.normal(0, 1, (P, 1))) + 0.05 * rng.normal(0, 1, (P, D))).astype(np.float32)
    qb = ops.prescale_queries_log2(torch.from_numpy(Q))
    idx, logp, lse = _run(ops, cuda0, qb, kb)
    o = oracle_lib.corr_argmax_bf16(_bits(qb), kbits, logit_scale=np.log(2.0))
    assert np.array_equal(idx.numpy(), o["idx"])
    assert ops.corr_screen_redone()[0] >= 3 * (P // 32)                  # the rivals' tiles (three besides the star's) were formed exactly
    np.testing.assert_allclose(logp.numpy(), o["maxlogit"] - o["lse"], atol=3e-5)


def test_screened_full_size_properties(cuda0, oracle_lib):
    """BASELINE configs[1]'s shape: 640 x 480 queries against 20 000 keys.  768 rows against the oracle; every planted key
    recovered; launch independence on a slice; the screen redoes a few percent of the tile items and hands nothing over."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(41)
    P, N = 307200, 20000
    Q, K, gt = _planted(rng, P, N)
    qb, kb = ops.prescale_queries_log2(torch.from_numpy(Q)), torch.from_numpy(K).bfloat16()
    idx, logp, lse = _run(ops, cuda0, qb, kb)
    redone, handed = ops.corr_screen_redone()
    assert handed == 0 and redone < 0.03 * (P // 32) * ((N + 31) // 32)
    assert np.array_equal(idx.numpy(), gt)
    rows = rng.choice(P, 768, replace=False)
    o = oracle_lib.corr_argmax_bf16(_bits(qb)[rows], _bits(kb), logit_scale=np.log(2.0))
    assert np.array_equal(idx.numpy()[rows], o["idx"])
    np.testing.assert_allclose(logp.numpy()[rows], o["maxlogit"] - o["lse"], atol=3e-5)
    np.testing.assert_allclose(lse.numpy()[rows], o["lse"], rtol=2e-6, atol=3e-5)
    part = _run(ops, cuda0, qb[100000:100700].contiguous(), kb)
    assert torch.equal(part[0], idx[100000:100700]) and torch.equal(part[1], logp[100000:100700]) and torch.equal(part[2], lse[100000:100700])


def test_screened_dtype_on_other_shapes_runs_the_unscreened_kernels(cuda0):
    """D != 64: the dtype is accepted and means ISR_DTYPE_BF16_LOG2 (the same bits)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(2)
    K = rng.normal(0, 1, (3000, 32)).astype(np.float32)
    Q = (K[rng.integers(3000, size=900)] + 0.3 * rng.normal(0, 1, (900, 32))).astype(np.float32)
    qb, kb = ops.prescale_queries_log2(torch.from_numpy(Q)).to(cuda0), torch.from_numpy(K).bfloat16().to(cuda0)
    a = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True)
    b = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True, screened=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert ops.corr_screen_redone() == (0, 0)


def test_device_quantiser_equals_the_oracle_s_image_byte_for_byte(cuda0):
    """isr_corr_quantize_fp6 (the kernel isr_corr_argmax runs on queries and keys) against oracle/fp6_screen_oracle.py: the 64-byte
    row images — codes, scale bytes, padding — are equal, the norms agree to f32 rounding.  With tests/test_oracle_fp6_screen_cpu.py
    (the bound holds for that image, and is tight) this pins the numbers the screen's proof is made of."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    from oracle import fp6_screen_oracle as fo
    rng = np.random.default_rng(8)
    X = np.concatenate([fo.bf16_round((rng.normal(0, 1, (3000, 64)) * s).astype(np.float32)) for s in (1.0, 0.003, 500.0, 1e-20, 1e20, 3e-38, 2e37)])
    X[5] = 0.0
    X[6, 32:] = 0.0
    X[7, 9] = 2.0e30
    sig = np.where(rng.random(64) < 0.5, -1.0, 1.0)
    X[8] = sig * 1.0625 * 0.25                                  # rounding midpoints (ties to even: toward zero here)
    X[9] = sig * 1.1875
    X[10] = 7.5
    X[11] = 7.53                                                 # bf16(7.53) = 7.53125 > 7.5: the next scale
    X = fo.bf16_round(X)
    img, nrm, kmax = ops.corr_quantize_fp6(torch.from_numpy(X).bfloat16().to(cuda0))
    torch.cuda.synchronize()
    o = fo.quantize_e2m3(X)
    assert np.array_equal(img.cpu().numpy(), o["image"])
    np.testing.assert_allclose(nrm.cpu().numpy(), o["nrm"], rtol=3e-7)
    np.testing.assert_allclose(kmax.cpu().numpy(), [o["d2"].max(), o["t2"].max()], rtol=3e-7)


def test_screened_dtype_beyond_the_tile_index_range_runs_unscreened(cuda0):
    """N > 262 144 keys: pass 0 carries the tile index in 13 mantissa bits, so such calls take the unscreened kernels (the same
    bits as ISR_DTYPE_BF16_LOG2)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(4)
    K = torch.randn(300000, D, device=cuda0, generator=g)
    K = 8.0 * K / K.norm(dim=1, keepdim=True)
    gt = torch.randint(300000, (200,), device=cuda0, generator=g)
    qb = ops.prescale_queries_log2(K[gt] + 0.35 * torch.randn(200, D, device=cuda0, generator=g))
    kb = K.bfloat16()
    a = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True)
    b = ops.corr_argmax(qb, kb, want_lse=True, log2_prescaled=True, screened=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and ops.corr_screen_redone() == (0, 0)
    assert bool((a[0].long() == gt).all())


@pytest.mark.parametrize("N", [33, 167, 232, 256, 300, 1000])
def test_screened_pieces_enter_in_tile_order(cuda0, oracle_lib, N):
    """Round-5 stress find: zero rows INSIDE a live block (every logit equal: the arg-max is key 0, the first tile reaching the
    maximum) came out as the first key of the query's pass-0 tile, because that tile's stored pieces were applied while the stage
    was screened and the fetched tiles behind it.  Also: keys repeated across tiles (equal maxima in several tiles), and queries
    with several pieces that count — their sums must carry the dense kernel's order of additions, so the bytes of the sparse
    kernel (few flagged items) and of the dense kernel (block handed over) agree; checked against the unscreened route's index
    and, for the order, through a launch in which the same queries sit in a block that is handed over."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(N)
    P = 4096
    Q, K, _ = _planted(rng, P, N, tau=8.0)
    K[N // 2:] = K[: N - N // 2]                               # every key twice, in different tiles for N > 64
    Q[rng.random(P) < 0.2] = 0.0                               # padding rows scattered through live blocks
    (idx, logp, lse), o, (qb, kb) = _check(ops, oracle_lib, cuda0, Q, K)
    zero = np.nonzero(~Q.any(axis=1))[0]
    assert len(zero) > 100 and (idx.numpy()[zero] == 0).all()
    a = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0), want_lse=True, log2_prescaled=True)
    assert torch.equal(a[0].cpu(), idx)


def test_screened_sparse_and_dense_kernels_add_in_the_same_order(cuda0):
    """Queries with SEVERAL pieces that count (their key copied, at 0.8 ... 1.0 of its length, into five other tiles) placed (a)
    among planted queries, whose block stays with the sparse kernel, and (b) among flat queries, whose block goes to the dense
    kernel: the same bytes in both, so the fetched pieces, the stored pass-0 pieces and the dense kernel's tiles all enter a sum
    at the same position."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(77)
    N = 6000
    _, K, gt = _planted(rng, 512, N, tau=8.0)
    g = rng.choice(N, size=64, replace=False)
    for s_ in (0.8, 0.9, 0.95, 0.99, 1.0):
        K[rng.choice(N, size=64, replace=False)] = K[g] * s_
    Qp = (K[gt] + 0.35 * rng.normal(0, 1, (512, D))).astype(np.float32)    # planted on the keys as they now are
    mid = (K[g] + 0.35 * rng.normal(0, 1, (64, D))).astype(np.float32)
    flat = rng.normal(0, 1, (512, D)).astype(np.float32)
    kb = torch.from_numpy(K).bfloat16()
    A = np.concatenate([Qp[:192], mid, Qp[192:256 + 192]])    # block 0: 256 queries, 64 of them `mid`: stays sparse
    B = np.concatenate([flat[:192], mid, flat[192:]])          # block 0: flat company: handed over
    ra = _run(ops, cuda0, ops.prescale_queries_log2(torch.from_numpy(A)), kb)
    red_a, handed_a = ops.corr_screen_redone()
    rb = _run(ops, cuda0, ops.prescale_queries_log2(torch.from_numpy(B)), kb)
    red_b, handed_b = ops.corr_screen_redone()
    assert handed_a == 0 and handed_b >= 1 and red_a >= 64 * 3, (handed_a, handed_b, red_a)
    for x, y in zip(ra, rb):
        assert torch.equal(x[192:256], y[192:256])
