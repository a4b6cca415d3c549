"""GPU parity: isr_corr_argmax through the C ABI vs the C oracle.
f32 path: arg-max indices bit-exact (the f32 MFMA is a k-ordered fmaf chain, as the oracle).
bf16 paths: arg-max indices bit-exact as well — queries whose top-2 margin is inside the f32
accumulation error bound are decided by the exact recheck (f64 sums of exact bf16 products in the
oracle's order, lowest key on ties); logp / lse within 2e-5 (f32 exp/sum vs f64).
A query's (idx, logp) is a function of (query, keys) only: bit-identical whatever launch it rides in."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _planted(rng, P, N, D, tau=8.0, sigma=0.35):
    K = rng.normal(0, 1, (N, D))
    K *= tau / np.linalg.norm(K, axis=1, keepdims=True)
    gt = rng.integers(N, size=P)
    Q = K[gt] + sigma * rng.normal(0, 1, (P, D))
    return Q.astype(np.float32), K.astype(np.float32), gt


def _bits(t):
    return t.view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("P,N,D", [
    (1, 1, 12), (5, 33, 12), (4500, 20000, 12),      # reference shape (D=12 -> padded 16)
    (257, 1000, 7), (300, 4097, 33), (1024, 6400, 64), (300, 3000, 100), (129, 2100, 128),
])
def test_corr_f32_bit_exact(cuda0, oracle_lib, P, N, D):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(P + N + D)
    Q, K, _ = _planted(rng, P, N, D)
    idx, logp, lse = ops.corr_argmax(torch.from_numpy(Q).to(cuda0), torch.from_numpy(K).to(cuda0),
                                     want_lse=True)
    torch.cuda.synchronize()
    o = oracle_lib.corr_argmax_f32(Q, K)
    assert np.array_equal(idx.cpu().numpy(), o["idx"])
    ref_logp = o["maxlogit"].astype(np.float64) - o["lse"]
    np.testing.assert_allclose(logp.cpu().numpy(), ref_logp, atol=2e-5)
    np.testing.assert_allclose(lse.cpu().numpy(), o["lse"], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("P,N,D", [
    (3, 5, 16), (64, 64, 32), (1000, 20000, 64), (777, 3001, 128), (4096, 50000, 64), (100, 999, 48),
])
def test_corr_bf16(cuda0, oracle_lib, P, N, D):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(P * 3 + N + D)
    Q, K, gt = _planted(rng, P, N, D)
    qb, kb = torch.from_numpy(Q).bfloat16(), torch.from_numpy(K).bfloat16()   # rounded ONCE on host
    idx, logp, lse = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0), want_lse=True)
    torch.cuda.synchronize()
    o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb))
    assert np.array_equal(idx.cpu().numpy(), o["idx"])
    np.testing.assert_allclose(logp.cpu().numpy(), o["maxlogit"] - o["lse"], atol=2e-5)
    np.testing.assert_allclose(lse.cpu().numpy(), o["lse"], rtol=2e-6, atol=2e-5)


def test_corr_random_unplanted_and_spiked(cuda0, oracle_lib):
    """No planted peak (flat softmax: l sums 20k comparable terms) and a late spike that forces
    the rescale branch in the last key tile."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(11)
    P, N, D = 512, 20000, 64
    Q = rng.normal(0, 0.4, (P, D)).astype(np.float32)
    K = rng.normal(0, 0.4, (N, D)).astype(np.float32)
    K[N - 1] = 3.0 * Q[7]          # query 7 meets its max at the very last key
    K[0] = 3.0 * Q[9]              # query 9 at the very first
    qb, kb = torch.from_numpy(Q).bfloat16(), torch.from_numpy(K).bfloat16()
    idx, logp = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0))
    torch.cuda.synchronize()
    o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb))
    got = idx.cpu().numpy()
    assert got[7] == N - 1 and got[9] == 0
    assert np.array_equal(got, o["idx"])
    np.testing.assert_allclose(logp.cpu().numpy(), o["maxlogit"] - o["lse"], atol=3e-5)


def test_corr_ties_lowest_key(cuda0):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    K = torch.zeros(200, 16)
    for j in (3, 4, 36, 70, 199):      # same lane half, other half, other tiles, last key
        K[j, 0] = 1.0
    Q = torch.zeros(2, 16)
    Q[0, 0] = 2.0
    Q[1, 0] = -1.0                      # every other key ties at 0 -> key 0
    for dt in (torch.float32, torch.bfloat16):
        idx, _ = ops.corr_argmax(Q.to(dt).to(cuda0), K.to(dt).to(cuda0))
        assert idx.cpu().tolist() == [3, 0]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_corr_logsoftmax_and_topk_leaves(cuda0, dt):
    """Materialising variant (poseEstSurf.py:70) and getCors(leaves=3) vs the literal torch expression."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, registration
    rng = np.random.default_rng(21)
    Q, K, _ = _planted(rng, 70, 1234, 12)
    q, k = torch.from_numpy(Q).to(dt), torch.from_numpy(K).to(dt)
    got = ops.corr_logsoftmax(q.to(cuda0), k.to(cuda0)).cpu()
    ref = torch.log_softmax(q.double() @ k.double().T, dim=-1)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=5e-5)
    idx, vals = registration.getCors(q.to(cuda0), k.to(cuda0), leaves=3)
    rv, ri = torch.topk(ref, k=3, dim=-1)
    assert idx.shape == (70, 3) and not idx.is_cuda and vals.is_cuda
    # 99 %, not equality: `ref` is an f64 matmul, the kernel ranks by the k-ordered f32 chain (bf16: by the f32 MFMA logits), so
    # two of a row's top-3 whose f64 logits differ by less than the f32 rounding may swap places; the strict check of the
    # ranking against the chain's own logits is test_corr_topk_without_the_matrix
    assert (idx == ri).float().mean() > 0.99
    np.testing.assert_allclose(vals.cpu().numpy(), rv.numpy(), atol=5e-5)


@pytest.mark.parametrize("P,N,D", [(3, 5, 16), (1000, 20000, 64), (777, 3001, 128), (300, 999, 32), (2048, 50000, 64)])
def test_corr_bf16_log2_prescaled(cuda0, oracle_lib, P, N, D):
    """ISR_DTYPE_BF16_LOG2: queries rounded to bf16 AFTER a log2(e) prescale; the oracle gets the
    same bits and logit_scale = ln 2.  Indices bit-exact as on the plain bf16 path."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(P * 5 + N + D)
    Q, K, gt = _planted(rng, P, N, D)
    qb = ops.prescale_queries_log2(torch.from_numpy(Q))
    kb = torch.from_numpy(K).bfloat16()
    idx, logp, lse = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0), want_lse=True, log2_prescaled=True)
    torch.cuda.synchronize()
    o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb), logit_scale=np.log(2.0))
    got = idx.cpu().numpy()
    assert np.array_equal(got, o["idx"])
    np.testing.assert_allclose(logp.cpu().numpy(), o["maxlogit"] - o["lse"], atol=3e-5)
    np.testing.assert_allclose(lse.cpu().numpy(), o["lse"], rtol=2e-6, atol=3e-5)
    if P >= 300:
        assert (got == gt).mean() > 0.99


def test_corr_bf16_log2_extreme_logits(cuda0, oracle_lib):
    """Very negative / very large logits: the integer reference M2 must follow in both directions
    (first-key initialisation below zero, bumps of several hundred)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(8)
    N, D = 700, 32
    K = rng.normal(0, 1, (N, D)).astype(np.float32)
    Q = np.stack([-40.0 * K[5], 55.0 * K[690], 0.01 * K[3], 30.0 * K[0]]).astype(np.float32)
    Q[0] -= 30 * np.sign(K).mean(0).astype(np.float32)       # pushes every logit of row 0 far below zero
    qb, kb = ops.prescale_queries_log2(torch.from_numpy(Q)), torch.from_numpy(K).bfloat16()
    idx, logp, lse = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0), want_lse=True, log2_prescaled=True)
    torch.cuda.synchronize()
    o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb), logit_scale=np.log(2.0))
    assert np.array_equal(idx.cpu().numpy(), o["idx"])
    np.testing.assert_allclose(logp.cpu().numpy(), o["maxlogit"] - o["lse"], atol=1e-4)
    np.testing.assert_allclose(lse.cpu().numpy(), o["lse"], rtol=3e-6, atol=1e-3)


def _check_log2(ops, oracle_lib, cuda0, Q, K, atol=3e-5, log2=True):
    """log2=True: ISR_DTYPE_BF16_LOG2 (queries prescaled); False: plain bf16 (natural units) —
    both run corr_bf16_direct_kernel, with corr_bf16_kernel as the per-query fallback."""
    qb = ops.prescale_queries_log2(torch.from_numpy(Q)) if log2 else torch.from_numpy(Q).bfloat16()
    kb = torch.from_numpy(K).bfloat16()
    idx, logp, lse = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0), want_lse=True, log2_prescaled=log2)
    torch.cuda.synchronize()
    o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb), logit_scale=np.log(2.0) if log2 else 1.0)
    got = idx.cpu().numpy()
    bad = np.nonzero(got != o["idx"])[0]
    assert len(bad) == 0, f"{len(bad)} index mismatches, margins {(o['maxlogit'] - o['top2'])[bad][:5]}"
    scale = np.maximum(1.0, np.abs(o["lse"]))
    assert np.max(np.abs(logp.cpu().numpy() - (o["maxlogit"] - o["lse"]))) <= atol
    assert np.max(np.abs(lse.cpu().numpy() - o["lse"]) / scale) <= atol
    return o


@pytest.mark.parametrize("log2", [True, False])
def test_corr_bf16_log2_reference_bumps(cuda0, oracle_lib, log2):
    """Logits that keep growing along the key scan, maxima near +400 log2 units — far outside what the
    direct kernel's unreferenced f32 sum can hold: every query is marked bad and redone by the
    per-query-reference kernel, chunk by chunk (references of several hundred merged in finalize)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(21)
    P, N, D = 700, 6000, 64
    base = rng.normal(0, 1, (N, D))
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    K = (base * (2.0 + 14.0 * np.arange(N)[:, None] / N)).astype(np.float32)     # |k| grows 2 -> 16
    gt = rng.integers(N, size=P)
    Q = (18.0 * base[gt] + 0.5 * rng.normal(0, 1, (P, D))).astype(np.float32)
    o = _check_log2(ops, oracle_lib, cuda0, Q, K, atol=4e-5, log2=log2)
    assert o["maxlogit"].max() * np.log2(np.e) > 250.0


@pytest.mark.parametrize("log2", [True, False])
def test_corr_bf16_log2_mixed_fallback(cuda0, oracle_lib, log2):
    """Neighbouring queries whose maxima are hundreds of log2 units apart: the out-of-range ones are
    redone PER QUERY by the per-query-reference kernel, their in-range neighbours keep the direct
    result.  Also: very negative logits, and one huge late key (overflow inside one chunk)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(22)
    P, N, D = 1024, 3000, 64
    Q, K, gt = _planted(rng, P, N, D, tau=5.0)
    Q[256:512:2] *= 0.02            # wave-mates with maxima ~0.5 and ~36 nats: still one reference
    Q[512:768:2] *= 12.0            # ~430 nats next to ~36: flagged
    Q[768:] = -Q[768:] - 6.0 * np.sign(K).mean(0)    # everything far below zero
    K[2900] *= 40.0                 # a late key that dwarfs everything for queries aligned with it
    Q[100] = K[2900] / 40.0
    _check_log2(ops, oracle_lib, cuda0, Q, K, atol=2e-4, log2=log2)     # logits of +-600 log2 units: f32 ulp 6e-5


@pytest.mark.parametrize("log2", [True, False])
def test_corr_bf16_fallback_list_at_large_P(cuda0, oracle_lib, log2):
    """One key range, more 256-query blocks (1 368) than the fixed grids of the fallback / finalize kernels
    (1 024): both stride over the direct kernel's list of blocks that hold an out-of-range query.  Out-of-range
    queries are sprinkled over ~1 100 of the blocks (some blocks hold none), their neighbours stay in range."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(23)
    P, N, D = 350_000, 1500, 16
    Q, K, gt = _planted(rng, P, N, D, tau=4.0)
    hot = rng.choice(P, size=2200, replace=False)
    Q[hot[:1100]] *= 60.0                                   # maxima of ~ +900 log2 units: the plain f32 sum overflows
    Q[hot[1100:]] = -Q[hot[1100:]] * 30.0 - 8.0 * np.sign(K).mean(0)   # everything far below zero
    blocks = np.unique(hot // 256)
    assert 1024 < (P + 255) // 256 and 900 < len(blocks) < (P + 255) // 256
    _check_log2(ops, oracle_lib, cuda0, Q, K, atol=4e-4, log2=log2)    # logits up to +-1000 log2 units: f32 ulp 1e-4


@pytest.mark.parametrize("fill", ["ff", "neg", "zero"])
@pytest.mark.parametrize("log2", [True, False])
def test_corr_single_range_does_not_read_workspace_history(cuda0, oracle_lib, log2, fill):
    """One key range (N <= 4096) with out-of-range queries: corr_finalize_kernel finishes them and takes |q|^2
    for the margin test from the workspace.  The cached grow-only 'corr' workspace is filled with NaN bit
    patterns / negative floats / zeros before the call: the result must not depend on what an earlier call left
    there (round-2 advisor finding: the direct kernel stored |q|^2 only on the key-split route)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(29)
    P, N, D = 3000, 1500, 32
    Q, K, gt = _planted(rng, P, N, D, tau=4.0)
    hot = rng.choice(P, size=400, replace=False)
    Q[hot[:200]] *= 60.0
    Q[hot[200:]] = -Q[hot[200:]] * 30.0 - 8.0 * np.sign(K).mean(0)
    K[900:1400] = K[:500]                                   # exact ties too: bad queries that need the recheck
    ws = ops.workspace(cuda0, 64 << 20, "corr")
    if fill == "ff":
        ws.fill_(0xFF)
    elif fill == "neg":
        ws.view(torch.float32).fill_(-3.0e30)
    else:
        ws.zero_()
    _check_log2(ops, oracle_lib, cuda0, Q, K, atol=4e-4, log2=log2)


@pytest.mark.parametrize("kind", ["bf16_log2", "bf16", "f32"])
@pytest.mark.parametrize("N", [3000, 20000])
def test_corr_zero_queries_are_finished_without_a_recheck(cuda0, oracle_lib, kind, N):
    """Padding rows of a capacity-sized crop batch (isr_prep_queries_batch) are zero vectors: every logit is exactly 0,
    the arg-max is key 0, logp = -ln N.  A workgroup whose 256 queries are all zero leaves at once (one key range:
    N = 3 000; the f32 kernel on every route), a zero row inside a mixed workgroup runs the loop — both give the SAME
    bits, and neither puts the row on the exact-recheck list (round 3: the margin test with eps = 0 listed every
    padding row, 12 ms per 75 x 75 crop)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(31)
    P, D = 2048, 16
    Q, K, gt = _planted(rng, P, N, D, tau=4.0)
    Q[100:256] = 0.0                      # zero rows inside a mixed workgroup
    Q[512:1536] = 0.0                     # four whole workgroups of zero rows
    Q[2000:] = 0.0                        # the tail, ending in a partial workgroup
    zero = np.zeros(P, bool)
    zero[100:256] = zero[512:1536] = zero[2000:] = True
    if kind == "f32":
        idx, logp, lse = ops.corr_argmax(torch.from_numpy(Q).to(cuda0), torch.from_numpy(K).to(cuda0), want_lse=True)
        o = oracle_lib.corr_argmax_f32(Q, K)
        ref_logp, ref_lse = o["maxlogit"].astype(np.float64) - o["lse"], o["lse"]
    else:
        log2 = kind == "bf16_log2"
        qb = ops.prescale_queries_log2(torch.from_numpy(Q)) if log2 else torch.from_numpy(Q).bfloat16()
        kb = torch.from_numpy(K).bfloat16()
        idx, logp, lse = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0), want_lse=True, log2_prescaled=log2)
        o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb), logit_scale=np.log(2.0) if log2 else 1.0)
        ref_logp, ref_lse = o["maxlogit"] - o["lse"], o["lse"]
        assert ops.corr_recheck_count() < 20               # the planted rows only; 1 568 zero rows are not listed
    torch.cuda.synchronize()
    idx, logp, lse = idx.cpu().numpy(), logp.cpu().numpy(), lse.cpu().numpy()
    assert np.array_equal(idx, o["idx"]) and (idx[zero] == 0).all()
    np.testing.assert_allclose(logp, ref_logp, atol=3e-5)
    np.testing.assert_allclose(lse, ref_lse, atol=3e-5)
    # one value for every zero row, whichever route finished it: -ln N rounded once
    assert len(np.unique(logp[zero].view(np.int32))) == 1 and len(np.unique(lse[zero].view(np.int32))) == 1
    assert abs(float(logp[zero][0]) + np.log(float(N))) < 2e-6


@pytest.mark.parametrize("log2", [True, False])
def test_corr_bf16_log2_ties_lowest_key(cuda0, oracle_lib, log2):
    """Duplicate keys: every query has an exact tie — the margin test sends all of them to the exact
    recheck, where the lowest key index wins exactly as in the oracle."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(23)
    P, N, D = 300, 1500, 32
    Q, K, gt = _planted(rng, P, N, D, tau=5.0)
    K[700:1400] = K[:700]           # every key of the first 700 appears again 700 rows later
    o = _check_log2(ops, oracle_lib, cuda0, Q, K, log2=log2)
    assert (o["idx"] < 700).mean() > 0.8 and not ((o["idx"] >= 700) & (o["idx"] < 1400)).any()


@pytest.mark.parametrize("log2", [True, False])
def test_corr_full_size_properties(cuda0, oracle_lib, log2):
    """BASELINE configs[1] size (640x480 queries x 20 000 keys x 64-D), where the oracle cannot run
    the whole problem in seconds: (1) a 768-row sample against the oracle, indices bit-exact; (2) a
    query's result does not depend on the rest of the launch: the sample run on its own gives
    bit-identical idx and logp; (3) permuting the keys permutes the arg-max exactly;
    (4) log-probabilities are <= 0 up to rounding and the planted keys are recovered."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    P, N, D = 640 * 480, 20000, 64
    g = torch.Generator(device=cuda0).manual_seed(77)
    K = torch.randn(N, D, device=cuda0, generator=g)
    K = 6.0 * K / K.norm(dim=1, keepdim=True)
    gt = torch.randint(N, (P,), device=cuda0, generator=g)
    Qf = K[gt] + 0.5 * torch.randn(P, D, device=cuda0, generator=g)
    Q = ops.prescale_queries_log2(Qf) if log2 else Qf.bfloat16()
    Kb = K.bfloat16()
    idx, logp, lse = ops.corr_argmax(Q, Kb, want_lse=True, log2_prescaled=log2)
    assert (idx.long() == gt).float().mean().item() > 0.999
    assert logp.max().item() <= 0.0
    rows = torch.randperm(P, device=cuda0, generator=g)[:768].sort().values
    qs = Q[rows].contiguous()
    o = oracle_lib.corr_argmax_bf16(_bits(qs.cpu()), _bits(Kb.cpu()), logit_scale=np.log(2.0) if log2 else 1.0)
    assert np.array_equal(idx[rows].cpu().numpy(), o["idx"])
    np.testing.assert_allclose(logp[rows].cpu().numpy(), o["maxlogit"] - o["lse"], atol=3e-5)
    np.testing.assert_allclose(lse[rows].cpu().numpy(), o["lse"], rtol=2e-6, atol=3e-5)
    idx_s, logp_s = ops.corr_argmax(qs, Kb, log2_prescaled=log2)   # 768 rows: other wave-mates, split key range
    assert torch.equal(idx_s, idx[rows])
    assert torch.equal(logp_s, logp[rows])
    perm = torch.randperm(N, device=cuda0, generator=g)
    idx_p, logp_p = ops.corr_argmax(Q, Kb[perm].contiguous(), log2_prescaled=log2)
    # the exact arg-max is unique here (no duplicate keys): the permuted run must find the same key
    assert torch.equal(perm[idx_p.long()], idx.long())
    assert torch.allclose(logp_p, logp, rtol=0, atol=3e-5)      # another summation order: the path's tolerance


def test_corr_config5_stress_shape(cuda0, oracle_lib):
    """BASELINE configs[4] (K1 stress: 1024^2 queries x 200 000 keys x 128-D, log2 domain): the (P x N)
    matrix would be 839 GB; checks planted recovery, log-probabilities <= 0 and a 96-row oracle sample."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    P, N, D = 1024 * 1024, 200000, 128
    g = torch.Generator(device=cuda0).manual_seed(55)
    K = torch.randn(N, D, device=cuda0, generator=g)
    K = 6.0 * K / K.norm(dim=1, keepdim=True)
    gt = torch.randint(N, (P,), device=cuda0, generator=g)
    Q = ops.prescale_queries_log2(K[gt] + 0.3 * torch.randn(P, D, device=cuda0, generator=g))
    Kb = K.bfloat16()
    idx, logp, lse = ops.corr_argmax(Q, Kb, want_lse=True, log2_prescaled=True)
    assert (idx.long() == gt).float().mean().item() > 0.999
    assert logp.max().item() <= 0.0 and torch.isfinite(logp).all() and torch.isfinite(lse).all()
    rows = torch.randperm(P, device=cuda0, generator=g)[:96].sort().values
    o = oracle_lib.corr_argmax_bf16(_bits(Q[rows].cpu()), _bits(Kb.cpu()), logit_scale=np.log(2.0))
    assert np.array_equal(idx[rows].cpu().numpy(), o["idx"])
    np.testing.assert_allclose(logp[rows].cpu().numpy(), o["maxlogit"] - o["lse"], atol=3e-5)


@pytest.mark.parametrize("log2", [True, False])
@pytest.mark.parametrize("P,N,D", [(6000, 2500, 64), (5625, 80000, 16), (40000, 20000, 64), (3000, 9000, 128)])
def test_corr_result_is_independent_of_the_launch(cuda0, log2, P, N, D):
    """idx and logp of a query are functions of (query, keys) only.  Slices at odd offsets change a
    query's wave-mates and lane; small launches split the key range over workgroups, large ones do
    not; a query repeated next to out-of-range neighbours keeps its bits.  Everything must be
    torch.equal to the one big launch (inference.py:282-290: the strict `>` of the top-80 % cut makes
    every bit of logp matter)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(91 + P)
    K = torch.randn(N, D, device=cuda0, generator=g)
    K = 5.0 * K / K.norm(dim=1, keepdim=True)
    Qf = K[torch.randint(N, (P,), device=cuda0, generator=g)] + 0.35 * torch.randn(P, D, device=cuda0, generator=g)
    Qf[P // 2:] = torch.randn(P - P // 2, D, device=cuda0, generator=g)      # unplanted half: flat softmax, small margins
    Q = ops.prescale_queries_log2(Qf) if log2 else Qf.bfloat16()
    Kb = K.bfloat16()
    idx, logp = ops.corr_argmax(Q, Kb, log2_prescaled=log2)
    for a, b in [(0, 1), (1, 2), (7, 300), (P // 2 - 33, P // 2 + 190), (P - 257, P), (13, P - 5)]:
        i1, l1 = ops.corr_argmax(Q[a:b].contiguous(), Kb, log2_prescaled=log2)
        assert torch.equal(i1, idx[a:b]) and torch.equal(l1, logp[a:b]), (a, b)
    # the same rows tiled into a launch 8x as large (no key split any more, other workgroups)
    rep = Q.repeat(8, 1)
    i8, l8 = ops.corr_argmax(rep, Kb, log2_prescaled=log2)
    assert torch.equal(i8.reshape(8, P), idx.expand(8, P)) and torch.equal(l8.reshape(8, P), logp.expand(8, P))
    # out-of-range neighbours (logits of several hundred log2 units) in the same waves
    mixed = Q[:512].clone()
    mixed[1::2] = (Q[1:512:2].float() * 40.0).bfloat16()
    im, lm = ops.corr_argmax(mixed, Kb, log2_prescaled=log2)
    assert torch.equal(im[0::2], idx[0:512:2]) and torch.equal(lm[0::2], logp[0:512:2])


def test_corr_randomised_shapes_distributions_and_scales(cuda0, oracle_lib):
    """A slice of tools/stress_corr.py (60 cases there): random P, N, D, logit scale, descriptor distribution
    (Gaussian, planted, duplicate keys, sparse) and path (bf16 / bf16-log2).  Indices equal the oracle's exactly,
    logp to 5e-5 of the logit scale, and a random sub-launch reproduces its rows bit for bit."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(2)
    for c in range(14):
        D = int(rng.choice([16, 32, 64, 128]))
        P, N = int(rng.integers(1, 2500)), int(rng.integers(1, 25000))
        scale = float(rng.choice([0.05, 0.5, 1.0, 3.0, 8.0, 25.0]))
        kind = rng.choice(["gauss", "planted", "dups", "sparse"])
        K = rng.normal(0, 1, (N, D)).astype(np.float32)
        if kind == "sparse":
            K *= rng.uniform(size=(N, D)) < 0.2
        if kind == "dups" and N > 4:
            K[N // 2:] = K[: N - N // 2]
        Q = rng.normal(0, 1, (P, D)).astype(np.float32)
        if kind == "planted":
            Q = K[rng.integers(N, size=P)] + 0.3 * Q
        Q *= scale / np.sqrt(D) * (1 + 3 * (rng.uniform(size=(P, 1)) < 0.05))
        log2 = bool(rng.integers(2))
        qb = ops.prescale_queries_log2(torch.from_numpy(Q)) if log2 else torch.from_numpy(Q).bfloat16()
        kb = torch.from_numpy(K).bfloat16()
        idx, logp = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0), log2_prescaled=log2)
        o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb), logit_scale=np.log(2.0) if log2 else 1.0)
        assert np.array_equal(idx.cpu().numpy(), o["idx"]), (c, kind, P, N, D, scale, log2)
        err = np.abs(logp.cpu().numpy() - (o["maxlogit"] - o["lse"])) / np.maximum(1.0, np.abs(o["maxlogit"]) * 1e-1)
        assert err.max() < 5e-5, (c, kind, float(err.max()))
        a = int(rng.integers(0, P))
        b = int(rng.integers(a + 1, P + 1))
        i2, l2 = ops.corr_argmax(qb[a:b].contiguous().to(cuda0), kb.to(cuda0), log2_prescaled=log2)
        assert torch.equal(i2, idx[a:b]) and torch.equal(l2, logp[a:b]), (c, kind, a, b)


@pytest.mark.parametrize("P,N,D", [(300, 31, 12), (1000, 4097, 12), (777, 12289, 5), (5000, 20000, 16), (64, 128, 8),
                                   (20000, 80000, 12), (513, 4096, 13)])
def test_corr_f32_split_route_equals_chain_route(cuda0, oracle_lib, P, N, D):
    """f32 descriptors with D <= 16 run on the bf16 matrix cores (three-way bf16 split of every f32 number, six plane pairs as one
    96-wide bf16 dot product, log2-domain direct kernel) with an f32-chain exact recheck: the indices are those of the f32-MFMA
    chain kernel (ISR_TUNE_K1_F32_CHAIN) and of the oracle — ties to the lowest key, duplicates across a chunk boundary, zero
    rows, a spiked and a negated query — logp / lse agree to a few 1e-7 relative to the logit scale, and a slice at an odd
    offset reproduces its rows bit for bit."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(P + N + D)
    K = torch.randn(N, D, device=cuda0, generator=g) * 2.0
    K[N // 2] = K[0]
    if N > 4200:
        K[4100] = K[3]
    gt = torch.randint(N, (P,), device=cuda0, generator=g)
    Q = K[gt] * 1.5 + 0.5 * torch.randn(P, D, device=cuda0, generator=g)
    Q[::7] = 0.0
    Q[1] *= 20.0
    Q[2] = -Q[2]
    Q[3] = K[0] * 1.5                                                      # exact tie between key 0 and its duplicate
    with ops.tuning(k1_f32_chain=1):
        want = ops.corr_argmax(Q, K, want_lse=True)
    got = ops.corr_argmax(Q, K, want_lse=True)
    assert torch.equal(got[0], want[0])
    scale = 1.0 + float(want[2].abs().max())
    assert float((got[1] - want[1]).abs().max()) < 3e-6 * scale and float((got[2] - want[2]).abs().max()) < 3e-6 * scale
    o = oracle_lib.corr_argmax_f32(Q[:256].cpu().numpy(), K.cpu().numpy())
    assert np.array_equal(got[0][:256].cpu().numpy(), o["idx"])
    if P > 300:
        # launch independence, for logits inside the direct kernel's range (|log2-unit logit| < 128, as on the bf16 paths: an
        # out-of-range query is still decided exactly, but which of its key ranges are redone with a per-query reference — and
        # with that the last bit of its logp — follows the launch's key split)
        Kn = 5.0 * K / K.norm(dim=1, keepdim=True)
        Qs = 1.2 * Kn[gt] + 0.3 * torch.randn(P, D, device=cuda0, generator=g)
        Qs[::5] = 0.0
        full = ops.corr_argmax(Qs, Kn, want_lse=True)
        assert float(full[2].max()) * 1.4427 < 100.0
        K = Kn
        for lo, n in ((37, 200), (0, 300)):
            sl = ops.corr_argmax(Qs[lo:lo + n], K, want_lse=True)
            for a, b in zip(sl, full):
                assert torch.equal(a, b[lo:lo + n])


@pytest.mark.parametrize("P,N,D,chain", [(300, 31, 17, 0), (1000, 4097, 24, 0), (777, 12289, 33, 0), (5000, 20000, 64, 0),
                                         (513, 4096, 40, 0), (6000, 33000, 64, 0), (2000, 9000, 32, 0), (64, 128, 48, 0),
                                         (1000, 4097, 12, 0), (5000, 20000, 16, 0), (777, 12289, 5, 0),
                                         (1000, 4097, 24, 2), (5000, 20000, 64, 2), (777, 12289, 33, 2), (2000, 9000, 32, 2),
                                         (1000, 4097, 12, 2), (5000, 20000, 16, 2), (777, 12289, 5, 4),
                                         (1000, 4097, 65, 0), (5000, 20000, 128, 0), (777, 12289, 100, 0), (300, 31, 96, 0)])
def test_corr_f32_three_plane_route_equals_chain_route(cuda0, oracle_lib, P, N, D, chain):
    """f32 descriptors on the 16-bit matrix cores (round 4), both plane forms (RowFrags in corr_argmax.hip):
    chain = 0, the default — f16 planes x1 | x2s | x1s, three plane pairs per 16-wide block, D <= 128 (above 64: one wave per
    SIMD with the whole register file, dynamic LDS); chain = 2 — bf16 planes (D <= 64)
    x1 | x2 | x3, six plane pairs; chain = 4 — round 3's 96-wide rows (D <= 16).  Each with its margin test and the recheck
    by the f32 fmaf chain of the original rows.  Indices = the f32-MFMA chain kernel's (ISR_TUNE_K1_F32_CHAIN = 1) = the
    oracle's, with duplicate keys (one pair across a canonical chunk boundary), an exact tie, zero rows (a whole workgroup
    of them), a spiked and a negated query; logp / lse within 3e-6 of the chain kernel's relative to the logit scale;
    slices at odd offsets and forced key-range counts reproduce their rows bit for bit."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(P + N + D)
    K = torch.randn(N, D, device=cuda0, generator=g) * (2.0 * (12.0 / max(D, 12)) ** 0.5)      # |k|^2 ~ 48 whatever D
    K[N // 2] = K[0]
    if N > 4200:
        K[4100] = K[3]
    gt = torch.randint(N, (P,), device=cuda0, generator=g)
    Q = K[gt] * 1.5 + 0.5 * (12.0 / max(D, 12)) ** 0.5 * torch.randn(P, D, device=cuda0, generator=g)
    Q[::7] = 0.0
    Q[1] *= 20.0
    Q[2] = -Q[2]
    Q[3] = K[0] * 1.5                                                      # exact tie between key 0 and its duplicate
    if P >= 1000:
        Q[512:768] = 0.0                                                   # one whole workgroup of zero rows
    with ops.tuning(k1_f32_chain=1):
        want = ops.corr_argmax(Q, K, want_lse=True)
    with ops.tuning(k1_f32_chain=chain):
        got = ops.corr_argmax(Q, K, want_lse=True)
        rechecked = ops.corr_recheck_count_f32(D)
    assert torch.equal(got[0], want[0])
    assert 0 <= rechecked and (N < 4000 or rechecked < max(64, P // 4))     # the margin test certifies the bulk (few keys: duplicates tie)
    scale = 1.0 + float(want[2].abs().max())
    assert float((got[1] - want[1]).abs().max()) < 3e-6 * scale and float((got[2] - want[2]).abs().max()) < 3e-6 * scale
    o = oracle_lib.corr_argmax_f32(Q[:256].cpu().numpy(), K.cpu().numpy())
    assert np.array_equal(got[0][:256].cpu().numpy(), o["idx"])
    if P > 300:
        Kn = 5.0 * K / K.norm(dim=1, keepdim=True)
        Qs = 1.2 * Kn[gt] + 0.3 * (12.0 / max(D, 12)) ** 0.5 * torch.randn(P, D, device=cuda0, generator=g)
        Qs[::5] = 0.0
        with ops.tuning(k1_f32_chain=chain):
            full = ops.corr_argmax(Qs, Kn, want_lse=True)
            assert float(full[2].max()) * 1.4427 < 100.0
            for lo, n in ((37, 200), (0, 300)):
                sl = ops.corr_argmax(Qs[lo:lo + n], Kn, want_lse=True)
                for a, b in zip(sl, full):
                    assert torch.equal(a, b[lo:lo + n])
            for ns in (1, 2, 3):
                with ops.tuning(k1_split=ns):
                    sp = ops.corr_argmax(Qs, Kn, want_lse=True)
                for a, b in zip(sp, full):
                    assert torch.equal(a, b), ns
        with ops.tuning(k1_f32_chain=1):
            ref = ops.corr_argmax(Qs, Kn, want_lse=True)
        assert torch.equal(full[0], ref[0])


@pytest.mark.parametrize("D", [12, 32])
def test_corr_f32_chain_kernel_all_zero_workgroups(cuda0, oracle_lib, D):
    """corr_f32_kernel's early exit for a workgroup whose 256 queries are all zero (ISR_TUNE_K1_F32_CHAIN = 1: since the split
    routes became the default nothing else reaches it): with one key range and with several (ISR_TUNE_K1_SPLIT 1, 2, 3) the
    zero workgroups' outputs — key 0, logp = lse-consistent -ln N — are bit for bit those of a zero row inside a mixed
    workgroup, and the non-zero rows are the oracle's."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(40 + D)
    P, N = 1536, 9000
    K = torch.randn(N, D, device=cuda0, generator=g) * (12.0 / D) ** 0.5
    Q = 1.5 * K[torch.randint(N, (P,), device=cuda0, generator=g)] + 0.3 * torch.randn(P, D, device=cuda0, generator=g)
    Q[256:768] = 0.0                     # two whole workgroups
    Q[1000] = 0.0                        # a zero row in a mixed workgroup
    outs = []
    for ns in (1, 2, 3):
        with ops.tuning(k1_f32_chain=1, k1_split=ns):
            outs.append(ops.corr_argmax(Q, K, want_lse=True))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)
    idx, logp, lse = outs[0]
    assert int(idx[256:768].abs().max()) == 0 and int(idx[1000]) == 0
    assert torch.equal(logp[256:768], logp[1000].expand(512)) and torch.equal(lse[256:768], lse[1000].expand(512))
    assert abs(float(logp[1000]) + np.log(N)) < 1e-5 and abs(float(lse[1000]) - np.log(N)) < 1e-5
    rows = np.r_[0:64, 250:262, 760:776, 995:1005]
    o = oracle_lib.corr_argmax_f32(Q[rows].cpu().numpy(), K.cpu().numpy())
    assert np.array_equal(idx[rows].cpu().numpy(), o["idx"])


@pytest.mark.parametrize("D,chain", [(64, 0), (64, 2), (32, 0), (12, 0), (12, 2), (128, 0)])
def test_corr_f32_plane_routes_repeat_bit_for_bit(cuda0, D, chain):
    """Six calls on a chip-filling shape (the keys reach the LDS by buffer_load ... lds, stage s + 2 in flight under stage
    s + 1: a missing wait between a DMA piece landing and the barrier that publishes it shows as outputs that change from
    call to call — round 4's first f16 build did exactly that at this size and passed every small-shape test)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(3 * D + chain)
    P, N = 150000, 20000
    K = torch.randn(N, D, device=cuda0, generator=g)
    K = 5.0 * K / K.norm(dim=1, keepdim=True)
    Q = K[torch.randint(N, (P,), device=cuda0, generator=g)] + 0.3 * (12.0 / max(D, 12)) ** 0.5 * torch.randn(P, D, device=cuda0, generator=g)
    with ops.tuning(k1_f32_chain=chain):
        first = ops.corr_argmax(Q, K, want_lse=True)
        for _ in range(5):
            again = ops.corr_argmax(Q, K, want_lse=True)
            for a, b in zip(again, first):
                assert torch.equal(a, b)
    with ops.tuning(k1_f32_chain=1):
        ref = ops.corr_argmax(Q, K, want_lse=True)
    assert torch.equal(first[0], ref[0])
    assert float((first[1] - ref[1]).abs().max()) < 2e-5 and float((first[2] - ref[2]).abs().max()) < 2e-4


@pytest.mark.parametrize("kind,P,N,D,chain", [("f32", 50176, 80000, 12, 0), ("f32", 5476, 80000, 12, 0), ("f32", 3000, 10960, 12, 0),
                                              ("f32", 4000, 9000, 40, 0), ("f32", 4000, 9000, 12, 2), ("f32", 4000, 9000, 12, 1),
                                              ("f32", 4000, 9000, 12, 4), ("bf16_log2", 6000, 20000, 64, 0), ("bf16", 3000, 5000, 32, 0),
                                              ("f32", 300, 31, 12, 0)])
def test_corr_lse_only_call_returns_the_full_call_s_lse(cuda0, kind, P, N, D, chain):
    """isr_corr_argmax with idx = logp = NULL (ops.corr_lse: pose_refine.py:56's denominator image, estimate_pose's row sums):
    the LSE instantiations track no maxima, recover no rows and recheck nothing, queries whose Cauchy-Schwarz bound
    |q||k|_max leaves the direct sum's range are redone with a per-query reference as bad ones are — and the values are
    torch.equal to the lse output of the full call, on every route (routes without an LSE instantiation run the full kernels
    with the index writes off), with zero rows, out-of-range rows, one key range and several."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(P + N + D)
    K = torch.randn(N, D, device=cuda0, generator=g)
    K = 5.0 * K / K.norm(dim=1, keepdim=True)
    Q = 1.2 * K[torch.randint(N, (P,), device=cuda0, generator=g)] + 0.3 * (12.0 / max(D, 12)) ** 0.5 * torch.randn(P, D, device=cuda0, generator=g)
    Q[::9] = 0.0
    Q[5] *= 30.0                                   # far outside the direct sum's range: the per-query fallback
    Q[7] *= 12.0                                   # fails the bound (|q||k| > 99 log2 units) while its logits would have fitted
    Q[11] = -8.0 * K[11]                           # every logit far below zero except none: maximum below kLow
    if P >= 3000:
        Q[1024:1280] = 0.0
    log2 = kind == "bf16_log2"
    if kind != "f32":
        Q, K = (ops.prescale_queries_log2(Q) if log2 else Q.bfloat16()), K.bfloat16()
    for ns in ((0, 1, 2, 3) if P >= 3000 else (0,)):
        with ops.tuning(k1_f32_chain=chain, k1_split=ns):
            full = ops.corr_argmax(Q, K, want_lse=True, log2_prescaled=log2)[2]
            only = ops.corr_lse(Q, K, log2_prescaled=log2)
        assert torch.equal(only, full), (ns, int((only != full).sum()))
    ref = torch.logsumexp((Q.double() / (ops.LOG2E if log2 else 1.0)) @ K.double().T, dim=-1)
    assert float((only.double() - ref).abs().max()) < (2e-5 if kind == "f32" else 2e-2) * (1.0 + float(ref.abs().max()) / 50.0)


@pytest.mark.parametrize("P,N,D,k", [(5625, 80000, 12, 3), (300, 31, 12, 8), (1000, 4097, 40, 5), (257, 9000, 100, 2), (64, 5, 12, 8),
                                     (2000, 20000, 64, 1)])
def test_corr_topk_without_the_matrix(cuda0, P, N, D, k):
    """getCors(leaves = k) through isr_corr_topk (the (P, N) matrix never exists) against torch.topk on the materialised
    log-softmax matrix of isr_corr_logsoftmax — the round-3 route, whose elements are the same k-ordered fmaf chains minus
    the same lse: indices equal wherever the matrix's k + 1 largest entries of a row are distinct, values to 2e-6; duplicate
    keys (equal values: the lower key first), zero rows (every value equal: keys 0 .. k - 1), N < k (-1 beyond the keys)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, registration
    g = torch.Generator(device=cuda0).manual_seed(P + N + D + k)
    K = torch.randn(N, D, device=cuda0, generator=g)
    K = 5.0 * K / K.norm(dim=1, keepdim=True)
    if N > 20:
        K[N // 2] = K[0]
        K[N - 1] = K[0]
    Q = 1.2 * K[torch.randint(N, (P,), device=cuda0, generator=g)] + 0.3 * (12.0 / max(D, 12)) ** 0.5 * torch.randn(P, D, device=cuda0, generator=g)
    Q[::11] = 0.0
    Q[3] = 1.5 * K[0]                                   # three equal winners: keys 0, N // 2, N - 1 in that order
    idx, vals = ops.corr_topk(Q, K, k)
    kk = min(k, N)
    cMat = ops.corr_logsoftmax(Q, K)
    rv, ri = torch.topk(cMat, k=min(kk + 1, N), dim=-1)
    # rows whose k + 1 best matrix entries are pairwise distinct have one right answer
    distinct = (rv[:, :-1] != rv[:, 1:]).all(dim=1) if rv.shape[1] > 1 else torch.ones(P, dtype=torch.bool, device=cuda0)
    assert int(distinct.sum()) > 0.7 * P - 30
    assert torch.equal(idx[distinct][:, :kk].long(), ri[distinct][:, :kk])
    assert float((vals[:, :kk] - rv[:, :kk]).abs().max()) < 2e-6 * (1.0 + float(rv[:, :kk].abs().max()))
    if N < k:
        assert bool((idx[:, N:] == -1).all())
    # ties: by ascending key
    assert idx[0, :kk].tolist() == list(range(kk))                                  # a zero row
    if N > 20 and k >= 3:
        assert idx[3, :3].tolist() == [0, N // 2, N - 1]
    # the reference's call shape
    ci, cv = registration.getCors(Q, K, leaves=k)
    assert ci.dtype == torch.int64 and not ci.is_cuda and cv.is_cuda
    if k > 1:
        assert tuple(ci.shape) == (P, k) and torch.equal(ci, idx.long().cpu()) and torch.equal(cv, vals)


@pytest.mark.parametrize("where", ["query", "key"])
def test_corr_f32_f16_planes_fall_through_when_a_descriptor_does_not_fit_f16(cuda0, oracle_lib, where):
    """The default f32 route keeps f16 planes; an |x| >= 65 000 anywhere raises the gate word in its split kernel (finite
    descriptors are a precondition of K1 on every route: corr_argmax.hip is built with -fno-honor-nans),
    every kernel of the route leaves at once and the f32-MFMA chain kernels queued behind it produce the call's outputs —
    the bits of an ISR_TUNE_K1_F32_CHAIN = 1 call, with no host round trip in between."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(77)
    P, N, D = 1500, 9000, 24
    K = torch.randn(N, D, device=cuda0, generator=g) * 0.7
    Q = 1.5 * K[torch.randint(N, (P,), device=cuda0, generator=g)] + 0.2 * torch.randn(P, D, device=cuda0, generator=g)
    if where == "query":
        Q[17, 3] = 7.0e4 / 1.4426950408889634 * 1.01           # the query planes hold q log2 e
    else:
        K[8000, 5] = -1.0e5
    with ops.tuning(k1_f32_chain=1):
        want = ops.corr_argmax(Q, K, want_lse=True)
    got = ops.corr_argmax(Q, K, want_lse=True)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    clean = torch.arange(100, 164, device=cuda0)
    o = oracle_lib.corr_argmax_f32(Q[clean].cpu().numpy(), K.cpu().numpy())
    if where != "key":                                          # the oracle's softmax overflows with the 1e5 key, the indices do not
        assert np.array_equal(got[0][clean].cpu().numpy(), o["idx"])


@pytest.mark.parametrize("kind", ["bf16_log2", "f32"])
def test_corr_result_is_independent_of_the_number_of_key_ranges(cuda0, kind):
    """The planner's choice of key ranges (cost model; ISR_TUNE_K1_SPLIT forces it) changes which workgroup computes a chunk,
    never a chunk's arithmetic: idx, logp and lse bit for bit for 1, 2, 3, 5 and the automatic number of ranges — P large
    enough that one range is a legal plan, zero rows (crop-batch padding) included."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    g = torch.Generator(device=cuda0).manual_seed(5)
    P, N, D = 6000, 33000, 12 if kind == "f32" else 16
    K = torch.randn(N, D, device=cuda0, generator=g)
    K = 5.0 * K / K.norm(dim=1, keepdim=True)
    Q = 1.2 * K[torch.randint(N, (P,), device=cuda0, generator=g)] + 0.3 * torch.randn(P, D, device=cuda0, generator=g)
    Q[1000:3500] = 0.0
    if kind == "bf16_log2":
        Q, K = ops.prescale_queries_log2(Q), K.bfloat16()
    call = lambda: ops.corr_argmax(Q, K, want_lse=True, log2_prescaled=(kind == "bf16_log2"))
    auto = call()
    for ns in (1, 2, 3, 5):
        with ops.tuning(k1_split=ns):
            got = call()
        for a, b in zip(got, auto):
            assert torch.equal(a, b), ns


@pytest.mark.parametrize("log2", [True, False])
@pytest.mark.parametrize("D", [16, 64, 128])
def test_corr_screened_maxima_near_ties_in_skipped_looking_tiles(cuda0, oracle_lib, log2, D):
    """The sum screen of the direct kernel (corr_direct.hpp, ISR_K1_SCREEN): a tile enters the maxima only when the sum of
    its exponentials reaches 2^(maximum so far - dlt).  Adversarial layout: a dominant key early in the stream (every later
    tile of plain keys is skipped), and far behind it — alone among near-zero keys — keys whose EXACT logit differs from
    the dominant one's by one bf16 ulp of one coordinate, by nothing (a duplicate), or by less than the MFMA's accumulation
    error, above and below.  The exact arg-max (lowest key on ties) must come out, through the recheck where the f32
    logits cannot tell."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(100 + D + (1 if log2 else 0))
    P, N = 384, 12000
    K = rng.normal(0, 0.02, (N, D)).astype(np.float32)                  # near-zero keys: logits ~ 0, tile sums ~ 16
    Q = rng.normal(0, 1.0, (P, D)).astype(np.float32)
    Q *= 6.0 / np.linalg.norm(Q, axis=1, keepdims=True)
    qb = torch.from_numpy(Q).bfloat16()
    Qr = qb.float().numpy()
    dom = Qr[:96] * (5.0 / 6.0)                                          # 96 dominant keys, parallel to queries 0..95 (logit ~ 30); the other queries see them at random angles
    pos_dom = 40 + 37 * np.arange(96)                                    # early keys, scattered over tiles
    K[pos_dom] = dom
    kb = torch.from_numpy(K).bfloat16()
    Kr = kb.float().numpy()
    # rivals far behind: copies of the rounded dominant key with a one-ulp nudge of one coordinate (up, down, none)
    pos_riv = 6000 + 61 * np.arange(96)
    riv = Kr[pos_dom].copy()
    kind = np.arange(96) % 3
    for j in range(96):
        c = int(rng.integers(D))
        v = torch.tensor(riv[j, c]).bfloat16()
        bits = v.view(torch.int16).item()
        if kind[j] == 1: bits += 1 if bits >= 0 else -1                 # one ulp away from zero
        if kind[j] == 2: bits -= 1 if bits > 0 else -1                  # one ulp towards zero
        riv[j, c] = torch.tensor(bits, dtype=torch.int16).view(torch.bfloat16).float().item()
    Kr[pos_riv] = riv
    Kr[3] = Kr[pos_dom[5]]                                               # and an exact duplicate IN FRONT of a dominant key
    kb = torch.from_numpy(Kr).bfloat16()
    if log2:
        ql = ops.prescale_queries_log2(torch.from_numpy(Qr).to(cuda0))
        idx, logp = ops.corr_argmax(ql, kb.to(cuda0), log2_prescaled=True)
        o = oracle_lib.corr_argmax_bf16(_bits(ql.cpu()), _bits(kb), logit_scale=np.log(2.0))
    else:
        idx, logp = ops.corr_argmax(qb.to(cuda0), kb.to(cuda0))
        o = oracle_lib.corr_argmax_bf16(_bits(qb), _bits(kb))
    torch.cuda.synchronize()
    got = idx.cpu().numpy()
    assert np.array_equal(got, o["idx"])
    # the layout does what it says: winners are dominant keys, their rivals or the duplicate in front
    assert np.isin(got, np.concatenate([pos_dom, pos_riv, [3]])).all()
    assert (np.isin(got, pos_riv)).sum() > 0 and (np.isin(got, pos_dom)).sum() > 0
    assert ops.corr_recheck_count() > 0
    np.testing.assert_allclose(logp.cpu().numpy(), o["maxlogit"] - o["lse"], atol=3e-5)


@pytest.mark.parametrize("D,chain", [(12, 0), (12, 2), (32, 0), (32, 2), (64, 0), (100, 0), (128, 0)])
def test_corr_f32_screened_maxima_near_ties_in_skipped_looking_tiles(cuda0, oracle_lib, D, chain):
    """The same adversarial layout on the f32 plane routes (the screen is compiled in at SP = 1, 2, 8: D <= 32 and D > 64):
    rivals one f32 ulp away from a dominant key, far behind it among near-zero keys; the f32 fmaf-chain arg-max of the oracle
    (lowest key on ties) must come out."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(300 + D + chain)
    P, N = 384, 12000
    K = rng.normal(0, 0.02, (N, D)).astype(np.float32)
    Q = rng.normal(0, 1.0, (P, D)).astype(np.float32)
    Q *= (6.0 / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    pos_dom = 40 + 37 * np.arange(96)
    pos_riv = 6000 + 61 * np.arange(96)
    K[pos_dom] = Q[:96] * np.float32(5.0 / 6.0)
    riv = K[pos_dom].copy()
    kind = np.arange(96) % 3
    for j in range(96):
        c = int(rng.integers(D))
        if kind[j] == 1: riv[j, c] = np.nextafter(riv[j, c], np.float32(np.inf) * np.sign(riv[j, c]))
        if kind[j] == 2: riv[j, c] = np.nextafter(riv[j, c], np.float32(0))
    K[pos_riv] = riv
    K[3] = K[pos_dom[5]]
    with ops.tuning(k1_f32_chain=chain):
        idx, logp = ops.corr_argmax(torch.from_numpy(Q).to(cuda0), torch.from_numpy(K).to(cuda0))
        torch.cuda.synchronize()
    o = oracle_lib.corr_argmax_f32(Q, K)
    got = idx.cpu().numpy()
    assert np.array_equal(got, o["idx"])
    assert np.isin(got, np.concatenate([pos_dom, pos_riv, [3]])).all()
    assert (np.isin(got, pos_riv)).sum() > 0 and (np.isin(got, pos_dom)).sum() > 0
    np.testing.assert_allclose(logp.cpu().numpy(), o["maxlogit"].astype(np.float64) - o["lse"], atol=3e-5)


@pytest.mark.gpu
def test_mfma_f16_keeps_subnormal_inputs(cuda0, tmp_path):
    """split_eabs (corr_argmax.hip) prices subnormal x1s plane elements at 2^-25 ABSOLUTE each, which is true only if
    v_mfma_f32_32x32x16_f16 keeps subnormal f16 INPUTS (MI200 flushed them).  The probe in tools/mfma_f16_denorm.hip is
    built and run here (the claim used to rest on a committed binary)."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mfma_f16_denorm.hip")
    exe = str(tmp_path / "mfma_f16_denorm")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-w", "-o", exe, src], check=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout.splitlines()
    kept = float(out[0].split(":")[1].split()[0])          # 16 products of 2^-20 (subnormal) x 2^10
    tiny = float(out[1].split(":")[1].split()[0])          # 16 products of 2^-14 x 2^-24 (the smallest subnormal)
    assert kept == 2.0 ** -6                                 # (the probe prints nine significant digits)
    assert tiny == pytest.approx(16.0 * 2.0 ** -38, rel=1e-8)


@pytest.mark.gpu
def test_corr_f32_full_size_properties(cuda0, oracle_lib):
    """The reference's precision at BASELINE configs[1]'s full size: 640 x 480 f32 queries of 64 dimensions against
    20 000 f32 keys (inference.py:142-149 is an f32 matmul) on the default f16-plane route.  768 sampled rows against the C
    oracle's k-ordered f32 chain (indices array_equal, logp 3e-5); the whole index vector torch.equal to the f32-MFMA chain
    kernel's (ISR_TUNE_K1_F32_CHAIN = 1); a slice computed in a launch of its own carries the same bits."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(64)
    P, N, D = 307200, 20000, 64
    Q, K, gt = _planted(rng, P, N, D, tau=5.0)
    q, k = torch.from_numpy(Q).to(cuda0), torch.from_numpy(K).to(cuda0)
    idx, logp, lse = ops.corr_argmax(q, k, want_lse=True)
    torch.cuda.synchronize()
    assert ops.corr_recheck_count_f32(D) >= 0                      # the plane route ran (not the fall-through)
    rows = rng.choice(P, 768, replace=False)
    o = oracle_lib.corr_argmax_f32(Q[rows], K)
    assert np.array_equal(idx.cpu().numpy()[rows], o["idx"])
    np.testing.assert_allclose(logp.cpu().numpy()[rows], o["maxlogit"].astype(np.float64) - o["lse"], atol=3e-5)
    np.testing.assert_allclose(lse.cpu().numpy()[rows], o["lse"], rtol=2e-6, atol=3e-5)
    with ops.tuning(k1_f32_chain=1):
        idx_c, logp_c = ops.corr_argmax(q, k)
    assert torch.equal(idx, idx_c)
    assert float((logp - logp_c).abs().max()) < 3e-5
    part = ops.corr_argmax(q[200000:200900].contiguous(), k, want_lse=True)
    assert torch.equal(part[0], idx[200000:200900]) and torch.equal(part[1], logp[200000:200900]) and torch.equal(part[2], lse[200000:200900])
