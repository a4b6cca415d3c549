"""Pins the C oracle's getCors restatement against the literal reference expression
(inference.py:142-149): torch.log_softmax(q @ f.T, -1) + torch.topk(k=1), on torch-CPU."""
import numpy as np
import torch


def _ref_getcors(q, f):
    cMat = torch.log_softmax(q @ f.T, dim=-1)
    vals, idx = torch.topk(cMat, k=1, dim=-1)
    return idx[..., 0], vals


def _planted(rng, P, N, D, tau=8.0, sigma=0.35):
    K = rng.normal(0, 1, (N, D))
    K *= tau / np.linalg.norm(K, axis=1, keepdims=True)     # |k| = tau: the planted key wins the dot product
    gt = rng.integers(N, size=P)
    Q = K[gt] + sigma * rng.normal(0, 1, (P, D))
    return Q.astype(np.float32), K.astype(np.float32), gt


def test_corr_f32_matches_torch_literal(oracle_lib):
    rng = np.random.default_rng(0)
    Q, K, gt = _planted(rng, 700, 5000, 12)       # reference shape: D = 12
    o = oracle_lib.corr_argmax_f32(Q, K)
    idx, vals = _ref_getcors(torch.from_numpy(Q), torch.from_numpy(K))
    assert np.array_equal(o["idx"], idx.numpy())
    logp = o["maxlogit"].astype(np.float64) - o["lse"]
    np.testing.assert_allclose(logp, vals[:, 0].numpy(), atol=2e-5)
    assert (o["idx"] == gt).mean() > 0.95


def test_corr_bf16_matches_torch_on_rounded_inputs(oracle_lib):
    rng = np.random.default_rng(1)
    Q, K, gt = _planted(rng, 300, 3000, 64)
    qb, kb = torch.from_numpy(Q).bfloat16(), torch.from_numpy(K).bfloat16()
    o = oracle_lib.corr_argmax_bf16(qb.view(torch.int16).numpy().view(np.uint16),
                                    kb.view(torch.int16).numpy().view(np.uint16))
    idx, vals = _ref_getcors(qb.double(), kb.double())   # same rounded inputs, f64 reference
    assert np.array_equal(o["idx"], idx.numpy())
    np.testing.assert_allclose(o["maxlogit"] - o["lse"], vals[:, 0].numpy(), atol=1e-9)


def test_corr_tie_lowest_index(oracle_lib):
    K = np.zeros((10, 4), np.float32)
    K[3] = K[7] = [1, 0, 0, 0]
    Q = np.array([[2, 0, 0, 0], [-1, 0, 0, 0]], np.float32)
    o = oracle_lib.corr_argmax_f32(Q, K)
    assert o["idx"].tolist() == [3, 0]
