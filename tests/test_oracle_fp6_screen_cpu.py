"""CPU: the inequality K1's FP6 screen rests on (csrc/corr_sparse.hpp, getCors of inference.py:142-149 behind
ISR_DTYPE_BF16_LOG2_SCREENED), on the oracle's restatement of the quantiser and of the bound:
    |<q, k> - <q~, k~>| <= E_q = |q| max|k - k~| + |q - q~| max|k~| (+ accumulation terms)
for random rows over many scales, rows with huge dynamic range inside a block, zero rows — and rows built to make the
Cauchy-Schwarz step an equality (every key element at a rounding midpoint, the errors' signs aligned with the query's)."""
import numpy as np

from oracle import fp6_screen_oracle as fo


def _rows(rng, R, scale):
    return fo.bf16_round((rng.normal(0, 1, (R, 64)) * scale).astype(np.float32))


def _check(Q, K):
    q, k = fo.quantize_e2m3(Q), fo.quantize_e2m3(K)
    S = Q.astype(np.float64) @ K.astype(np.float64).T
    St = q["deq"].astype(np.float64) @ k["deq"].astype(np.float64).T
    E = fo.screen_error(q["nrm"][:, 0], q["nrm"][:, 1], k["d2"].max(), k["t2"].max())
    slack = E[:, None].astype(np.float64) - np.abs(S - St)
    assert slack.min() >= 0.0, slack.min()
    return np.abs(S - St).max(), E.max()


def test_quantiser_values_are_e2m3_numbers_within_half_a_step():
    rng = np.random.default_rng(1)
    X = _rows(rng, 500, 3.0)
    X[7] = 0.0
    X[8, :32] = 0.0
    X[9, 5] = 3.0e30                       # a block with one huge element: everything else in it quantises to 0
    o = fo.quantize_e2m3(X)
    s = np.ldexp(1.0, o["scale_exp"]).repeat(32, axis=1)
    y = np.abs(o["deq"]) / s
    grid = np.concatenate([np.arange(0, 16) * 0.125, 2 + np.arange(0, 8) * 0.25, 4 + np.arange(0, 8) * 0.5])
    assert np.isin(y, grid).all() and y.max() <= 7.5
    assert (np.sign(o["deq"]) * np.sign(X) >= 0).all()
    step = np.where(y < 2, 0.125, np.where(y < 4, 0.25, 0.5)) * s
    assert (np.abs(o["deq"] - X) <= 0.5 * step * (1 + 1e-6)).all()
    # the scale is the smallest power of two that holds the block's maximum
    mx = np.abs(X).reshape(500, 2, 32).max(-1)
    sc = np.ldexp(1.0, o["scale_exp"])
    nz = mx > 0
    assert (mx[nz] <= 7.5 * sc[nz]).all() and (mx[nz] > 7.5 * sc[nz] / 2).all()
    assert (o["scale_exp"][~nz] == -127).all() and (o["deq"][7] == 0).all()


def test_screen_bound_holds_on_random_rows():
    rng = np.random.default_rng(2)
    for sq, sk in ((1.5, 1.0), (0.01, 300.0), (40.0, 0.002), (1e-12, 1e12), (3e-38, 1.0)):
        err, E = _check(_rows(rng, 300, sq), _rows(rng, 700, sk))
        assert err > 0 and E < 40 * err + 2e-6        # a bound, not a wild over-estimate (Cauchy-Schwarz over 64 random signs; + E's 1e-6 absolute term)
    K = _rows(rng, 400, 1.0)
    K[::7, 3] *= 1000.0                                 # blocks dominated by one element
    Q = _rows(rng, 100, 1.0)
    Q[::5] = 0.0
    _check(Q, K)


def test_screen_bound_is_reached_by_aligned_midpoint_rows():
    """Key elements at rounding midpoints that round toward zero (1.0625, 1.3125, 1.5625, 1.8125 in units of the block scale),
    with the query's sign pattern and a query that quantises exactly: every product's error has the same sign, and the
    Cauchy-Schwarz step is an equality — |s - s~| comes within a few percent of E_q and must not pass it."""
    rng = np.random.default_rng(3)
    sig = np.where(rng.random(64) < 0.5, -1.0, 1.0)
    mids = np.array([1.0625, 1.3125, 1.5625, 1.8125])
    K = np.stack([sig * np.full(64, mids[j % 4]) for j in range(64)]).astype(np.float32)
    K *= 0.25                                            # another binade
    Q = np.stack([sig * 1.5, -sig * 3.0]).astype(np.float32)
    assert np.array_equal(fo.bf16_round(K), K) and np.array_equal(fo.bf16_round(Q), Q)
    q, k = fo.quantize_e2m3(Q), fo.quantize_e2m3(K)
    assert np.array_equal(q["deq"], Q)                   # 6.0 and 6.0 in units of their scales
    S = Q.astype(np.float64) @ K.astype(np.float64).T
    St = q["deq"].astype(np.float64) @ k["deq"].astype(np.float64).T
    E = fo.screen_error(q["nrm"][:, 0], q["nrm"][:, 1], k["d2"].max(), k["t2"].max()).astype(np.float64)
    ratio = np.abs(S - St).max(axis=1) / E
    assert (ratio <= 1.0).all() and (ratio > 0.95).all(), ratio


def test_screen_T():
    assert fo.screen_T(1) == 21 and fo.screen_T(2) == 22 and fo.screen_T(20000) == 36 and fo.screen_T(32768) == 36 and fo.screen_T(32769) == 37
