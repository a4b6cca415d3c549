"""GPU parity: isr_nn_batched (through the C ABI) vs the C oracle — per-query winners and f64
distances bit for bit, sums to 1e-12 relative (tree vs sequential order)."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu


def _poses(rng, B, tz=700.0):
    R = Rotation.random(B, random_state=int(rng.integers(1 << 30))).as_matrix()
    t = np.array([0.0, 0.0, tz]) + rng.normal(0, 20, (B, 3))
    return np.concatenate([R, t[:, :, None]], axis=2)


def _run(cuda0, q, t, Tq, Tt, radius):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    tq = None if Tq is None else torch.from_numpy(Tq).to(cuda0)
    tt = None if Tt is None else torch.from_numpy(Tt).to(cuda0)
    r = ops.nn_batched(torch.from_numpy(q).to(cuda0), torch.from_numpy(t).to(cuda0), tq, tt,
                       radius=radius, want_idx=True, want_dist=True, want_cov=True)
    torch.cuda.synchronize()
    return {k: getattr(r, k).cpu().numpy() for k in ("sum_d", "sum_d2", "n_in", "nn_idx", "nn_d", "cov")}


def _compare(g, o):
    assert np.array_equal(g["nn_idx"], o["nn_idx"])
    assert np.array_equal(g["nn_d"], o["nn_d"])           # bit-exact f64 distances
    assert np.array_equal(g["n_in"], o["n_in"])
    np.testing.assert_allclose(g["sum_d"], o["sum_d"], rtol=1e-12)
    np.testing.assert_allclose(g["sum_d2"], o["sum_d2"], rtol=1e-12)
    np.testing.assert_allclose(g["cov"][:, :15], o["cov"][:, :15], rtol=1e-10, atol=1e-6)


@pytest.mark.parametrize("Nq,Nt,B,radius", [
    (1, 1, 1, -1.0),            # degenerate
    (63, 257, 1, -1.0),         # ragged, RQ=1, split targets
    (1500, 3000, 3, -1.0),      # ADD-S-like, batched transforms
    (5000, 5000, 1, 20.0),      # ICP step at config-1 size, RQ=4 + target split
    (4097, 1023, 5, 6.0),       # RQ=4 ragged with radius
])
def test_nn_batched_parity(cuda0, oracle_lib, Nq, Nt, B, radius):
    rng = np.random.default_rng(Nq * 7 + Nt)
    q = rng.normal(0, 40, (Nq, 3)).astype(np.float32)
    t = rng.normal(0, 40, (Nt, 3)).astype(np.float32)
    Tq = _poses(rng, B) if B > 1 or Nq == 5000 else None
    Tt = _poses(rng, B) if B > 1 else None
    if Tq is not None and Tt is None:   # ICP-like: source near target
        Tq = np.concatenate([np.eye(3), [[0.5], [-0.3], [0.2]]], axis=1)[None]
    g = _run(cuda0, q, t, Tq, Tt, radius)
    o = oracle_lib.nn_batched(q, t, Tq, Tt, radius)
    _compare(g, o)


def test_nn_ties_and_duplicates(cuda0, oracle_lib):
    rng = np.random.default_rng(5)
    base = rng.integers(-3, 4, (400, 3)).astype(np.float32)   # lattice: many exact ties
    t = np.concatenate([base, base])                          # duplicated targets
    q = rng.integers(-3, 4, (300, 3)).astype(np.float32)
    g = _run(cuda0, q, t, None, None, -1.0)
    o = oracle_lib.nn_batched(q, t)
    _compare(g, o)
    assert (g["nn_idx"] < 400).all()                          # lowest index wins


def test_nn_large_batch_vote_shape(cuda0, oracle_lib):
    """Many batch items sharing the clouds (the n x n vote shape); oracle on a sample."""
    rng = np.random.default_rng(9)
    q = rng.normal(0, 40, (1200, 3)).astype(np.float32)
    t = rng.normal(0, 40, (2000, 3)).astype(np.float32)
    B = 300
    Tq, Tt = _poses(rng, B), _poses(rng, B)
    g = _run(cuda0, q, t, Tq, Tt, -1.0)
    sel = [0, 1, 150, 299]
    o = oracle_lib.nn_batched(q, t, Tq[sel], Tt[sel], -1.0)
    assert np.array_equal(g["nn_idx"][sel], o["nn_idx"])
    assert np.array_equal(g["nn_d"][sel], o["nn_d"])
    np.testing.assert_allclose(g["sum_d"][sel], o["sum_d"], rtol=1e-12)


def _both_paths(cuda0, q, t, Tq, Tt, radius):
    """The same call through both grid searches and through brute force (the ISR_TUNE_NN_PATH knob
    forces the path): everything must agree bit for bit."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    out = {}
    for flag in ("2", "1", "0"):       # block-cooperative grid, per-query grid, brute force
        with ops.tuning(nn_path=int(flag)):
            out[flag] = _run(cuda0, q, t, Tq, Tt, radius)
    b = out["0"]
    for flag in ("2", "1"):
        for k in ("nn_idx", "nn_d", "n_in", "sum_d", "sum_d2", "cov"):
            assert np.array_equal(out[flag][k], b[k]), (flag, k)
    return out["2"]


def _surface(rng, n):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import synth
    return np.ascontiguousarray(synth.tless_like(rng, n), np.float32)


@pytest.mark.parametrize("radius", [-1.0, 4.0])
def test_nn_grid_chamfer_pairs(cuda0, oracle_lib, radius):
    """The pick's shape: one surface cloud under B pairs of nearby poses (verfication.py:61-102)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import synth
    rng = np.random.default_rng(31)
    pc = _surface(rng, 6000)
    B = 8
    Ta = _poses(rng, B)
    Tb = Ta.copy()
    for i in range(B):
        R, t = synth.perturb_pose(rng, Ta[i, :, :3], Ta[i, :, 3], 3.0, 2.0)
        Tb[i, :, :3], Tb[i, :, 3] = R, t
    g = _both_paths(cuda0, pc, pc, Ta, Tb, radius)
    _compare(g, oracle_lib.nn_batched(pc, pc, Ta, Tb, radius))


def test_nn_grid_far_nonrigid_and_degenerate(cuda0, oracle_lib):
    """Queries far outside the grid (brute-force pass), a target transform that is not rigid (item
    goes to the brute-force pass), lattice ties with duplicated targets, and targets on a line."""
    rng = np.random.default_rng(32)
    pc = _surface(rng, 3000)
    B = 4
    Ta, Tb = _poses(rng, B), _poses(rng, B)
    Ta[1, :, 3] += np.array([900.0, -400.0, 300.0])            # item 1: every query is far away
    Tb[2, :, :3] *= 1.3                                         # item 2: scaled target
    g = _both_paths(cuda0, pc[:2000], pc, Ta, Tb, -1.0)
    _compare(g, oracle_lib.nn_batched(pc[:2000], pc, Ta, Tb, -1.0))

    base = rng.integers(-6, 7, (1500, 3)).astype(np.float32)
    t = np.concatenate([base, base])
    q = rng.integers(-8, 9, (4200, 3)).astype(np.float32)
    g = _both_paths(cuda0, q, t, None, None, -1.0)
    _compare(g, oracle_lib.nn_batched(q, t))
    assert (g["nn_idx"] < 1500).all()

    line = np.zeros((2500, 3), np.float32)
    line[:, 0] = np.linspace(-50, 50, 2500)
    q = rng.normal(0, 30, (4100, 3)).astype(np.float32)
    g = _both_paths(cuda0, q, line, None, None, 10.0)
    _compare(g, oracle_lib.nn_batched(q, line, None, None, 10.0))
    same = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (2100, 1))
    g = _both_paths(cuda0, q, same, None, None, -1.0)
    assert (g["nn_idx"] == 0).all()


def test_nn_rejects_cpu_tensors(hip_lib):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    from imagesequenceregistrationfor6dposeestimationlabeling_amd._capi import IsrError
    with pytest.raises(IsrError):
        ops.nn_batched(torch.zeros(4, 3), torch.zeros(4, 3))


def test_icp_loop_ignores_the_nn_path_knob(cuda0, monkeypatch):
    """isr_icp_point_to_point on partially overlapping halves (many source points have no target within
    a few grid cells, only within the 20 mm radius).  The loop always takes the brute-force search with exact
    near-tie resolution (round 3; the grid variants of the loop were 7x / 40x slower here and are gone): the
    ISR_TUNE_NN_PATH knob, which isr_nn_batched honours, must not change T, fitness or rmse; warm-started and cold
    passes return the same neighbours."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration, synth
    rng = np.random.default_rng(31)
    cloud = synth.tless_like(rng, 24000)
    upper, lower = synth.split_halves(rng, cloud, 6000)
    R, t = synth.random_poses(rng, 1)
    Rp, tp = synth.perturb_pose(rng, R[0], t[0], 2.0, 2.0)
    src = (upper.astype(np.float64) @ R[0].T + t[0]).astype(np.float32)
    init = np.linalg.inv(np.vstack([np.hstack([Rp, tp[:, None]]), [0, 0, 0, 1]]))
    out = {}
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    for mode in ("0", "1", "2"):
        with ops.tuning(nn_path=int(mode)):
            out[mode] = registration.icp_point_to_point(src, lower, 20, init)
    T0, f0, r0 = out["0"]
    assert 0.5 < f0 <= 1.0 and r0 > 1.0          # far from a trivial all-matched case
    for mode in ("1", "2"):
        T, f, r = out[mode]
        assert np.array_equal(T, T0) and f == f0 and r == r0, mode
    assert np.array_equal(registration.icp_point_to_point(src, lower, 20, init)[0], T0)     # default plan
    with ops.tuning(icp_warm=0):                                                              # every pass cold
        Tc, fc, rc = registration.icp_point_to_point(src, lower, 20, init)
    assert np.array_equal(Tc, T0) and fc == f0 and rc == r0
