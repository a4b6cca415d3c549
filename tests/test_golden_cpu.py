"""CPU: the C / NumPy oracle against the committed golden vectors (tests/golden/make_golden.py).
getcors/filter/adds vectors were produced by the reference's literal expressions with its own
libraries (torch, sklearn); the C oracle must reproduce them."""
from pathlib import Path

import numpy as np
import torch

G = Path(__file__).resolve().parent / "golden"


def test_getcors_golden(oracle_lib):
    g = np.load(G / "getcors_d12.npz")
    o = oracle_lib.corr_argmax_f32(g["Q"], g["K"])
    # torch's CPU matmul accumulates in a different order than a k-ordered fma chain: indices can
    # differ only where the top-2 logits are within f32 noise
    bad = np.nonzero(o["idx"] != g["idx"])[0]
    assert len(bad) <= 1 and ((o["maxlogit"] - o["top2"])[bad] < 1e-5).all()
    np.testing.assert_allclose(o["maxlogit"] - o["lse"], g["vals"][:, 0], atol=2e-5)


def test_filter_golden():
    from oracle import registration_oracle as ro
    g = np.load(G / "getcors_d12.npz")
    assert np.array_equal(ro.filter_top(torch.from_numpy(g["vals"])), g["nidx"])
    g = np.load(G / "filter_small.npz")
    assert np.array_equal(ro.filter_top(torch.from_numpy(g["vals"])), g["nidx"])
    assert len(g["nidx"]) == len(g["vals"]) - 2      # n <= 500: threshold is the 2nd smallest


def test_adds_golden(oracle_lib):
    g = np.load(G / "adds.npz")
    Tq = np.concatenate([g["gtR"], g["gtT"][:, None]], 1)[None]
    Tt = np.concatenate([g["R"], g["T"][:, None]], 1)[None]
    o = oracle_lib.nn_batched(g["V"], g["S"], Tq, Tt)
    assert abs(o["sum_d"][0] / len(g["V"]) - float(g["adds"])) < 1e-5


def test_chamfer_pairs_golden(oracle_lib):
    g = np.load(G / "chamfer_pairs.npz")
    pc, Rp, Rr = g["pc"], g["R_pred"], g["R_rel"]
    n1 = len(Rr)
    Tg = np.zeros((n1, 3, 4)); Tp = np.zeros((n1, 3, 4))
    Tg[:, :, :3] = np.einsum("nji,njk->nik", Rr, Rp[:-1])
    Tp[:, :, :3] = np.transpose(Rp[1:], (0, 2, 1))
    ab = oracle_lib.nn_batched(pc, pc, Tp, Tg)
    ba = oracle_lib.nn_batched(pc, pc, Tg, Tp)
    np.testing.assert_allclose(0.5 * (ab["sum_d"] + ba["sum_d"]) / len(pc), g["chamfer"], atol=1e-5)


def test_vote_golden(oracle_lib):
    g = np.load(G / "vote.npz")
    n = g["gt_rel"].shape[0]
    o = oracle_lib.nn_batched(g["V"], g["S"], g["gt_rel"][..., :3, :].reshape(-1, 12),
                              g["pred_rel"][..., :3, :].reshape(-1, 12), want_idx=False, want_dist=False,
                              want_cov=False)
    adds = (o["sum_d"] / len(g["V"])).reshape(n, n)
    np.testing.assert_allclose(adds, g["adds"], atol=1e-4)
    assert np.array_equal((adds < 0.1 * float(g["diameter"])).astype(float), g["error"])


def test_pnp_golden():
    from oracle import pnp_oracle as po
    g = np.load(G / "pnp_ransac.npz")
    o = po.pnp_ransac(g["p3d"], g["p2d"], g["K"], H=int(g["H"]), reperr=2.0, seed=int(g["seed"]), confidence=1.0)
    assert np.array_equal(o["samples"], g["samples"]) and np.array_equal(o["n_inl"], g["n_inl"])
    assert o["best"] == int(g["best"]) and np.array_equal(o["inliers"], g["inliers"])
    np.testing.assert_allclose(o["Rt"], g["pose"], atol=1e-9)
    # adaptive termination (confidence 0.99): the loop stops at 96 of 500 hypotheses
    g = np.load(G / "pnp_ransac_conf99.npz")
    o = po.pnp_ransac(g["p3d"], g["p2d"], g["K"], H=int(g["H"]), reperr=2.0, seed=int(g["seed"]), confidence=float(g["confidence"]))
    assert o["n_eval"] == int(g["n_eval"]) == 96 and np.array_equal(o["n_inl"], g["n_inl"])
    assert o["best"] == int(g["best"]) != int(g["best_of_all"]) and np.array_equal(o["inliers"], g["inliers"])
    np.testing.assert_allclose(o["Rt"], g["pose"], atol=1e-9)


def test_icp_golden():
    from oracle import registration_oracle as ro
    g = np.load(G / "icp.npz")
    T, fit, rmse, traj = ro.icp_point_to_point(g["source"], g["target"], 20, g["init"])
    np.testing.assert_allclose(T, g["T"], atol=1e-9)
    assert abs(fit - float(g["fitness"])) < 1e-12 and abs(rmse - float(g["rmse"])) < 1e-9
    assert len(traj) - 1 == int(g["n_iter"])
