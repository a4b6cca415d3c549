"""GPU: the HIP path (through registration.py / the C ABI) against the committed golden vectors."""
from pathlib import Path

import numpy as np
import pytest
import torch

from imagesequenceregistrationfor6dposeestimationlabeling_amd import synth

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def reg(cuda0):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration
    return registration


def test_getcors_and_filter(reg):
    g = np.load(G / "getcors_d12.npz")
    idx, vals = reg.getCors(torch.from_numpy(g["Q"]).cuda(), torch.from_numpy(g["K"]).cuda(), 1)
    assert (idx.numpy() != g["idx"]).sum() <= 1
    np.testing.assert_allclose(vals.cpu().numpy(), g["vals"], atol=3e-5)
    # the filter is integer-exact on identical values: feed it the golden values
    assert np.array_equal(reg.filter_top(torch.from_numpy(g["vals"]).cuda()), g["nidx"])
    s = np.load(G / "filter_small.npz")
    assert np.array_equal(reg.filter_top(torch.from_numpy(s["vals"]).cuda()), s["nidx"])


def test_add_adds(reg):
    g = np.load(G / "adds.npz")
    assert abs(reg.ADD(g["V"], g["gtR"], g["gtT"], g["R"], g["T"]) - float(g["add"])) < 1e-6
    assert abs(reg.ADDS(g["V"], g["gtR"], g["gtT"], g["R"], g["T"], surface_pts=g["S"]) - float(g["adds"])) < 1e-4


def test_chamfer_pairs(reg):
    g = np.load(G / "chamfer_pairs.npz")
    got = reg.chamfer_pairs(g["pc"], g["R_pred"], g["R_rel"]).cpu().numpy()
    np.testing.assert_allclose(got, g["chamfer"], atol=1e-4)
    assert reg.choose_best(got)[0] == int(np.argmin(g["chamfer"]))


def test_vote(reg):
    g = np.load(G / "vote.npz")
    err, adds = reg.vote_error_rows(g["V"], g["S"], g["gt_rel"], g["pred_rel"], float(g["diameter"]))
    np.testing.assert_allclose(adds, g["adds"], atol=1e-4)
    assert np.array_equal(err, g["error"])
    np.testing.assert_allclose(reg.relative_pose_table(g["R"], g["t"], "choose"), g["gt_rel"], atol=1e-12)


def test_pnp(reg):
    g = np.load(G / "pnp_ransac.npz")
    R, t, inl = reg.pnp(g["p3d"], g["p2d"], g["K"], itr=int(g["H"]), reperr=2, seed=int(g["seed"]), confidence=1.0)
    assert np.array_equal(inl, g["inliers"])              # bit-exact inlier set for the fixed seed
    assert synth.rot_angle(R, g["pose"][:, :3]) < 1e-4 and np.linalg.norm(t - g["pose"][:, 3]) < 1e-3
    # default confidence (0.99, as cv2's): the staged loop stops at 96 of 500 hypotheses
    g = np.load(G / "pnp_ransac_conf99.npz")
    R, t, inl = reg.pnp(g["p3d"], g["p2d"], g["K"], itr=int(g["H"]), reperr=2, seed=int(g["seed"]))
    assert np.array_equal(inl, g["inliers"])
    assert synth.rot_angle(R, g["pose"][:, :3]) < 1e-4 and np.linalg.norm(t - g["pose"][:, 3]) < 1e-3


def test_icp(reg):
    g = np.load(G / "icp.npz")
    f0, r0 = reg.evaluate_registration(g["source"], g["target"], 20, g["init"])
    assert abs(f0 - float(g["fitness0"])) < 1e-12 and abs(r0 - float(g["rmse0"])) < 1e-6
    T, fit, rmse = reg.icp_point_to_point(g["source"], g["target"], 20, g["init"])
    assert synth.rot_angle(T[:3, :3], g["T"][:3, :3]) < 1e-9          # north_star: 1e-4 rad / 1e-3 mm
    assert np.linalg.norm(T[:3, 3] - g["T"][:3, 3]) < 1e-6 and abs(fit - float(g["fitness"])) < 1e-12
    assert abs(reg.final_chamfer(g["source"], g["target"], T, g["cad"]) - float(g["final_chamfer"])) < 1e-3
