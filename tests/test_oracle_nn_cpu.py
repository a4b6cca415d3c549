"""Pins the C oracle's nearest-neighbour arithmetic against the reference's own library calls:
sklearn KDTree(leaf_size=2).query (inference.py:118-120, ADD-S) and scipy cKDTree (stand-in for
open3d compute_point_cloud_distance, verfication.py:97-101)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation
from sklearn.neighbors import KDTree


def _pose(rng, tz=700.0):
    R = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix()
    t = np.array([0.0, 0.0, tz]) + rng.normal(0, 20, 3)
    return np.concatenate([R, t[:, None]], axis=1)


def test_nn_matches_sklearn_adds(oracle_lib):
    rng = np.random.default_rng(0)
    S = rng.normal(0, 40, (3000, 3)).astype(np.float32)   # surfacePointsScaled
    V = rng.normal(0, 40, (1500, 3)).astype(np.float32)   # CAD verts
    gt, pr = _pose(rng), _pose(rng)
    # reference ADDS: targets = S.R^T + T (pred), queries = V.gtR^T + gtT
    tree = KDTree(S.astype(np.float64).dot(pr[:, :3].T) + pr[:, 3], leaf_size=2)
    d_ref, i_ref = tree.query(V.astype(np.float64).dot(gt[:, :3].T) + gt[:, 3], k=1)
    o = oracle_lib.nn_batched(V, S, Tq=gt[None], Tt=pr[None])
    # f32 search can flip only between near-equidistant neighbours: distances agree to 1e-4 mm
    np.testing.assert_allclose(o["nn_d"][0], d_ref[:, 0], atol=2e-4)
    assert (o["nn_idx"][0] == i_ref[:, 0]).mean() > 0.999
    assert abs(o["sum_d"][0] / len(V) - d_ref.mean()) < 1e-5
    assert o["n_in"][0] == len(V)


def test_nn_identity_matches_ckdtree_and_radius(oracle_lib):
    rng = np.random.default_rng(1)
    A = rng.normal(0, 30, (2000, 3)).astype(np.float32)
    B = rng.normal(0, 30, (2500, 3)).astype(np.float32)
    d_ref, i_ref = cKDTree(B.astype(np.float64)).query(A.astype(np.float64), k=1)
    o = oracle_lib.nn_batched(A, B)
    assert np.array_equal(o["nn_idx"][0], i_ref)          # untransformed f32 clouds: exact
    np.testing.assert_allclose(o["nn_d"][0], d_ref, rtol=1e-12, atol=1e-12)
    r = 3.0
    o = oracle_lib.nn_batched(A, B, radius=r)
    inl = d_ref <= r
    assert o["n_in"][0] == inl.sum()
    assert np.array_equal(o["nn_idx"][0], np.where(inl, i_ref, -1))
    np.testing.assert_allclose(o["sum_d2"][0], (d_ref[inl] ** 2).sum(), rtol=1e-12)
    # Kabsch sums
    q, t = A[inl].astype(np.float64), B[i_ref[inl]].astype(np.float64)
    np.testing.assert_allclose(o["cov"][0, 0:3], q.sum(0), rtol=1e-12)
    np.testing.assert_allclose(o["cov"][0, 3:6], t.sum(0), rtol=1e-12)
    np.testing.assert_allclose(o["cov"][0, 6:15].reshape(3, 3), q.T @ t, rtol=1e-11)


def test_nn_tie_lowest_index(oracle_lib):
    tgt = np.array([[1, 0, 0], [-1, 0, 0], [1, 0, 0], [0, 5, 0]], np.float32)
    qry = np.array([[0, 0, 0], [1, 0, 0], [0, 4, 0]], np.float32)
    o = oracle_lib.nn_batched(qry, tgt)
    assert o["nn_idx"][0].tolist() == [0, 0, 3]
