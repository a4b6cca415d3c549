"""GPU parity of the verification stage AT THE SIZES bench.py RUNS (round-2 verdict, weak point 2): the consecutive-
pair Chamfer pick over 63 pairs of 20 000 x 20 000 points (BASELINE configs[1]) and of 50 000 x 50 000 points
(configs[3]) — the block-cooperative grid search with its bench-size tile counts, run lists and box growth —, per-query
neighbours against the C oracle through every NN path, and the ICP loop (warm-started filter passes) + final Chamfer at
20 000 / 20 000 / 5 000 points against the Kabsch oracle loop.
Tolerances: Chamfer values 1e-4 mm absolute with the same first minimum (verfication.py:105-106); neighbours and f64
distances bit-exact; ICP 1e-9 rad / 1e-6 mm against the loop with exact f64 neighbours (north_star asks for 1e-4 rad /
1e-3 mm)."""
import numpy as np
import pytest
import torch

from imagesequenceregistrationfor6dposeestimationlabeling_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg(cuda0):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration
    return registration


@pytest.fixture(scope="module")
def ro():
    from oracle import registration_oracle
    return registration_oracle


def _sequence(rng, n):
    """GT poses and predictions as the bench's registration leaves them: most within a fraction of a degree of the
    truth (near copies: small search boxes), a few degrees off here and there, three failures with unrelated
    orientations (wide boxes, long run lists)."""
    Rg, tg = synth.random_poses(rng, n)
    deg = np.full(n, 0.02)
    deg[rng.choice(n, n // 8, replace=False)] = 2.0
    deg[[5, n // 2]] = 25.0
    Rp = np.array([synth.perturb_pose(rng, Rg[i], tg[i], deg[i], 0)[0] for i in range(n)])
    for i in (11, n - 2, n - 1):
        Rp[i] = synth.random_poses(rng, 1)[0][0]
    return Rg, tg, Rp


def _rel(ro, Rg, tg):
    return np.array([ro.calculate_relative_pose(Rg[i], tg[i], Rg[i + 1], tg[i + 1])[0] for i in range(len(Rg) - 1)])


@pytest.mark.parametrize("N,n", [(20000, 64), (50000, 64)])
def test_chamfer_pick_at_bench_size(reg, ro, N, n):
    """63 pairs of N x N points, both directions: every pair's Chamfer value against the f64 cKDTree oracle, and the
    same first minimum."""
    rng = np.random.default_rng(20240)
    pc = synth.tless_like(rng, N)                      # bench.make_model's cloud
    Rg, tg, Rp = _sequence(np.random.default_rng(99), n)
    Rrel = _rel(ro, Rg, tg)
    got = reg.chamfer_pairs(pc, Rp, Rrel).cpu().numpy()
    ref = ro.chamfer_pairs(pc.astype(np.float64), Rp, Rrel)
    assert got.shape == (n - 1,)
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)
    assert int(np.argmin(got)) == int(np.argmin(ref))
    assert reg.choose_best(got)[0] == int(np.argmin(ref))


@pytest.mark.parametrize("N", [20000, 50000])
def test_pick_neighbours_bit_exact_on_every_path(reg, oracle_lib, cuda0, N):
    """The transforms of the pick (rotation only, verfication.py:83-85) for 63 pairs; per-query winners and f64
    distances of five items (near copies, a few degrees, unrelated) equal to orc_nn_batched bit for bit, through
    brute force, the per-lane grid and the block-cooperative grid (what the bench takes)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(20240)
    pc = synth.tless_like(rng, N)
    n = 64
    Rg, tg, Rp = _sequence(np.random.default_rng(99), n)
    from oracle import registration_oracle as ro
    Rrel = _rel(ro, Rg, tg)
    Tg = np.zeros((n - 1, 3, 4))
    Tp = np.zeros((n - 1, 3, 4))
    Tg[:, :, :3] = np.einsum("nji,njk->nik", Rrel, Rp[:-1])       # registration.chamfer_pairs
    Tp[:, :, :3] = np.transpose(Rp[1:], (0, 2, 1))
    sel = [0, 4, 10, 31, 62]                                        # 4 and 10 touch the 25-degree / failed images
    pcd = torch.from_numpy(pc).to(cuda0)
    tg_d, tp_d = torch.from_numpy(Tg.reshape(-1, 12)).to(cuda0), torch.from_numpy(Tp.reshape(-1, 12)).to(cuda0)
    o = oracle_lib.nn_batched(pc, pc, Tp[sel], Tg[sel], -1.0, want_cov=False)
    for path in (2, 1, 0):
        with ops.tuning(nn_path=path):
            if path == 0:       # brute force on the selected items only (63 x 50 000^2 pairs are not worth the minutes)
                r = ops.nn_batched(pcd, pcd, tp_d[sel], tg_d[sel], want_idx=True, want_dist=True)
                idx, d, sd = r.nn_idx.cpu().numpy(), r.nn_d.cpu().numpy(), r.sum_d.cpu().numpy()
            else:
                r = ops.nn_batched(pcd, pcd, tp_d, tg_d, want_idx=True, want_dist=True)
                idx, d, sd = r.nn_idx.cpu().numpy()[sel], r.nn_d.cpu().numpy()[sel], r.sum_d.cpu().numpy()[sel]
        assert np.array_equal(idx, o["nn_idx"]), path
        assert np.array_equal(d, o["nn_d"]), path
        np.testing.assert_allclose(sd, o["sum_d"], rtol=1e-12)


def test_icp_and_final_chamfer_at_bench_size(reg, ro):
    """bench.make_model's halves (20 000 points each of an 80 000-point cloud), CAD cloud of 5 000 points, init =
    inverse of a pose 2e-4 rad / 0.05 mm from the truth (what the bench's registration delivers) and of a pose
    2 degrees / 2 mm off: T, fitness, rmse and the final Chamfer against the oracle."""
    rng = np.random.default_rng(20240)
    N = 20000
    synth.tless_like(rng, N)
    cloud = synth.tless_like(rng, 4 * N)
    upper, lower = synth.split_halves(rng, cloud, N)
    cad = synth.tless_like(rng, 5000)
    Rg, tg = synth.random_poses(np.random.default_rng(99), 2)
    for k, (rot_deg, trans) in enumerate([(0.012, 0.05), (2.0, 2.0)]):
        Rp, tp = synth.perturb_pose(np.random.default_rng(7 + k), Rg[k], tg[k], rot_deg, trans)
        src = (upper.astype(np.float64) @ Rg[k].T + tg[k]).astype(np.float32)           # icp.py:68
        init = np.linalg.inv(np.vstack([np.hstack([Rp, tp[:, None]]), [0, 0, 0, 1]]))   # icp.py:88-92
        f, r = reg.evaluate_registration(src, lower, 20, init)
        rf, rr, _, _ = ro.evaluate_registration(src, lower, 20, init)
        assert abs(f - rf) < 1e-12 and abs(r - rr) < 1e-9
        T, fit, rmse = reg.icp_point_to_point(src, lower, 20, init)
        # The oracle loop with EXACT f64 neighbours (cKDTree; what Open3D's KD-tree on doubles returns, as far as is
        # known).  The device searches on f32-rounded coordinates, flags every near tie and decides those in f64
        # (nn_search_kernel<.., EXACT>): each pass returns the exact neighbours, and the 30-pass trajectories agree to
        # rounding.  (Round 2's loop kept the f32 winner: on these dense clouds one or two points per pass have two
        # targets equidistant to f32 rounding, and the trajectories separated by up to 6e-6 rad / 4e-3 mm —
        # profiles/r03_icp_bench_size_vs_oracle.txt.)
        Tr, rfit, rrmse, traj = ro.icp_point_to_point(src, lower, 20, init, search="f64")
        assert len(traj) > 3                                   # several warm-started passes ran
        assert synth.rot_angle(T[:3, :3], Tr[:3, :3]) < 1e-9
        assert np.linalg.norm(T[:3, 3] - Tr[:3, 3]) < 1e-6
        assert abs(fit - rfit) < 1e-12 and abs(rmse - rrmse) < 1e-9
        c = reg.final_chamfer(src, lower, T, cad)
        assert abs(c - ro.final_chamfer(src, lower, Tr, cad)) < 1e-6
        # the f32-winner definition (oracle search="f32") is the one that drifts away now
        T32, _, _, _ = ro.icp_point_to_point(src, lower, 20, init, search="f32")
        assert synth.rot_angle(T[:3, :3], T32[:3, :3]) < 1e-4
