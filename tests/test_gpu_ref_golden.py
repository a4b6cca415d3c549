"""GPU: the HIP path (through the C ABI) against vectors produced by EXECUTING THE REFERENCE'S OWN
CODE (tests/golden/ref_*.npz, generator tests/golden/make_golden_from_reference.py).  The reference
itself does not exist on the GPU box: only the committed data does."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def test_getcors_vs_reference_output(cuda0):
    """registration.getCors (exact-f32 MFMA kernel) vs the reference's getCors (inference.py:142-149) run
    on torch-CPU: the same winning indices, log-probabilities to f32 rounding; leaves=3 as the reference."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration
    g = np.load(G / "ref_getcors.npz")
    for c in range(int(g["n_cases"])):
        Q, K, leaves = torch.from_numpy(g[f"Q{c}"]).to(cuda0), torch.from_numpy(g[f"K{c}"]).to(cuda0), int(g[f"leaves{c}"])
        idx, vals = registration.getCors(Q, K, leaves)
        assert not idx.is_cuda and idx.dtype == torch.int64 and vals.is_cuda
        assert idx.shape == g[f"idx{c}"].shape and vals.shape == g[f"vals{c}"].shape
        assert np.array_equal(idx.numpy(), g[f"idx{c}"])
        np.testing.assert_allclose(vals.cpu().numpy(), g[f"vals{c}"], atol=2e-5)


def test_filter_vs_reference_output(cuda0):
    """isr_select_top vs the reference's statements inference.py:282-288: integer-exact kept set and
    the same threshold value, both branches (n > 500, n <= 500), tied values included."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, registration
    g = np.load(G / "ref_filter.npz")
    for c in range(int(g["n_cases"])):
        v = torch.from_numpy(g[f"in{c}"]).to(cuda0)
        keep, M, thr = ops.select_top(v[:, 0])
        m = int(M.item())
        assert np.array_equal(keep[:m].cpu().numpy(), g[f"nidx{c}"]), c
        assert np.float32(thr.item()) == g[f"thr{c}"], c
        assert np.array_equal(registration.filter_top(v), g[f"nidx{c}"])


def test_add_adds_vs_reference_output(cuda0):
    """isr_add_metric / isr_nn_batched vs the reference's ADD / ADDS (inference.py:116-120, sklearn
    KDTree(leaf_size=2)).  Clouds are uploaded as f32: 1e-4 mm covers it."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration
    g = np.load(G / "ref_add_adds.npz")
    registration.set_surface_points(g["surface"])
    for i in range(len(g["add"])):
        a = registration.ADD(g["verts"], g["Rg"][i], g["tg"][i], g["Rp"][i], g["tp"][i])
        s = registration.ADDS(g["verts"], g["Rg"][i], g["tg"][i], g["Rp"][i], g["tp"][i])
        assert abs(a - g["add"][i]) < 1e-4 and abs(s - g["adds"][i]) < 1e-4


def test_rel_pose_tables_vs_reference_output(cuda0):
    """isr_rel_pose_table (both modes) vs compute_rel_poses (choosePose.py:43-51) and
    calculate_relative_pose (verfication.py:9-19) executed from the reference."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration
    g = np.load(G / "ref_relpose.npz")
    tab = registration.relative_pose_table(g["R"], g["t"], "choose")
    np.testing.assert_allclose(tab[:, :, :3, :], g["choose"], rtol=0, atol=1e-12)
    tab = registration.relative_pose_table(g["R"], g["t"], "verif")
    np.testing.assert_allclose(tab[:, :, :3, :], g["verif"], rtol=0, atol=1e-9)    # np.linalg.inv vs R^T
    assert np.all(tab[:, :, 3, :] == [0, 0, 0, 1])


def test_estimate_pose_front_vs_reference_statements(cuda0):
    """isr_ep_prepare, isr_corr_logsoftmax, isr_ep_pool_corr vs poseEstSurf.py:37-107 executed from the
    reference (avg_queries=True)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, pose_est_surf as pes
    g = np.load(G / "ref_estimate_front.npz")
    ml, q, keys = (torch.from_numpy(g[f"avg_{k}"]).to(cuda0) for k in ("mask_lgts", "query_img", "obj_keys"))
    mlp, nmlp, mp, queries, res = pes.prepare(ml, q, int(g["down_sample_scale"]), True)
    np.testing.assert_allclose(mlp.cpu().numpy(), g["avg_mask_log_prob"], atol=1e-6)
    np.testing.assert_allclose(nmlp.cpu().numpy(), g["avg_neg_mask_log_prob"], atol=1e-6)
    np.testing.assert_allclose(mp.cpu().numpy(), g["avg_mask_prob"], atol=1e-6)
    raw = ops.corr_logsoftmax(queries, keys)
    np.testing.assert_allclose(pes.pool_corr(raw, res).cpu().numpy(), g["avg_corr_matrix_log"], atol=1e-5)
    np.testing.assert_allclose((raw.exp() * mp[:, None]).cpu().numpy(), g["avg_corr_matrix"], atol=1e-6)
    assert np.array_equal(pes._k_scaled(g["avg_K_in"], 3), g["avg_K"])


def test_estimate_pose_patch_branch_vs_reference_statements(cuda0):
    """isr_ep_patch_corr (+ isr_ep_pool_corr) vs poseEstSurf.py:72-107 executed from the reference with
    avg_queries=False: the sampling matrix (block centres) and the scoring matrix (block maxima, 3x3 pooled)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_est_surf as pes
    g = np.load(G / "ref_estimate_front.npz")
    ml, q, keys = (torch.from_numpy(g[f"patch_{k}"]).to(cuda0) for k in ("mask_lgts", "query_img", "obj_keys"))
    _, _, mp, _, res = pes.prepare(ml, q, int(g["down_sample_scale"]), True)
    centre, bmax, res2 = pes.patch_corr(q, keys, int(g["down_sample_scale"]))
    assert res2 == res
    np.testing.assert_allclose(pes.pool_corr(bmax, res).cpu().numpy(), g["patch_corr_matrix_log"], atol=1e-5)
    np.testing.assert_allclose((centre.exp() * mp[:, None]).cpu().numpy(), g["patch_corr_matrix"], atol=1e-6)


def test_refine_objective_vs_reference_statements(cuda0):
    """isr_refine_objective (value + analytic d/dt) vs pose_refine.py:60-68, 78-87 executed from the
    reference with torch autograd, including border-clamped samples."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_refine
    g = np.load(G / "ref_refine_objective.npz")
    q, den = torch.from_numpy(g["query_img"]).to(cuda0), torch.from_numpy(g["denom_img"]).to(cuda0)
    obj = pose_refine.RefineObjective(torch.from_numpy(g["X"]).to(cuda0), torch.from_numpy(g["keys"]).to(cuda0), q, den,
                                      g["K_crop"], g["R"])
    for t, s, gr in zip(g["t"], g["score"], g["grad_t"]):
        pose = np.concatenate([np.zeros(3), t])
        assert abs(obj(pose) - s) <= 2e-5 * max(1.0, abs(s))
        np.testing.assert_allclose(obj(pose, return_grad=True)[3:], gr, rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("mode", ["nearest", "bicubic"])
def test_refine_objective_modes_vs_reference_statements(cuda0, mode):
    """isr_refine_objective with ISR_INTERP_NEAREST / ISR_INTERP_BICUBIC vs pose_refine.py:60-68, 78-87 executed from the
    reference with interpolation = 'nearest' / 'bicubic' under autograd (nearest: zero gradient there and here)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_refine
    g = np.load(G / "ref_refine_modes.npz")
    q, den = torch.from_numpy(g["query_img"]).to(cuda0), torch.from_numpy(g["denom_img"]).to(cuda0)
    obj = pose_refine.RefineObjective(torch.from_numpy(g["X"]).to(cuda0), torch.from_numpy(g["keys"]).to(cuda0), q, den,
                                      g["K_crop"], g["R"], interpolation=mode)
    from oracle import refine_pose_oracle as rpo
    qh, dh = torch.from_numpy(g["query_img"]), torch.from_numpy(g["denom_img"])[..., None]
    kh, Xh = torch.from_numpy(g["keys"]), torch.from_numpy(g["X"])
    for t, s, gr in zip(g["t"], g[f"score_{mode}"], g[f"grad_t_{mode}"]):
        pose = np.concatenate([np.zeros(3), t])
        # nearest is discontinuous: a sample within f32 rounding of a half-integer may land on the neighbouring pixel
        # (the reference evaluates the coordinates in f32, the kernel in f64): one of 150 points moves the mean by < 2e-2
        tol = 2e-2 if mode == "nearest" else 5e-5 * max(1.0, abs(s))
        assert abs(obj(pose) - s) <= tol
        got = obj(pose, return_grad=True)[3:]
        # the reference's f32 autograd gradient (the fixture).  bicubic: a point that projects far outside the image
        # (camera-frame z near 0) has four coefficient derivatives that cancel only to f32 rounding, multiplied by
        # (c_x - u) / z ~ 1e4..1e6 — the fixture's d/dt_z carries that noise (2.4e-2 at t_z = 60, against the same
        # statements evaluated in f64 below); bilinear / nearest zero the gradient of clamped samples explicitly
        np.testing.assert_allclose(got, gr, rtol=5e-3, atol=5e-2 if mode == "bicubic" else 2e-6)
        # the same statements evaluated in f64 (torch autograd): the kernel's f64 value and gradient agree tightly
        v64, g64 = rpo.objective(t, g["R"], Xh, kh, qh, dh, g["K_crop"], return_grad=True, interpolation=mode, dtype=torch.float64)
        if mode != "nearest":
            assert abs(obj(pose) - v64) <= 1e-9 * max(1.0, abs(v64))
        np.testing.assert_allclose(got, g64, rtol=1e-7, atol=1e-9)
    with pytest.raises(ValueError):
        pose_refine.RefineObjective(torch.from_numpy(g["X"]).to(cuda0), torch.from_numpy(g["keys"]).to(cuda0), q, den,
                                    g["K_crop"], g["R"], interpolation="lanczos")


def test_vote_vs_reference_statements(cuda0):
    """a10 / a11 on the device against choosePose.py:98-107, :121-145 executed from the reference (ref_vote.npz): the two
    relative-pose tables, every entry of the n x n error matrix, the chosen image (ties -> the first, as np.argmax) and a top
    list with the reference's vote counts in the reference's order."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration as reg, sequence
    g = np.load(G / "ref_vote.npz")
    reg.set_surface_points(g["surface"])
    err, adds = reg.vote_error_rows(g["verts"], g["surface"], g["gt_rel"], g["pred_rel"], float(g["diameter"]))
    assert np.array_equal(err, g["error"])
    image_id, top, rows = sequence.vote_choose_image(g["verts"], g["surface"], g["R_gt"], g["t_gt"], g["R_pred"], g["t_pred"],
                                                     float(g["diameter"]))
    assert image_id == int(g["image_id"]) and np.array_equal(rows, g["error"])
    # the same with every item first bracketed from the surface cloud's distance field (the default from 1 024 items)
    st = {}
    image_b, _, rows_b = sequence.vote_choose_image(g["verts"], g["surface"], g["R_gt"], g["t_gt"], g["R_pred"], g["t_pred"],
                                                    float(g["diameter"]), bounds=True, stats=st)
    assert image_b == image_id and np.array_equal(rows_b, g["error"]) and st["by_bounds"] + st["exact"] == rows_b.size
    sums = g["error"].sum(1)
    assert np.array_equal(sums[np.asarray(top)], sums[g["top_indices"]])


def test_prune_vs_reference_statements(cuda0):
    """isr_ep_prune against poseEstSurf.py:119-121, :145, :147-177 executed from the reference (ref_estimate_prune.npz): the
    largest pairwise pixel distance (float32, as the reference's p2d), the depth-window and normal masks per solved sample,
    the ordered selection with and without pruning, the f32 poses handed to the scorer, and the returnPoints gathers."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_est_surf as pes
    g = np.load(G / "ref_estimate_prune.npz")
    res, m = int(g["res"]), int(g["m"])
    pm = g["poses_mask"]
    ci = torch.from_numpy(g["corr_idx"]).to(cuda0)
    poses = torch.from_numpy(g["poses"]).to(cuda0)
    ok = torch.from_numpy(pm.astype(np.uint8)).to(cuda0)
    pts = torch.from_numpy(g["obj_pts"]).to(cuda0)
    nrm = torch.from_numpy(g["obj_normals"]).to(cuda0)
    for tag, do_prune in (("prune", True), ("noprune", False)):
        max_eval = int(g[f"{tag}_max_eval"])
        dist, sm, nm, keep, kidx, nk, Rt32 = pes.prune(ci, poses, ok, pts, nrm, res, m, float(g["K"][0, 0]), float(g["diameter"]), 0.1,
                                                       do_prune, max_eval)
        torch.cuda.synchronize()
        assert np.array_equal(dist.cpu().numpy()[pm], g[f"{tag}_dist_2d"].astype(np.float32))
        assert np.array_equal(sm.cpu().numpy().astype(bool)[pm], g[f"{tag}_size_mask"])
        assert np.array_equal(nm.cpu().numpy().astype(bool)[pm], g[f"{tag}_normals_mask"])
        n = min(int(nk.item()), max_eval)
        assert n == int(g[f"{tag}_n_poses"])
        got = Rt32[:n].cpu().numpy()
        assert np.array_equal(got[:, :, :3], g[f"{tag}_R"].astype(np.float32)) and np.array_equal(got[:, :, 3], g[f"{tag}_t"].astype(np.float32))
        if do_prune:                       # returnPoints: the surviving samples' correspondences, all of them (:167-169)
            kept = kidx[:int(nk.item())].long()
            cik = ci[kept]
            p2 = torch.stack([(cik // m) % res, (cik // m) // res], dim=-1).float().cpu().numpy()
            assert np.array_equal(p2, g["prune_p2dCp"]) and np.array_equal(pts[cik % m].cpu().numpy(), g["prune_p3dCp"])


def test_masked_queries_getcors_filter_assembly_vs_reference_statements(cuda0):
    """The per-image front of inference.py on the device against its statements :252-263, :265-280, :282-290 executed from the
    reference (ref_assembly.npz): every third pixel of the network output under the mask in the reference's order
    (masked_queries / isr_prep_queries), getCors' indices (exact f32 path, 12 channels: the split route), the top-80 % cut on
    the reference's own values, and the 3-D / 2-D arrays handed to pnp — both branches of the 500-correspondence rule."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, registration as reg, sequence
    g = np.load(G / "ref_assembly.npz")
    for c in range(int(g["n_cases"])):
        feats = torch.from_numpy(g[f"feats{c}"]).to(cuda0)
        mask = g[f"mask{c}"]
        keys, pts = torch.from_numpy(g[f"keys{c}"]).to(cuda0), torch.from_numpy(g[f"pts{c}"]).to(cuda0)
        mf, ep2d = reg.masked_queries(feats, mask, down_sample=3, n_feat=12)
        assert torch.equal(mf.cpu(), torch.from_numpy(g[f"maskedfeats{c}"]))
        assert np.array_equal(ep2d, np.stack([g[f"Y1_{c}"], g[f"X1_{c}"]], 1).astype(np.float64))
        idx, vals = reg.getCors(mf, keys, leaves=1)
        assert np.array_equal(idx.numpy(), g[f"idx1_{c}"])
        np.testing.assert_allclose(vals.cpu().numpy(), g[f"in1_{c}"], atol=2e-5)
        nidx = reg.filter_top(torch.from_numpy(g[f"in1_{c}"]).to(cuda0))
        assert np.array_equal(nidx, g[f"nidx{c}"])
        keep, M, _ = ops.select_top(torch.from_numpy(g[f"in1_{c}"][:, 0]).to(cuda0))
        p3d, p2d = ops.gather_corr(torch.from_numpy(g[f"idx1_{c}"]).int().to(cuda0), keep, M, pts.float(),
                                   torch.from_numpy(ep2d).float().to(cuda0))
        m = int(M.item())
        assert np.array_equal(p3d[:m].cpu().numpy(), g[f"ep3d{c}"].astype(np.float32))
        assert np.array_equal(p2d[:m].cpu().numpy(), g[f"ep2d{c}"].astype(np.float32))
        # the camera of the sub-sampled crop (:260-263)
        from imagesequenceregistrationfor6dposeestimationlabeling_amd import formats
        K = g[f"camMat_in{c}"].copy()
        K[:2, 2] += 0.5; K[:2] /= 3; K[:2, 2] -= 0.5
        np.testing.assert_array_equal(K, g[f"camMat{c}"])


@pytest.mark.parametrize("dataset", ["tless", "ruapc"])
def test_acceptance_counts_vs_reference_statements(cuda0, dataset):
    """sequence.acceptance_counts (one batched call) against inference.py:299-320 executed from the reference image by image
    (ref_acceptance.npz): the two error values per image to 1e-4 mm (f32 upload), the same accept / reject decisions, workCT,
    rotWorkCT and the list of accepted image names, on both dataset branches (ADD-S for T-LESS, ADD otherwise)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import sequence
    g = np.load(G / "ref_acceptance.npz")
    n = len(g["R_gt"])
    poses = torch.from_numpy(np.concatenate([g["R_pred"], g["t_pred"][:, :, None]], 2).reshape(n, 12)).to(cuda0)
    status = torch.ones(n, dtype=torch.int32, device=cuda0)
    names = [f"{i:06d}.png" for i in range(n)]
    out = sequence.acceptance_counts(g["verts"], g["surface"], g["R_gt"], g["t_gt"], poses, status, float(g["diameter"]),
                                     dataset=dataset, names=names)
    np.testing.assert_allclose(out["final_error"], g[f"{dataset}_final_error"], atol=1e-4)
    np.testing.assert_allclose(out["final_errorR"], g[f"{dataset}_final_errorR"], atol=1e-4)
    assert out["workCT"] == int(g[f"{dataset}_workCT"]) and out["rotWorkCT"] == int(g[f"{dataset}_rotWorkCT"])
    assert out["correct_predicted_ids"] == [str(x) for x in g[f"{dataset}_correct"]]
    margin = np.abs(np.concatenate([g[f"{dataset}_final_error"], g[f"{dataset}_final_errorR"]]) - 0.1 * float(g["diameter"]))
    assert margin.min() > 1e-2                          # no decision of the fixture hangs on the f32 rounding


def test_refine_denominator_vs_reference_statement(cuda0):
    """pose_refine.denominator_image (K1's lse output: no (H*W x keys) matrix) against pose_refine.py:56 executed from the
    reference (ref_refine_denominator.npz)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_refine as pr
    g = np.load(G / "ref_refine_denominator.npz")
    d = pr.denominator_image(torch.from_numpy(g["query_img"]).to(cuda0), torch.from_numpy(g["keys_sampled"]).to(cuda0))
    assert d.shape == g["denom_img"].shape
    np.testing.assert_allclose(d.cpu().numpy(), g["denom_img"], atol=2e-5)


def test_zbuf_score_vs_reference_fragments(cuda0):
    """isr_zbuf_score against poseEstSurf.py:183-197, :201-212, :214-223 executed from the reference around its two
    torch_scatter calls (ref_batch_score.npz): projection and half-to-even rounding, the ignore bin, z > 0, mask / coordinate
    scores, -inf without hits (a pose behind the camera, a pose off the image), the normalisations.  (The matrix-free scorer
    returns isr_zbuf_score's bits: test_matrix_free_stages_equal_the_materialised_ones.)"""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_est_surf as pes
    g = np.load(G / "ref_batch_score.npz")
    res = int(g["res"])
    pts = torch.from_numpy(g["obj_pts"]).to(cuda0)
    R, t = torch.from_numpy(g["R"]).to(cuda0), torch.from_numpy(g["t"]).to(cuda0)
    mlp, nmlp = torch.from_numpy(g["mask_log_prob"]).to(cuda0), torch.from_numpy(g["neg_mask_log_prob"]).to(cuda0)
    corr = torch.from_numpy(g["corr_matrix_log"]).to(cuda0)
    got = pes.zbuf_score(pts, R, t, g["K"], res, mlp, nmlp, corr)
    for a, name, n_inf in zip(got, ("score", "mask_score", "coord_score"), (2, 0, 2)):
        a, b = a.cpu().numpy(), g[name]
        assert np.array_equal(np.isinf(a), np.isinf(b)) and int(np.isinf(b).sum()) == n_inf
        np.testing.assert_allclose(a[np.isfinite(b)], b[np.isfinite(b)], atol=2e-5)


def test_chamfer_pick_vs_reference_fragments(cuda0):
    """registration.chamfer_pairs / sequence.pick_by_chamfer against verfication.py:70-80, :83-85, :98, :100-102, :105-106
    executed from the reference around its two Open3D distance calls (ref_pick.npz): the rotation-only clouds as written there,
    every pair's Chamfer value to 1e-4 mm (f32 upload of the cloud), the picked pair."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration as reg, sequence
    g = np.load(G / "ref_pick.npz")
    n = len(g["R_gt"])
    ch = reg.chamfer_pairs(g["pc1"], g["R_pred"], g["R_relative"]).cpu().numpy()
    np.testing.assert_allclose(ch, g["chamferdis"], atol=1e-4)
    poses = torch.from_numpy(np.concatenate([g["R_pred"], g["t_pred"][:, :, None]], 2).reshape(n, 12)).to(cuda0)
    idx, val = sequence.pick_by_chamfer(torch.from_numpy(g["pc1"]).float().to(cuda0), poses, g["R_gt"], g["t_gt"], n)
    assert idx == int(g["min_index"]) and abs(val - float(g["min_chamfer"])) < 1e-4
    gap = np.sort(g["chamferdis"])
    assert gap[1] - gap[0] > 1e-2                          # the pick of the fixture does not hang on the f32 rounding
