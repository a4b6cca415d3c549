#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz by EXECUTING THE REFERENCE'S OWN CODE for the hot-path pieces
that need nothing but torch / numpy / scikit-learn.

Runs in the build container only (it reads /root/reference, which does not travel to the GPU box).
The reference's hot-path scripts cannot be imported — every one parses argv, needs cuda:0 and
imports cv2 / open3d / trimesh at import — so nothing is imported: each file is read as text,
parsed with `ast`, and only the named function definitions / the statements of the cited line
ranges are compiled and executed, in a namespace that provides the libraries the reference itself
uses (torch, numpy, torch.nn.functional as F, sklearn.neighbors.KDTree) and the inputs.  What is
written to disk is DATA ONLY — inputs and the outputs the reference code produced for them; no
reference text is stored.

  ref_getcors.npz        getCors               inference.py:142-149 (= finalposes.py:38-45 = choosePose.py:35-42)
  ref_filter.npz         top-80 % cut          inference.py:282-288 (statements)
  ref_add_adds.npz       ADD, ADDS             inference.py:116-120
  ref_relpose.npz        compute_rel_poses     choosePose.py:43-51;  calculate_relative_pose  verfication.py:9-19
  ref_cammat.npz         crop / down-sample    inference.py:203-206, 212-222, 260-263 (statements)
  ref_normalize.npz      normalize             inference.py:135-141
  ref_estimate_front.npz estimate_pose front   poseEstSurf.py:37-107 (statements; both avg_queries branches)
  ref_refine_objective.npz  refine_pose `sample` + objective body   pose_refine.py:60-68, 78-87 (statements)
  ref_refine_modes.npz      the same with interpolation = 'nearest' / 'bicubic'   (`... make_golden_from_reference.py refine_modes`)

Not reproducible this way (cv2 / open3d / torch_scatter are absent and must not be stood in for):
pnp (inference.py:123-134), estimate_pose's P3P loop and batch_score (poseEstSurf.py:133-237),
Chamfer / ICP (verfication.py:97-101, icp.py:96-117) — those stay "parity unpinned" (DESIGN.md §2).
  ref_acceptance.npz     (`acceptance` argument) inference.py:299-320: ADD / ADD-S acceptance counters of the per-image loop
  ref_assembly.npz       (`assembly` argument) inference.py:252-263, :265-280, :282-290: masked lattice pixels -> getCors -> top-80 % -> ep3d / ep2d
  ref_refine_denominator.npz (`denominator` argument) pose_refine.py:56: the log-sum-exp image over the sampled keys
  ref_batch_score.npz    (`batch_score` argument) poseEstSurf.py:183-197, :201-212, :214-223 (batch_score around its two torch_scatter calls)
  ref_pick.npz           (`pick` argument) verfication.py:70-80, :83-85, :98, :100-102, :105-106 (the Chamfer pick around its two Open3D distance calls)
  ref_estimate_prune.npz (`prune` argument) poseEstSurf.py:119-121, :145, :147-177: gathers, pruning masks, ordered selection
  ref_vote.npz           (`vote` argument) the n x n relative-pose table choosePose.py:98-107 and the ADD-S vote :121-145

Run from the repo root:  python tests/golden/make_golden_from_reference.py
"""
import ast
import contextlib
import io
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from sklearn.neighbors import KDTree

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _tree(fname: str) -> ast.Module:
    return ast.parse((REF / fname).read_text(), filename=fname)


def ref_function(fname: str, name: str, ns: dict):
    """Compile ONE top-level `def name` of a reference script into namespace `ns` and return it."""
    for node in _tree(fname).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, f"{fname}:{name}", "exec"), ns)
            return ns[name]
    raise KeyError(f"{fname}: def {name} not found")


def _stmts_in(body, lo, hi, out):
    for st in body:
        if st.lineno >= lo and st.end_lineno <= hi:
            out.append(st)
            continue
        if st.end_lineno < lo or st.lineno > hi:
            continue
        for field in ("body", "orelse", "finalbody"):
            sub = getattr(st, field, None)
            if isinstance(sub, list) and sub and isinstance(sub[0], ast.stmt):
                _stmts_in(sub, lo, hi, out)


def ref_statements(fname: str, lo: int, hi: int, must_contain: tuple[str, ...]):
    """The outermost statements of a reference script that lie entirely inside lines [lo, hi], as a
    compiled code object.  `must_contain` guards against a shifted line range."""
    out: list[ast.stmt] = []
    _stmts_in(_tree(fname).body, lo, hi, out)
    text = "\n".join(ast.unparse(s) for s in out)
    for needle in must_contain:
        assert needle in text, f"{fname}:{lo}-{hi}: expected `{needle}` in the extracted statements"
    return compile(ast.Module(body=out, type_ignores=[]), f"{fname}:{lo}-{hi}", "exec")


def unit_rows(rng, n, d, scale):
    k = rng.normal(size=(n, d))
    return (scale * k / np.linalg.norm(k, axis=1, keepdims=True)).astype(np.float32)


def random_rotation(rng):
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def main():
    torch.manual_seed(1)                       # inference.py:37
    rng = np.random.default_rng(20261004)
    base = {"torch": torch, "np": np, "F": F, "KDTree": KDTree}

    # ---------------------------------------------------------------- getCors (three identical copies)
    fns = [ref_function(f, "getCors", dict(base)) for f in ("inference.py", "finalposes.py", "choosePose.py")]
    out = {}
    for c, (P, N, D, tau, leaves) in enumerate([(400, 2000, 12, 5.0, 1), (257, 999, 12, 3.0, 1), (64, 500, 12, 5.0, 3)]):
        K = unit_rows(rng, N, D, tau)
        gt = rng.integers(N, size=P)
        Q = (K[gt] + 0.35 * rng.normal(size=(P, D))).astype(np.float32)
        res = [fn(torch.from_numpy(Q), torch.from_numpy(K), leaves) for fn in fns]
        for r in res[1:]:
            assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1])
        idx, vals = res[0]
        assert not idx.is_cuda
        out.update({f"Q{c}": Q, f"K{c}": K, f"leaves{c}": leaves, f"idx{c}": idx.numpy(), f"vals{c}": vals.numpy()})
    np.savez_compressed(OUT / "ref_getcors.npz", n_cases=3, **out)

    # ---------------------------------------------------------------- top-80 % cut (statements)
    code = ref_statements("inference.py", 282, 288, ("torch.sort", "threshval", "nidx"))
    out, c = {}, 0
    for n in (1500, 501, 500, 37, 2):
        for ties in (False, True):
            v = -np.abs(rng.normal(size=(n, 1))).astype(np.float32) * 1e-2
            if ties:
                v[rng.integers(n, size=n // 3), 0] = v[rng.integers(n), 0]    # a third of the values collide
            ns = dict(base, in1=torch.from_numpy(v))
            exec(code, ns)
            out.update({f"in{c}": v, f"nidx{c}": np.asarray(ns["nidx"]), f"thr{c}": np.float32(ns["threshval"].item())})
            c += 1
    np.savez_compressed(OUT / "ref_filter.npz", n_cases=c, **out)

    # ---------------------------------------------------------------- ADD / ADDS
    ns = dict(base)
    ADD = ref_function("inference.py", "ADD", ns)
    ADDS = ref_function("inference.py", "ADDS", ns)
    ns2 = dict(base)
    ADD2, ADDS2 = ref_function("choosePose.py", "ADD", ns2), ref_function("choosePose.py", "ADDS", ns2)
    verts = rng.normal(size=(300, 3)) * [40, 25, 15]
    surf = (rng.normal(size=(800, 3)) * [40, 25, 15]).astype(np.float32).astype(np.float64)
    ns["surfacePointsScaled"] = surf            # the module global ADDS reads (inference.py:88, choosePose.py:160)
    ns2["surfacePointsScaled"] = surf
    Rg, tg, Rp, tp, add, adds = [], [], [], [], [], []
    for i in range(5):
        R1, R2 = random_rotation(rng), random_rotation(rng)
        if i < 3:                                # near pose: a perturbation of the GT
            w = rng.normal(size=3) * 0.03
            Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            R2 = (np.eye(3) + Wx) @ R1
            R2, _ = np.linalg.qr(R2)
            R2 *= np.sign(np.diag(R2.T @ R1))[None, :]
        t1 = np.array([0, 0, 700.0]) + rng.normal(size=3) * 20
        t2 = t1 + rng.normal(size=3) * (2.0 if i < 3 else 30.0)
        Rg.append(R1); tg.append(t1); Rp.append(R2); tp.append(t2)
        add.append(ADD(verts, R1, t1, R2, t2))
        adds.append(ADDS(verts, R1, t1, R2, t2))
        assert add[-1] == ADD2(verts, R1, t1, R2, t2) and adds[-1] == ADDS2(verts, R1, t1, R2, t2)
    np.savez_compressed(OUT / "ref_add_adds.npz", verts=verts, surface=surf, Rg=np.array(Rg), tg=np.array(tg),
                        Rp=np.array(Rp), tp=np.array(tp), add=np.array(add), adds=np.array(adds))

    # ---------------------------------------------------------------- relative poses
    crp = ref_function("choosePose.py", "compute_rel_poses", dict(base))
    calc = ref_function("verfication.py", "calculate_relative_pose", dict(base))
    n = 6
    R = np.array([random_rotation(rng) for _ in range(n)])
    t = rng.normal(size=(n, 3)) * 50 + [0, 0, 700]
    rel_c = np.zeros((n, n, 3, 4))
    rel_v = np.zeros((n, n, 3, 4))
    for i in range(n):
        for j in range(n):
            a, b = crp(R[i], t[i], R[j], t[j])
            rel_c[i, j, :, :3], rel_c[i, j, :, 3] = a, b
            a, b = calc(R[i], t[i], R[j], t[j])
            rel_v[i, j, :, :3], rel_v[i, j, :, 3] = a, b
    np.savez_compressed(OUT / "ref_relpose.npz", R=R, t=t, choose=rel_c, verif=rel_v)

    # ---------------------------------------------------------------- camera matrix of the crop (statements)
    even = ref_statements("inference.py", 203, 206, ("w % 2", "h % 2"))
    crop = ref_statements("inference.py", 212, 222, ("centerX", "camMat", "camparams"))
    down = ref_statements("inference.py", 260, 263, ("camMat[:2, 2] += 0.5", "down_sample"))
    boxes, Ks, cams, Ms = [], [], [], []
    for (x, y, w, h) in [(100, 80, 200, 150), (311, 7, 97, 133), (0, 0, 640, 480), (250, 200, 51, 50), (17, 300, 224, 1)]:
        camparams = np.array([[1075.65 + rng.normal(), 0, 320 + rng.normal()], [0, 1073.9 + rng.normal(), 240 + rng.normal()], [0, 0, 1]])
        ns = dict(base, x=x, y=y, w=w, h=h, camparams=camparams, camMatScaling=True, down_sample=3)
        exec(even, ns)
        exec(crop, ns)
        Ms.append(np.array(ns["M"]))
        exec(down, ns)
        boxes.append((x, y, w, h)); Ks.append(camparams); cams.append(np.array(ns["camMat"]))
    np.savez_compressed(OUT / "ref_cammat.npz", boxes=np.array(boxes), K=np.array(Ks), camMat=np.array(cams), M=np.array(Ms))

    # ---------------------------------------------------------------- normalize
    normalize = ref_function("inference.py", "normalize", dict(base))
    img = rng.integers(0, 256, size=(9, 11, 3), dtype=np.uint8)
    np.savez_compressed(OUT / "ref_normalize.npz", img=img, out=normalize(img))

    # ---------------------------------------------------------------- estimate_pose, front part (statements)
    front = ref_statements("poseEstSurf.py", 37, 107, ("F.logsigmoid", "corr_matrix_log", "F.avg_pool2d", "mask_prob"))
    out = {}
    for tag, avg in (("avg", True), ("patch", False)):
        r, e, m, scale = 30, 12, 160, 3
        mask_lgts = torch.from_numpy(rng.normal(size=(r, r)).astype(np.float32) * 3)
        query_img = torch.from_numpy(rng.normal(size=(r, r, e)).astype(np.float32) * 0.7)
        obj_keys = torch.from_numpy(unit_rows(rng, m, e, 3.0))
        Kc = np.array([[600.0, 0, r / 2 - 0.3], [0, 590.0, r / 2 + 0.2], [0, 0, 1]])
        ns = dict(base, mask_lgts=mask_lgts.clone(), query_img=query_img.clone(), obj_keys=obj_keys, K=Kc.copy(),
                  down_sample_scale=scale, max_pool=True, avg_queries=avg)
        exec(front, ns)
        out.update({f"{tag}_mask_lgts": mask_lgts.numpy(), f"{tag}_query_img": query_img.numpy(),
                    f"{tag}_obj_keys": obj_keys.numpy(), f"{tag}_K_in": Kc, f"{tag}_K": ns["K"],
                    f"{tag}_mask_log_prob": ns["mask_log_prob"].numpy(), f"{tag}_neg_mask_log_prob": ns["neg_mask_log_prob"].numpy(),
                    f"{tag}_mask_prob": ns["mask_prob"].numpy(), f"{tag}_img_pts": ns["img_pts"].numpy(),
                    f"{tag}_corr_matrix": ns["corr_matrix"].numpy(), f"{tag}_corr_matrix_log": ns["corr_matrix_log"].numpy()})
    np.savez_compressed(OUT / "ref_estimate_front.npz", down_sample_scale=3, **out)

    # ---------------------------------------------------------------- refine_pose: sample + objective body
    sample_def = ref_statements("pose_refine.py", 60, 68, ("F.grid_sample", "padding_mode"))
    body = ref_statements("pose_refine.py", 78, 87, ("p_img_norm", "log_nominator", "score"))
    res, e, Npt = 32, 12, 150
    query_img = torch.from_numpy(rng.normal(size=(res, res, e)).astype(np.float32))
    denom_img = torch.from_numpy(rng.normal(size=(res, res, 1)).astype(np.float32) + 5)
    keys_masked = torch.from_numpy(rng.normal(size=(Npt, e)).astype(np.float32))
    X = rng.normal(size=(Npt, 3)).astype(np.float32) * 20
    coord_masked = torch.from_numpy(np.concatenate([X, np.ones((Npt, 1), np.float32)], 1))
    K_crop = torch.tensor([[45.0, 0, 15.5], [0, 45.0, 15.5], [0, 0, 1]])
    Rm = torch.from_numpy(random_rotation(rng).astype(np.float32))
    scores, grads, ts = [], [], []
    for tz in (120.0, 60.0, 35.0):               # the last one pushes projections across the border (clamped samples)
        tvec = torch.tensor([1.5, -2.0, tz], requires_grad=True)
        ns = dict(base, interpolation="bilinear", query_img=query_img, denom_img=denom_img, keys_masked=keys_masked,
                  coord_masked=coord_masked, K_crop=K_crop, res_crop=res, Rt=torch.cat((Rm, tvec[:, None]), dim=1))
        exec(sample_def, ns)
        exec(body, ns)
        ns["score"].backward()
        scores.append(ns["score"].item()); grads.append(tvec.grad.numpy().copy()); ts.append(tvec.detach().numpy().copy())
    np.savez_compressed(OUT / "ref_refine_objective.npz", query_img=query_img.numpy(), denom_img=denom_img.numpy()[..., 0],
                        keys=keys_masked.numpy(), X=X, K_crop=K_crop.numpy().astype(np.float64), R=Rm.numpy().astype(np.float64),
                        t=np.array(ts, np.float64), score=np.array(scores), grad_t=np.array(grads))
    print("wrote", sorted(p.name for p in OUT.glob("ref_*.npz")))


def vote():
    """choosePose.py:98-107 (the n x n table of relative poses) and :121-145 (the ADD-S vote, its argmax and the ranking
    written to top_50_choices.txt) executed from the reference's own statements on a small sequence: predicted poses are
    the ground truth with errors of graded size, so that some entries agree and some do not, and two rows tie."""
    rng = np.random.default_rng(20261005)
    base = {"torch": torch, "np": np, "F": F, "KDTree": KDTree}
    ns = dict(base)
    ref_function("choosePose.py", "compute_rel_poses", ns)
    ref_function("choosePose.py", "ADDS", ns)
    table = ref_statements("choosePose.py", 98, 107, ("compute_rel_poses", "transformation_matrix", "relative_poses"))
    loop = ref_statements("choosePose.py", 121, 138, ("final_error", "0.1 * diameter", "agreed_poses"))
    pick = ref_statements("choosePose.py", 144, 145, ("np.argmax", "np.argsort"))
    n = 7
    verts = rng.normal(size=(250, 3)) * [40, 25, 15]
    surf = (rng.normal(size=(600, 3)) * [40, 25, 15]).astype(np.float32).astype(np.float64)
    diameter = float(np.linalg.norm(verts.max(0) - verts.min(0)))
    Rg = np.array([random_rotation(rng) for _ in range(n)])
    tg = rng.normal(size=(n, 3)) * 40 + [0, 0, 700]
    Rp, tp = Rg.copy(), tg.copy()
    for i, (deg, mm) in enumerate([(0.2, 0.5), (0.3, 1.0), (14.0, 22.0), (0.2, 120.0), (70.0, 75.0), (0.4, 0.8), (0.1, 0.3)]):
        w = rng.normal(size=3); w *= np.deg2rad(deg) / np.linalg.norm(w)
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        th = np.linalg.norm(w)
        Rp[i] = (np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * Wx @ Wx) @ Rg[i]
        d = rng.normal(size=3); tp[i] = tg[i] + mm * d / np.linalg.norm(d)
    out = {}
    for tag, (RL, TL) in (("gt", (Rg, tg)), ("pred", (Rp, tp))):
        ns.update(RList=list(RL), TList=list(TL))
        with contextlib.redirect_stdout(io.StringIO()):
            exec(table, ns)
        out[tag] = np.array(ns["relative_poses"])
    ns.update(pred_rel_poses=out["pred"], gt_rel_poses=out["gt"], modelVerts=verts, surfacePointsScaled=surf,
              diameter=diameter, args=types.SimpleNamespace(dataset="tless"))
    with contextlib.redirect_stdout(io.StringIO()):
        exec(loop, ns)
        exec(pick, ns)
    err = np.array(ns["error"])
    assert 0 < err.sum() < err.size and len(set(err.sum(1).tolist())) < n, "the case must discriminate and hold a tie"
    np.savez_compressed(OUT / "ref_vote.npz", verts=verts, surface=surf, diameter=diameter, R_gt=Rg, t_gt=tg, R_pred=Rp, t_pred=tp,
                        gt_rel=out["gt"], pred_rel=out["pred"], error=err, agreed=np.array(ns["agreed_poses"]),
                        image_id=int(ns["image_id"]), top_indices=np.array(ns["top_indices"]))
    print("wrote ref_vote.npz: agreed", int(err.sum()), "of", err.size, "image_id", int(ns["image_id"]), "row sums", err.sum(1).tolist())


def assembly():
    """inference.py:252-263 (every third pixel of features and mask, the camera of the sub-sampled crop), :265-280 (masked
    pixels, their features, getCors, the 3-D / 2-D correspondence arrays) and :282-290 (the top-80 % cut and the filtered
    arrays handed to pnp) executed from the reference's own statements on a synthetic network output."""
    torch.manual_seed(3)
    rng = np.random.default_rng(20261007)
    base = {"torch": torch, "np": np, "F": F}
    ns0 = dict(base)
    ref_function("inference.py", "getCors", ns0)
    sub = ref_statements("inference.py", 252, 263, ("down_sample", "imfeats[:, ::down_sample, ::down_sample]", "camMat[:2, 2] += 0.5"))
    corr = ref_statements("inference.py", 265, 280, ("torch.where(inputMask)", "getCors(", "ep2d[:, 0] = maskIds[1]", "ep2d[:, 0:2]"))
    cut = ref_statements("inference.py", 282, 290, ("threshval", "nidx", "ep3d[nidx"))
    out = {}
    for c, (H, N, fill) in enumerate([(120, 3000, 0.55), (60, 800, 0.12)]):            # > 500 and <= 500 masked lattice pixels
        keys = unit_rows(rng, N, 12, 5.0)
        pts = (rng.normal(size=(N, 3)) * [40, 25, 15]).astype(np.float32).astype(np.float64)
        nrm = pts / np.linalg.norm(pts, axis=1, keepdims=True)
        feats = (0.4 * rng.normal(size=(1, H, H, 12))).astype(np.float32)
        mask = np.zeros((H, H, 3), np.uint8)
        yy, xx = np.mgrid[:H, :H]
        blob = ((yy - H / 2) ** 2 / (0.42 * H) ** 2 + (xx - H / 2.2) ** 2 / (0.3 * H) ** 2 < 1) & (rng.random((H, H)) < fill + 0.4)
        mask[blob] = 255
        own = rng.integers(0, N, size=(H, H))
        feats[0][blob] = keys[own[blob]] + 0.25 * rng.normal(size=(int(blob.sum()), 12)).astype(np.float32)
        camMat = np.array([[1100.0, 0, H / 2 - 0.7], [0, 1090.0, H / 2 + 0.4], [0, 0, 1]])
        ns = dict(ns0, cropMask=mask, imfeats=torch.from_numpy(feats), inputMask=torch.from_numpy(mask[:, :, 0]),
                  camMatScaling=True, camMat=camMat.copy(), sfeats=torch.from_numpy(keys), surfacePointsScaled=pts, n2Scaled=nrm)
        for code in (sub, corr, cut):
            exec(code, ns)
        n_masked = len(ns["in1"])
        assert (n_masked > 500) == (c == 0), n_masked
        out.update({f"feats{c}": feats, f"mask{c}": mask, f"keys{c}": keys, f"pts{c}": pts, f"camMat_in{c}": camMat,
                    f"camMat{c}": ns["camMat"], f"maskedfeats{c}": ns["maskedfeats"].numpy(), f"idx1_{c}": ns["idx1"].numpy(),
                    f"in1_{c}": ns["in1"].numpy(), f"X1_{c}": ns["X1"], f"Y1_{c}": ns["Y1"], f"nidx{c}": ns["nidx"],
                    f"ep3d{c}": ns["ep3d"], f"ep2d{c}": ns["ep2d"], f"n3d{c}": np.asarray(ns["n3d"])})
        print("case", c, "masked", n_masked, "kept", len(ns["nidx"]))
    np.savez_compressed(OUT / "ref_assembly.npz", n_cases=2, **out)
    print("wrote ref_assembly.npz")


def acceptance():
    """inference.py:299-320 — the per-image acceptance bookkeeping after pnp (ADD-S on T-LESS, ADD otherwise, of the pose and of
    its rotation alone against 0.1 * diameter; workCT, rotWorkCT, correct_predicted_ids) — executed from the reference's own
    statements in its own per-image loop shape, for both dataset branches, on poses with graded errors."""
    rng = np.random.default_rng(20261008)
    base = {"torch": torch, "np": np, "F": F, "KDTree": KDTree}
    body = ref_statements("inference.py", 299, 320, ("final_errorR", "0.1 * diameter", "rotWorkCT", "correct_predicted_ids"))
    n = 10

    def bar_with_arm(k):                      # an elongated, asymmetric object: rotations move its ends a long way
        a = rng.uniform([-100, -8, -5], [100, 8, 5], size=(k, 3))
        b = rng.uniform([60, 8, -5], [80, 70, 5], size=(k // 3, 3))
        return np.concatenate([a, b])
    verts = bar_with_arm(240)
    surf = bar_with_arm(700).astype(np.float32).astype(np.float64)
    diameter = float(np.linalg.norm(verts.max(0) - verts.min(0)))
    Rg = np.array([random_rotation(rng) for _ in range(n)])
    tg = rng.normal(size=(n, 3)) * 30 + [0, 0, 700]
    Rp, tp = Rg.copy(), tg.copy()
    errs = [(0.3, 0.5), (2.0, 90.0), (50.0, 1.0), (1.0, 2.0), (9.0, 14.0), (120.0, 1.0), (0.2, 0.4), (25.0, 60.0), (3.0, 150.0), (0.5, 1.0)]
    for i, (deg, mm) in enumerate(errs):
        w = rng.normal(size=3); w *= np.deg2rad(deg) / np.linalg.norm(w)
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        th = np.linalg.norm(w)
        Rp[i] = (np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * Wx @ Wx) @ Rg[i]
        d = rng.normal(size=3); tp[i] = tg[i] + mm * d / np.linalg.norm(d)
    out = {}
    for ds in ("tless", "ruapc"):
        ns = dict(base, modelVerts=verts, surfacePointsScaled=surf, diameter=diameter, workCT=0, rotWorkCT=0, correct_predicted_ids=[],
                  args=types.SimpleNamespace(dataset=ds))
        ref_function("inference.py", "ADD", ns)
        ref_function("inference.py", "ADDS", ns)
        fe, fr = [], []
        with contextlib.redirect_stdout(io.StringIO()):
            for i in range(n):
                ns.update(gtR=Rg[i], gtT=tg[i], R2=Rp[i], T2=tp[i], path=f"bop/x/train/000001/depth/{i:06d}.png", imID=i)
                exec(body, ns)
                fe.append(ns["final_error"]); fr.append(ns["final_errorR"])
        out.update({f"{ds}_final_error": np.array(fe), f"{ds}_final_errorR": np.array(fr), f"{ds}_workCT": ns["workCT"],
                    f"{ds}_rotWorkCT": ns["rotWorkCT"], f"{ds}_correct": np.array(ns["correct_predicted_ids"])})
        assert 0 < ns["workCT"] < ns["rotWorkCT"] < n, (ns["workCT"], ns["rotWorkCT"])
        print(ds, "workCT", ns["workCT"], "rotWorkCT", ns["rotWorkCT"])
    np.savez_compressed(OUT / "ref_acceptance.npz", verts=verts, surface=surf, diameter=diameter, R_gt=Rg, t_gt=tg, R_pred=Rp, t_pred=tp, **out)
    print("wrote ref_acceptance.npz")


def denominator():
    """pose_refine.py:56 — the log-sum-exp denominator image over the sampled keys — executed from the reference's statement."""
    rng = np.random.default_rng(20261009)
    base = {"torch": torch, "np": np, "F": F}
    stmt = ref_statements("pose_refine.py", 56, 56, ("torch.logsumexp", "keys_sampled.T"))
    res, e, nk = 40, 12, 700
    query_img = torch.from_numpy((rng.normal(size=(res, res, e)) * 0.8).astype(np.float32))
    keys_sampled = torch.from_numpy(unit_rows(rng, nk, e, 4.0))
    ns = dict(base, query_img=query_img, keys_sampled=keys_sampled)
    exec(stmt, ns)
    np.savez_compressed(OUT / "ref_refine_denominator.npz", query_img=query_img.numpy(), keys_sampled=keys_sampled.numpy(),
                        denom_img=ns["denom_img"].numpy())
    print("wrote ref_refine_denominator.npz", tuple(ns["denom_img"].shape))


def batch_score_fragments():
    """poseEstSurf.py:183-197 (projection of every vertex under every pose, rounding, the ignore bin), :201-212 (populated
    pixels, the mask score) and :214-223 (the no-hit case, normalisation, the sum) executed from the reference's own statements.
    The two torch_scatter calls in between (:200 scatter_min, :213 scatter_mean) cannot run here — the library is absent — and
    are NOT stood in for by the reference: the per-pixel minimum depth / its vertex and the per-pose mean are computed by plain
    torch (amin with the lowest vertex on equal depth; sum / count) and labelled as such.  What the fixture pins is everything
    around them: the projection and rounding convention, the ignore bin, z > 0, the mask / coordinate score assembly, -inf for a
    pose without hits, the normalisations."""
    rng = np.random.default_rng(20261010)
    base = {"torch": torch, "np": np, "F": F}
    pre = ref_statements("poseEstSurf.py", 183, 197, ("obj_pts_cam", "round_()", "mask_neg", "u.long()"))
    mid = ref_statements("poseEstSurf.py", 201, 212, ("z > 0", "mask_score_2d", "corr_matrix_log[u, z_arg]"))
    post = ref_statements("poseEstSurf.py", 214, 223, ("coord_score_mask", "np.log(2)", "np.log(m)", "score = mask_score + coord_score"))
    res, m, B = 16, 500, 9
    n = res * res
    obj_pts = torch.from_numpy((rng.normal(size=(m, 3)) * [40, 25, 15]).astype(np.float32))
    K = torch.tensor([[76.0, 0, res / 2 - 0.4], [0, 74.0, res / 2 + 0.2], [0, 0, 1]])
    R = torch.from_numpy(np.array([random_rotation(rng) for _ in range(B)]).astype(np.float32))
    t = torch.from_numpy((rng.normal(size=(B, 3)) * [6, 6, 40] + [0, 0, 500]).astype(np.float32))
    t[1] = torch.tensor([0.0, 0.0, -400.0])                 # behind the camera: populated pixels with z < 0 -> no hit
    t[2] = torch.tensor([900.0, 0.0, 500.0])                # out of the image: everything in the ignore bin
    mask_log_prob = torch.from_numpy(-np.abs(rng.normal(size=n)).astype(np.float32))
    neg_mask_log_prob = torch.from_numpy(-np.abs(rng.normal(size=n) * 2).astype(np.float32))
    corr_matrix_log = torch.from_numpy((-np.abs(rng.normal(size=(n, m))) * 4).astype(np.float32))
    ns = dict(base, R=R, t=t, obj_pts=obj_pts, K=K, res_sampled=res, n=n, m=m, device=torch.device("cpu"),
              mask_log_prob=mask_log_prob, neg_mask_log_prob=neg_mask_log_prob, corr_matrix_log=corr_matrix_log)
    exec(pre, ns)
    u_all = ns["u"].clone()
    # ---- not the reference: torch_scatter.scatter_min(z, u, dim_size=n + 1) by plain torch
    z_in, u_in = ns["z"], ns["u"]
    zmin = torch.full((B, n + 1), float("inf")).scatter_reduce(1, u_in, z_in, "amin", include_self=True)
    vid = torch.arange(m).expand(B, m)
    zarg = torch.full((B, n + 1), m, dtype=torch.long).scatter_reduce(
        1, u_in, torch.where(z_in == zmin.gather(1, u_in), vid, torch.full_like(vid, m)), "amin", include_self=True)
    zmin = torch.where(torch.isfinite(zmin), zmin, torch.zeros_like(zmin))          # torch_scatter leaves empty bins at 0
    zarg = torch.where(zarg == m, torch.zeros_like(zarg), zarg)
    ns.update(z=zmin, z_arg=zarg)
    exec(mid, ns)
    # ---- not the reference: torch_scatter.scatter_mean(coord_score, mask_pose_idx, dim_size=n_poses) by plain torch
    cs, mpi = ns["coord_score"], ns["mask_pose_idx"]
    ssum = torch.zeros(B).index_add_(0, mpi, cs)
    scnt = torch.zeros(B).index_add_(0, mpi, torch.ones_like(cs))
    ns["coord_score"] = torch.where(scnt > 0, ssum / scnt.clamp(min=1), torch.zeros(B))
    exec(post, ns)
    assert torch.isinf(ns["coord_score"][1]) and torch.isinf(ns["coord_score"][2]) and torch.isfinite(ns["coord_score"][0])
    np.savez_compressed(OUT / "ref_batch_score.npz", res=res, obj_pts=obj_pts.numpy(), K=K.numpy().astype(np.float64), R=R.numpy(),
                        t=t.numpy(), mask_log_prob=mask_log_prob.numpy(), neg_mask_log_prob=neg_mask_log_prob.numpy(),
                        corr_matrix_log=corr_matrix_log.numpy(), u=u_all.numpy(), score=ns["score"].numpy(),
                        mask_score=ns["mask_score"].numpy(), coord_score=ns["coord_score"].numpy())
    print("wrote ref_batch_score.npz: scores", ns["score"].numpy().round(3).tolist())


def pick():
    """verfication.py:70-80 (the two images' poses out of scene_gt.json / pred6d.json dictionaries, calculate_relative_pose),
    :83-85 (the rotation-only transforms of the model cloud, translations commented out), :98, :100-102 (means, the Chamfer
    value, the list) per pair and :105-106 (min, list.index) executed from the reference's own statements.  The two
    open3d compute_point_cloud_distance calls in between (:97, :99) cannot run here (Open3D absent) and are NOT the reference's:
    exact nearest-neighbour distances by scipy.spatial.cKDTree, labelled as such."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(20261011)
    base = {"torch": torch, "np": np}
    ns = dict(base)
    ref_function("verfication.py", "calculate_relative_pose", ns)
    poses_stmt = ref_statements("verfication.py", 70, 80, ("cam_R_m2c", "calculate_relative_pose", "pred6d[key4]"))
    cloud_stmt = ref_statements("verfication.py", 83, 85, ("pc1.dot(R1pred.T)", "pc1p1.dot(R_relative)", "pc1.dot(R2pred)"))
    m1 = ref_statements("verfication.py", 98, 98, ("np.mean(dists_pcdpred_to_pcdgt)",))
    m2 = ref_statements("verfication.py", 100, 102, ("np.mean(dists_pcdgt_to_pcdpred)", "chamfer_distance", "chamferdis.append"))
    fin = ref_statements("verfication.py", 105, 106, ("min(chamferdis)", "chamferdis.index"))
    n = 7
    pc1 = (rng.normal(size=(700, 3)) * [45, 20, 10] + [10, 0, 0]).astype(np.float32).astype(np.float64)
    Rg = np.array([random_rotation(rng) for _ in range(n)])
    tg = rng.normal(size=(n, 3)) * 30 + [0, 0, 700]
    Rp = Rg.copy()
    for i, deg in enumerate([3.0, 0.4, 0.5, 9.0, 0.4, 0.5, 20.0]):
        w = rng.normal(size=3); w *= np.deg2rad(deg) / np.linalg.norm(w)
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        th = np.linalg.norm(w)
        Rp[i] = (np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * Wx @ Wx) @ Rg[i]
    tp = tg + rng.normal(size=(n, 3))
    keys = [str(i) for i in range(n)]
    data = {k: [{"cam_R_m2c": Rg[i].reshape(-1).tolist(), "cam_t_m2c": tg[i].tolist()}] for i, k in enumerate(keys)}
    pred6d = {k: [{"R": Rp[i].reshape(-1).tolist(), "T": tp[i].tolist()}] for i, k in enumerate(keys)}
    ns.update(data=data, pred6d=pred6d, keysgt=keys, keyspred=keys, pc1=pc1, chamferdis=[])
    rel, clouds = [], None
    for i in range(0, n - 1):
        ns.update(i=i, key1=keys[i], key2=keys[i + 1], key3=keys[i], key4=keys[i + 1])        # :62-65
        exec(poses_stmt, ns)
        exec(cloud_stmt, ns)
        rel.append(np.array(ns["R_relative"]))
        if i == 0:
            clouds = (np.array(ns["pcgt"]), np.array(ns["pcpred"]))
        # ---- not the reference: open3d compute_point_cloud_distance = exact nearest-neighbour distances
        ns["dists_pcdpred_to_pcdgt"] = cKDTree(ns["pcgt"]).query(ns["pcpred"], k=1)[0]
        exec(m1, ns)
        ns["dists_pcdgt_to_pcdpred"] = cKDTree(ns["pcpred"]).query(ns["pcgt"], k=1)[0]
        exec(m2, ns)
    exec(fin, ns)
    ch = np.array(ns["chamferdis"])
    assert ch.max() > 1.5 * ch.min() and ns["min_index"] == int(np.argmin(ch))
    np.savez_compressed(OUT / "ref_pick.npz", pc1=pc1, R_gt=Rg, t_gt=tg, R_pred=Rp, t_pred=tp, R_relative=np.array(rel),
                        pcgt0=clouds[0], pcpred0=clouds[1], chamferdis=ch, min_index=int(ns["min_index"]), min_chamfer=float(ns["min_chamfer"]))
    print("wrote ref_pick.npz: chamferdis", ch.round(4).tolist(), "min_index", ns["min_index"])


def prune():
    """poseEstSurf.py:119-121 (sample indices -> pixel / surface-point / normal gathers), :145 (the solved samples) and
    :147-177 (the three pruning masks, the pruned pose list and its truncation) executed from the reference's own statements
    on given samples and poses (the poses themselves come out of cv2.solveP3P in the reference: here they are inputs)."""
    rng = np.random.default_rng(20261006)
    base = {"torch": torch, "np": np, "F": F}
    gather = ref_statements("poseEstSurf.py", 119, 121, ("p2d_idx", "img_pts[p2d_idx]", "obj_normals["))
    tonp = ref_statements("poseEstSurf.py", 124, 124, ("p2d.cpu().numpy()",))
    solved = ref_statements("poseEstSurf.py", 145, 145, ("poses_mask",))
    masks = ref_statements("poseEstSurf.py", 147, 169, ("dist_2d_min * res_sampled", "z_min", "normals_dot", "do_prune"))
    tail = ref_statements("poseEstSurf.py", 174, 177, ("max_pose_evaluations", "n_poses"))
    res, m, S = 24, 500, 400
    obj_pts = (rng.normal(size=(m, 3)) * [40, 25, 15]).astype(np.float32)
    obj_normals = obj_pts.astype(np.float64) / np.linalg.norm(obj_pts, axis=1, keepdims=True)       # f64, as normals_scaled.npy
    K = np.array([[130.0, 0, res / 2 - 0.4], [0, 128.0, res / 2 + 0.1], [0, 0, 1]])
    diameter = float(np.linalg.norm(obj_pts.max(0) - obj_pts.min(0)))
    ys, xs = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    img_pts = torch.from_numpy(np.stack([xs.ravel(), ys.ravel()], -1))                             # (n, 2) xy, :56-59
    R0 = random_rotation(rng)
    t0 = np.array([3.0, -2.0, 650.0])
    poses = np.zeros((S, 3, 4))
    for i in range(S):
        w = rng.normal(size=3) * 0.15
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        Rq, _ = np.linalg.qr((np.eye(3) + Wx) @ R0)
        Rq *= np.sign(np.diag(Rq.T @ R0))[None, :]
        poses[i, :, :3] = Rq
        poses[i, :, 3] = t0 + rng.normal(size=3) * [8, 8, 60] + (0 if i % 9 else [0, 0, rng.choice([-2000.0, 4000.0])])
    poses_mask = rng.random(S) > 0.1
    # samples: surface points that face the camera under R0 and their projections (+ a share of arbitrary picks)
    cam = obj_pts.astype(np.float64) @ R0.T + t0
    uv = cam @ K.T
    uv = np.rint(uv[:, :2] / uv[:, 2:]).astype(np.int64)
    vis = np.nonzero(((obj_normals @ R0.T) * cam).sum(1) < 0)[0]
    ks = rng.choice(vis, (S, 4))
    ks[::5] = rng.choice(m, (len(ks[::5]), 4))
    pix = np.clip(uv[ks, 1], 0, res - 1) * res + np.clip(uv[ks, 0], 0, res - 1)
    pix[::7, 1:3] = pix[::7, :1]                                                                   # three samples from one pixel area
    corr_idx = torch.from_numpy(pix * m + ks)
    out = {}
    for tag, do_prune, max_eval in (("prune", True, 60), ("noprune", False, 25)):
        ns = dict(base, corr_idx=corr_idx.clone(), m=m, img_pts=img_pts, obj_pts=torch.from_numpy(obj_pts), obj_normals=obj_normals,
                  poses=poses.copy(), poses_mask=poses_mask.copy(), dist_2d_min=0.1, res_sampled=res, K=K.copy(), obj_diameter=diameter,
                  do_prune=do_prune, max_pose_evaluations=max_eval)
        for code in (gather, tonp, solved, masks, tail):
            exec(code, ns)
        out.update({f"{tag}_dist_2d": ns["dist_2d"], f"{tag}_size_mask": ns["size_mask"], f"{tag}_normals_mask": ns["normals_mask"],
                    f"{tag}_dist_2d_mask": ns["dist_2d_mask"], f"{tag}_R": ns["R"], f"{tag}_t": ns["t"], f"{tag}_n_poses": ns["n_poses"],
                    f"{tag}_max_eval": max_eval})
        if do_prune:
            out["prune_p3dCp"], out["prune_p2dCp"] = ns["p3dCp"], ns["p2dCp"]
            kept = ns["dist_2d_mask"] & ns["size_mask"] & ns["normals_mask"]
            assert 5 < kept.sum() < 0.8 * len(kept) and (~ns["size_mask"]).any() and (~ns["normals_mask"]).any() and (~ns["dist_2d_mask"]).any()
    np.savez_compressed(OUT / "ref_estimate_prune.npz", res=res, m=m, obj_pts=obj_pts, obj_normals=obj_normals, K=K, diameter=diameter,
                        corr_idx=corr_idx.numpy(), poses=poses, poses_mask=poses_mask, **out)
    print("wrote ref_estimate_prune.npz: solved", int(poses_mask.sum()), "kept", int(out["prune_n_poses"]), "of max", 60)


def refine_modes():
    """refine_pose's objective with interpolation = 'nearest' / 'bicubic' (pose_refine.py:60-68 forwards `mode=` to
    F.grid_sample): the reference's own `sample` + objective statements executed under autograd.  Own RNG stream, so the
    files main() writes do not move.  -> ref_refine_modes.npz"""
    rng = np.random.default_rng(20261005)
    base = {"torch": torch, "np": np, "F": F}
    sample_def = ref_statements("pose_refine.py", 60, 68, ("F.grid_sample", "padding_mode"))
    body = ref_statements("pose_refine.py", 78, 87, ("p_img_norm", "log_nominator", "score"))
    res, e, Npt = 32, 12, 150
    query_img = torch.from_numpy(rng.normal(size=(res, res, e)).astype(np.float32))
    denom_img = torch.from_numpy(rng.normal(size=(res, res, 1)).astype(np.float32) + 5)
    keys_masked = torch.from_numpy(rng.normal(size=(Npt, e)).astype(np.float32))
    X = rng.normal(size=(Npt, 3)).astype(np.float32) * 20
    coord_masked = torch.from_numpy(np.concatenate([X, np.ones((Npt, 1), np.float32)], 1))
    K_crop = torch.tensor([[45.0, 0, 15.5], [0, 45.0, 15.5], [0, 0, 1]])
    Rm = torch.from_numpy(random_rotation(rng).astype(np.float32))
    out = dict(query_img=query_img.numpy(), denom_img=denom_img.numpy()[..., 0], keys=keys_masked.numpy(), X=X,
               K_crop=K_crop.numpy().astype(np.float64), R=Rm.numpy().astype(np.float64))
    ts = [(1.5, -2.0, 120.0), (1.5, -2.0, 60.0), (1.5, -2.0, 35.0)]     # the last pushes projections across the border
    out["t"] = np.array(ts, np.float64)
    for mode in ("nearest", "bicubic"):
        scores, grads = [], []
        for tv in ts:
            tvec = torch.tensor(tv, requires_grad=True)
            ns = dict(base, interpolation=mode, query_img=query_img, denom_img=denom_img, keys_masked=keys_masked,
                      coord_masked=coord_masked, K_crop=K_crop, res_crop=res, Rt=torch.cat((Rm, tvec[:, None]), dim=1))
            exec(sample_def, ns)
            exec(body, ns)
            ns["score"].backward()
            scores.append(ns["score"].item())
            grads.append(tvec.grad.numpy().copy())
        out[f"score_{mode}"] = np.array(scores)
        out[f"grad_t_{mode}"] = np.array(grads)
    np.savez_compressed(OUT / "ref_refine_modes.npz", **out)
    print("wrote ref_refine_modes.npz", {k: out[k] for k in out if k.startswith(("score", "grad"))})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "vote":
        sys.exit(vote())
    if len(sys.argv) > 1 and sys.argv[1] == "prune":
        sys.exit(prune())
    if len(sys.argv) > 1 and sys.argv[1] == "assembly":
        sys.exit(assembly())
    if len(sys.argv) > 1 and sys.argv[1] == "acceptance":
        sys.exit(acceptance())
    if len(sys.argv) > 1 and sys.argv[1] == "denominator":
        sys.exit(denominator())
    if len(sys.argv) > 1 and sys.argv[1] == "batch_score":
        sys.exit(batch_score_fragments())
    if len(sys.argv) > 1 and sys.argv[1] == "pick":
        sys.exit(pick())
    if len(sys.argv) > 1 and sys.argv[1] == "refine_modes":
        sys.exit(refine_modes())
    sys.exit(main())
