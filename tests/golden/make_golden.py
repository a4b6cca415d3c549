#!/usr/bin/env python3
"""Generates tests/golden/*.npz — small input/expected-output vectors for the hot path.

The reference (/root/reference) ships no tests, fixtures or data, and its hot-path scripts cannot
be imported (argv parsing / cuda:0 / cv2 / open3d at import), so nothing here is produced by
running reference code.  Expected outputs come from
  * the literal reference EXPRESSIONS evaluated with the reference's own libraries where those
    exist in this image: torch-CPU (getCors, the top-80 % filter: inference.py:142-149, 282-290)
    and sklearn KDTree (ADDS: inference.py:118-120);
  * scipy cKDTree exact 1-NN for the Open3D distance calls (verfication.py:97-101);
  * the build's own oracle for the stages whose library (OpenCV, Open3D ICP) is absent —
    those vectors pin the oracle against regressions, not against the reference.
Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from imagesequenceregistrationfor6dposeestimationlabeling_amd import synth  # noqa: E402
from oracle import pnp_oracle, registration_oracle as ro  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    rng = np.random.default_rng(2024)
    # --- getCors + filter at the reference's descriptor size (D = 12), literal torch expressions
    keys = synth.unit_keys(rng, 600, 12)
    gt = rng.integers(600, size=700)
    Q = (keys[gt] + 0.35 * rng.normal(size=(700, 12))).astype(np.float32)
    idx, vals = ro.getCors(torch.from_numpy(Q), torch.from_numpy(keys), 1)
    nidx = ro.filter_top(vals)
    np.savez_compressed(OUT / "getcors_d12.npz", Q=Q, K=keys, idx=idx.numpy(), vals=vals.numpy(), nidx=nidx)
    # small-n branch of the filter (n <= 500)
    v2 = vals[:123]
    np.savez_compressed(OUT / "filter_small.npz", vals=v2.numpy(), nidx=ro.filter_top(v2))

    # --- ADD / ADD-S with the reference's sklearn call
    S = synth.tless_like(rng, 1200)
    V = synth.tless_like(rng, 500)
    R, t = synth.random_poses(rng, 2)
    Rp, tp = synth.perturb_pose(rng, R[0], t[0], 4.0, 3.0)
    np.savez_compressed(OUT / "adds.npz", S=S, V=V, gtR=R[0], gtT=t[0], R=Rp, T=tp,
                        add=ro.ADD(V.astype(np.float64), R[0], t[0], Rp, tp),
                        adds=ro.ADDS(V.astype(np.float64), R[0], t[0], Rp, tp, S.astype(np.float64)))

    # --- Chamfer (exact NN) and the consecutive-pair loop
    pc = synth.bumpy_ellipsoid(rng, 900)
    n = 5
    Rg, tg = synth.random_poses(rng, n)
    Rpred = np.array([synth.perturb_pose(rng, Rg[i], tg[i], 2.0 if i != 2 else 30.0, 0)[0] for i in range(n)])
    Rrel = np.array([ro.calculate_relative_pose(Rg[i], tg[i], Rg[i + 1], tg[i + 1])[0] for i in range(n - 1)])
    np.savez_compressed(OUT / "chamfer_pairs.npz", pc=pc, R_pred=Rpred, R_rel=Rrel,
                        chamfer=ro.chamfer_pairs(pc.astype(np.float64), Rpred, Rrel))

    # --- vote (n = 4)
    n = 4
    Rg, tg = synth.random_poses(rng, n)
    P = [synth.perturb_pose(rng, Rg[i], tg[i], 3.0 if i != 1 else 70.0, 2.0) for i in range(n)]
    Rq, tq = np.array([p[0] for p in P]), np.array([p[1] for p in P])
    gt_rel, pr_rel = ro.rel_pose_table(Rg, tg), ro.rel_pose_table(Rq, tq)
    diam = synth.diameter(S)
    err, adds = ro.vote(V.astype(np.float64), S.astype(np.float64), gt_rel, pr_rel, diam)
    np.savez_compressed(OUT / "vote.npz", S=S, V=V, gt_rel=gt_rel, pred_rel=pr_rel, diameter=diam, error=err,
                        adds=adds, R=Rg, t=tg)

    # --- PnP + RANSAC (owned algorithm; pins the oracle)
    pts = synth.tless_like(rng, 1500)
    K = synth.camera()
    R, t = synth.random_poses(rng, 1)
    p3d, p2d, inl = synth.pnp_case(rng, pts, K, R[0], t[0], 1200)
    o = pnp_oracle.pnp_ransac(p3d, p2d, K, H=100, reperr=2.0, seed=42, confidence=1.0)   # every hypothesis scored
    np.savez_compressed(OUT / "pnp_ransac.npz", p3d=p3d, p2d=p2d, K=K, R=R[0], t=t[0], seed=42, H=100,
                        samples=o["samples"], n_inl=o["n_inl"], best=o["best"], inliers=o["inliers"],
                        pose=o["Rt"], ok=o["ok"])

    # --- ICP (config-1 style, smaller)
    cloud = synth.bumpy_ellipsoid(rng, 6000)
    upper, lower = synth.split_halves(rng, cloud, 1500)
    cad = synth.bumpy_ellipsoid(rng, 1500)
    Rg, tg = synth.random_poses(rng, 1)
    Rp, tp = synth.perturb_pose(rng, Rg[0], tg[0], 3.0, 3.0)
    src = (upper.astype(np.float64) @ Rg[0].T + tg[0]).astype(np.float32)
    init = np.linalg.inv(np.vstack([np.hstack([Rp, tp[:, None]]), [0, 0, 0, 1]]))
    T, fit, rmse, traj = ro.icp_point_to_point(src, lower, 20, init)
    f0, r0, _, _ = ro.evaluate_registration(src, lower, 20, init)
    np.savez_compressed(OUT / "icp.npz", source=src, target=lower, cad=cad, init=init, T=T, fitness=fit, rmse=rmse,
                        fitness0=f0, rmse0=r0, n_iter=len(traj) - 1, final_chamfer=ro.final_chamfer(src, lower, T, cad))
    print("golden vectors written to", OUT)


def pnp_adaptive():
    """PnP + RANSAC with the adaptive termination in force (confidence 0.99, cv2's default): 50 % wrong
    matches make the staged loop stop at 96 of 500 hypotheses, and the best of those is not the best of
    all 500 — the vector pins WHERE the loop stops.  Own RNG stream: does not disturb the files above."""
    rng = np.random.default_rng(22)
    pts = synth.tless_like(rng, 4000)
    K = synth.camera()
    R, t = synth.random_poses(rng, 1)
    p3d, p2d, inl = synth.pnp_case(rng, pts, K, R[0], t[0], 4000, 0.5, 0.5)
    o = pnp_oracle.pnp_ransac(p3d, p2d, K, H=500, reperr=2.0, seed=22, confidence=0.99)
    full = pnp_oracle.pnp_ransac(p3d, p2d, K, H=500, reperr=2.0, seed=22, confidence=1.0)
    assert o["n_eval"] == 96 and o["best"] != full["best"]
    np.savez_compressed(OUT / "pnp_ransac_conf99.npz", p3d=p3d, p2d=p2d, K=K, R=R[0], t=t[0], seed=22, H=500,
                        confidence=0.99, n_eval=o["n_eval"], n_inl=o["n_inl"], best=o["best"], inliers=o["inliers"],
                        pose=o["Rt"], best_of_all=full["best"])
    print("pnp_ransac_conf99.npz written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pnp_adaptive":
        pnp_adaptive()
    else:
        main()
        pnp_adaptive()
