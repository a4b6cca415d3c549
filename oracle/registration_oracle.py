"""TEST INFRASTRUCTURE, NOT PRODUCT — CPU restatement of the reference's hot-path functions with
the libraries the reference itself uses where they exist in this image (torch-CPU, sklearn
KDTree) and scipy cKDTree standing in for Open3D's exact KD-tree queries.

Citations are to /root/repo's reference at /root/reference (read as text; nothing is imported
from it: every hot-path script parses argv / needs cuda:0 / imports cv2 or open3d at import).

Pinned: getCors, the top-80 % filter (literal torch expressions), ADD, ADDS (the reference's own
sklearn call).  PARITY UNPINNED (Open3D absent, no reference tests): chamfer, evaluate_registration,
icp_point_to_point restate Open3D's documented behaviour from memory — exact 1-NN distances;
registration_icp = { correspondences within max distance; Kabsch/Umeyama without scale; T <- dT T;
stop when |d fitness| < 1e-6 and |d rmse| < 1e-6 or after 30 iterations }.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial import cKDTree
from sklearn.neighbors import KDTree


def getCors(queries: torch.Tensor, feats: torch.Tensor, leaves=1):
    """inference.py:142-149."""
    cMat = torch.log_softmax(queries @ feats.T, dim=-1)
    vals, idx = torch.topk(cMat, k=leaves, dim=-1)
    if leaves == 1:
        return idx[..., 0], vals
    return idx, vals


def getCors_chunked(queries: torch.Tensor, feats: torch.Tensor, chunk=8192):
    """Same expression evaluated in row chunks so a 640x480 image fits in host memory."""
    idx, vals = [], []
    for s in range(0, len(queries), chunk):
        i, v = getCors(queries[s:s + chunk], feats)
        idx.append(i)
        vals.append(v)
    return torch.cat(idx), torch.cat(vals)


def filter_top(in1: torch.Tensor):
    """inference.py:282-290.  in1 (P,1)."""
    if len(in1) > 500:
        perc = int(0.8 * len(in1))
        threshval = torch.sort(in1[:, 0])[0][-perc + 1]
    else:
        threshval = torch.sort(in1[:, 0])[0][-len(in1) + 1]
    return torch.where(in1[:, 0] > threshval)[0].cpu().numpy()


def ADD(verts, gtR1, gtT1, R1, T1):
    """inference.py:116-117."""
    return np.linalg.norm(verts.dot(gtR1.T) + gtT1 - verts.dot(R1.T) - T1, axis=-1).mean()


def ADDS(verts, gtR1, gtT1, R1, T1, surfacePointsScaled):
    """inference.py:118-120 (the module global made an argument)."""
    treeTr = KDTree(surfacePointsScaled.dot(R1.T) + T1, leaf_size=2)
    return treeTr.query(verts.dot(gtR1.T) + gtT1, k=1)[0].mean()


def compute_rel_poses(R1, t1, R2, t2):
    """choosePose.py:43-51."""
    return np.dot(R1.T, R2), t2 - t1


def calculate_relative_pose(R1, T1, R2, T2):
    """verfication.py:9-19."""
    RT1 = np.vstack([np.hstack((R1, T1.reshape(-1, 1))), [0, 0, 0, 1]])
    RT2 = np.vstack([np.hstack((R2, T2.reshape(-1, 1))), [0, 0, 0, 1]])
    Rel = np.dot(RT2, np.linalg.inv(RT1))
    return Rel[:3, :3], Rel[:3, -1]


def rel_pose_table(RList, TList, fn=compute_rel_poses):
    """choosePose.py:98-107."""
    n = len(TList)
    out = np.zeros((n, n, 4, 4))
    for i in range(n):
        for j in range(n):
            R, t = fn(RList[i], TList[i], RList[j], TList[j])
            T = np.eye(4)
            T[:3, :3] = R
            T[:3, 3:4] = np.asarray(t).reshape(3, 1)
            out[i][j] = T
    return out


def vote(modelVerts, surfacePointsScaled, gt_rel, pred_rel, diameter):
    """choosePose.py:121-145."""
    n0, n1 = pred_rel.shape[:2]
    error = np.zeros((n0, n1))
    adds = np.zeros((n0, n1))
    for i in range(n0):
        for j in range(n1):
            e = ADDS(modelVerts, gt_rel[i][j][:3, :3], np.squeeze(gt_rel[i][j][:3, 3:4]),
                     pred_rel[i][j][:3, :3], np.squeeze(pred_rel[i][j][:3, 3:4]), surfacePointsScaled)
            adds[i][j] = e
            if e < 0.1 * diameter:
                error[i][j] = 1
    return error, adds


def nn_dist(a, b):
    """open3d PointCloud(a).compute_point_cloud_distance(PointCloud(b)): exact 1-NN distances."""
    return cKDTree(np.asarray(b, np.float64)).query(np.asarray(a, np.float64), k=1, workers=-1)[0]


def chamfer(a, b):
    """verfication.py:97-101."""
    return (nn_dist(a, b).mean() + nn_dist(b, a).mean()) / 2


def chamfer_pairs(pc1, R_pred, R_rel_gt):
    """verfication.py:61-102 (rotation-only, as written)."""
    out = []
    for i in range(len(R_pred) - 1):
        pcgt = pc1.dot(R_pred[i].T).dot(R_rel_gt[i])
        pcpred = pc1.dot(R_pred[i + 1])
        out.append(chamfer(pcpred, pcgt))
    return np.array(out)


def evaluate_registration(source, target, threshold, T, search="f64"):
    """icp.py:97-99: (fitness, inlier_rmse, correspondences).
    search = "f64": exact nearest neighbours on doubles (cKDTree; what Open3D's KD-tree does, from memory).
    search = "f32": the DEVICE's definition of the neighbour — transformed source rounded to f32, squared distances
    in f32, lowest index on ties (oracle/isr_oracle.c:orc_nn_batched) — distances and sums still in f64.  The two
    differ only for source points with two targets equidistant to f32 rounding (a relative gap below ~1e-7)."""
    src = np.asarray(source, np.float64) @ T[:3, :3].T + T[:3, 3]
    if search == "f32":
        from . import cbind
        o = cbind.nn_batched(np.asarray(source, np.float32), np.asarray(target, np.float32), Tq=np.asarray(T, np.float64)[:3, :],
                             radius=-1.0, want_cov=False)
        j = o["nn_idx"][0]
        d = np.linalg.norm(src - np.asarray(target, np.float64)[j], axis=1)
    else:
        d, j = cKDTree(np.asarray(target, np.float64)).query(src, k=1, workers=-1)
    m = d <= threshold
    n = int(m.sum())
    return n / len(src), (float(np.sqrt((d[m] ** 2).mean())) if n else 0.0), src[m], np.asarray(target, np.float64)[j[m]]


def kabsch(P, Q):
    """Rigid T with Q ~ R P + t (Umeyama, no scale)."""
    mp, mq = P.mean(0), Q.mean(0)
    H = (P - mp).T @ (Q - mq)
    U, _, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1.0, 1.0, d]) @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mq - R @ mp
    return T


def icp_point_to_point(source, target, threshold, init, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, search="f64"):
    """icp.py:101-103 [Open3D defaults from memory].  Returns (T, fitness, rmse, trajectory).  `search`: see
    evaluate_registration."""
    T = np.asarray(init, np.float64).copy()
    fit, rmse, P, Q = evaluate_registration(source, target, threshold, T, search)
    traj = [(T.copy(), fit, rmse)]
    for _ in range(max_iter):
        if len(P) < 3:
            break
        T = kabsch(P, Q) @ T
        pf, pr = fit, rmse
        fit, rmse, P, Q = evaluate_registration(source, target, threshold, T, search)
        traj.append((T.copy(), fit, rmse))
        if abs(pf - fit) < rel_fitness and abs(pr - rmse) < rel_rmse:
            break
    return T, fit, rmse, traj


def final_chamfer(source, target, T, cad):
    """icp.py:110-117."""
    src = np.asarray(source, np.float64) @ T[:3, :3].T + T[:3, 3]
    full = np.concatenate([src, np.asarray(target, np.float64)], axis=0)
    return chamfer(full, np.asarray(cad, np.float64))
