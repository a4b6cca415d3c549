"""TEST INFRASTRUCTURE, NOT PRODUCT — torch-CPU / NumPy restatement of estimate_pose
(poseEstSurf.py:11-261), the SurfEmb-style sample-and-score pose estimator.

What is pinned and what is not
  * pooling, log-softmax correspondence matrix, 3x3 spatial max-pool, pruning masks and the z-buffer
    score are the reference's own torch / NumPy expressions (torch is importable here) with
    torch_scatter.scatter_min / scatter_mean (absent) replaced by Tensor.scatter_reduce('amin') and
    explicit means — PARITY UNPINNED for the arg-min tie rule (torch_scatter leaves it unspecified;
    here: lowest vertex index);
  * cv2.solveP3P(AP3P) is absent: P3P = oracle/pnp_oracle.p3p_grunert, solutions ordered by the
    4th point's reprojection error as OpenCV documents for 4-point input — PARITY UNPINNED;
  * the reference draws torch.rand on the device and an UNSEEDED np.random.randint
    (poseEstSurf.py:116,140); the build makes both explicit: uniforms and the solution pick come
    from Philox4x32-10(seed) so the device path and this oracle see the same numbers.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import pnp_oracle as po


def uniforms(n_samples: int, seed: int) -> np.ndarray:
    """(n_samples, 4) f64 in (0,1): (x + 0.5) / 2^32 of Philox(counter=(s,1,0,0), key=seed)."""
    ctr = np.zeros((n_samples, 4), np.uint32)
    ctr[:, 0] = np.arange(n_samples, dtype=np.uint32)
    ctr[:, 1] = 1
    r = po.philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.float64)
    return (r + 0.5) / 4294967296.0


def picks(n_samples: int, seed: int) -> np.ndarray:
    """(n_samples,) uint32: word 0 of Philox(counter=(s,2,0,0), key=seed) — the solution pick."""
    ctr = np.zeros((n_samples, 4), np.uint32)
    ctr[:, 0] = np.arange(n_samples, dtype=np.uint32)
    ctr[:, 1] = 2
    return po.philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[:, 0]


def prepare(mask_lgts: torch.Tensor, query_img: torch.Tensor, down_sample_scale=3, max_pool=True):
    """poseEstSurf.py:47-69: -> mask_log_prob (n,), neg_mask_log_prob (n,), mask_prob (n,), queries (n,e), res."""
    e = query_img.shape[-1]
    mask_log_prob, neg_mask_log_prob = [
        F.max_pool2d(F.logsigmoid(lgts)[None], down_sample_scale)[0] for lgts in (mask_lgts, -mask_lgts)]
    ml = F.avg_pool2d(mask_lgts[None], down_sample_scale)[0]
    res = len(ml)
    n = res ** 2
    mask_prob = torch.sigmoid(ml).view(n)
    if max_pool:
        mask_log_prob = F.max_pool2d(mask_log_prob[None], 3, 1, 1)[0]
        neg_mask_log_prob = F.max_pool2d(neg_mask_log_prob[None], 3, 1, 1)[0]
    queries = F.avg_pool2d(query_img.permute(2, 0, 1), down_sample_scale).reshape(e, n).T
    return mask_log_prob.reshape(n), neg_mask_log_prob.reshape(n), mask_prob, queries.contiguous(), res


def corr_matrices(queries, obj_keys, mask_prob, res, max_pool=True):
    """poseEstSurf.py:70-71, 97-107: corr_log (n,m) (spatially max-pooled when max_pool) and the
    UNPOOLED log matrix the sampling weights come from (corr = exp(.) * mask_prob at :71, :97)."""
    m = obj_keys.shape[0]
    corr_log = torch.log_softmax(queries @ obj_keys.T, dim=1)
    corr = corr_log.clone()
    if max_pool:
        cl = corr_log.view(res, res, m).permute(2, 0, 1)
        cl = F.max_pool2d(cl, kernel_size=3, stride=1, padding=1)
        corr_log = cl.permute(1, 2, 0).reshape(res * res, m)
    return corr_log.contiguous(), corr


def corr_matrices_patch(query_img, obj_keys, res, down_sample_scale=3, max_pool=True):
    """poseEstSurf.py:72-107, avg_queries=False, without the memory patching: per-pixel log_softmax over the
    keys; the value at each block's centre pixel (offset scale // 2) is the sampling log-matrix, the block
    maximum (then the 3x3 spatial max-pool) the scoring matrix.  -> corr_log (n, m), corr_centre_log (n, m)."""
    ds = down_sample_scale
    e = query_img.shape[-1]
    m = obj_keys.shape[0]
    R = res * ds
    q = query_img[:R, :R]
    off = ds // 2
    if R * R * m > (1 << 31):
        # the reference's size (r = 224, m = 80 000): the (R, R, m) matrix is 15.8 GB — the same expressions band by
        # band of `ds` pixel rows (log_softmax is per pixel and a block never crosses a band)
        centre = torch.empty((res, res, m))
        blk = torch.empty((m, res, res))
        for b in range(res):
            band = torch.log_softmax(q[b * ds:(b + 1) * ds].reshape(ds * R, e) @ obj_keys.T, dim=1).view(ds, R, m)
            centre[b] = band[off, off::ds]
            blk[:, b] = F.max_pool2d(band.permute(2, 0, 1), ds)[:, 0]
        centre = centre.reshape(res * res, m)
    else:
        full = torch.log_softmax(q.reshape(R * R, e) @ obj_keys.T, dim=1).view(R, R, m)
        centre = full[off::ds, off::ds].reshape(res * res, m)
        blk = F.max_pool2d(full.permute(2, 0, 1), ds)                              # (m, res, res)
    if max_pool:
        blk = F.max_pool2d(blk, kernel_size=3, stride=1, padding=1)
    return blk.permute(1, 2, 0).reshape(res * res, m).contiguous(), centre.contiguous()


def sample(corr_log_raw, mask_prob, alpha, n_samples, seed, return_cum=False):
    """poseEstSurf.py:111-119 with explicit uniforms and an f64 cumulative sum.  The weight
    (exp(corr_log) * mask_prob)^alpha is evaluated in f64 as exp(alpha*corr_log) * mask_prob^alpha
    (the reference does it in f32; for a sampling distribution the difference is immaterial and the
    f64 form lets the device path reproduce the indices)."""
    w = (np.exp(alpha * corr_log_raw.double().numpy()) * (mask_prob.double().numpy() ** alpha)[:, None]).reshape(-1)
    cum = np.cumsum(w)
    u = uniforms(n_samples, seed)
    idx = np.searchsorted(cum, u * cum[-1]).astype(np.int64)        # (n_samples, 4)
    return (idx, cum) if return_cum else idx


def p3p_sorted(X, uv, K):
    """cv2.solveP3P with 4 points: solutions from the first three, ordered by the 4th point's
    reprojection error (stable)."""
    sols = po.p3p_grunert(X[:3].astype(np.float64), uv[:3].astype(np.float64), K)
    errs = []
    for R, t in sols:
        pr, z = po.project(K, R, t, X[3:4].astype(np.float64))
        errs.append(np.sum((pr[0] - uv[3]) ** 2))
    order = np.argsort(errs, kind="stable")
    return [sols[i] for i in order]


def batch_score(R, t, K, obj_pts, res, mask_log_prob, neg_mask_log_prob, corr_log):
    """poseEstSurf.py:182-237 (f32 torch-CPU), scatter_min -> scatter_reduce('amin') + lowest-index arg."""
    n, m = res * res, obj_pts.shape[0]
    n_poses = len(R)
    cam = obj_pts @ R.permute(0, 2, 1) + t[:, None]
    z = cam[..., 2]
    img = cam @ K.T
    u = (img[..., :2] / img[..., 2:]).round_()
    mask_neg = torch.any(torch.logical_or(u < 0, res <= u), dim=-1)
    u = u[..., 1].mul_(res).add_(u[..., 0])
    u[mask_neg] = n
    u = u.long()
    zmin = torch.full((n_poses, n + 1), float("inf")).scatter_reduce(1, u, z, "amin", include_self=True)
    is_min = z == zmin.gather(1, u)
    vid = torch.arange(m).expand(n_poses, m)
    zarg = torch.full((n_poses, n + 1), m, dtype=torch.long).scatter_reduce(
        1, u, torch.where(is_min, vid, torch.full_like(vid, m)), "amin", include_self=True)
    zmin, zarg = zmin[:, :-1], zarg[:, :-1]
    mask = (zmin > 0) & torch.isfinite(zmin)
    mask_score = torch.where(mask, mask_log_prob[None].expand(n_poses, n),
                             neg_mask_log_prob[None].expand(n_poses, n)).mean(dim=1)
    coord = torch.full((n_poses,), -float("inf"))
    for p in range(n_poses):
        pix = torch.nonzero(mask[p])[:, 0]
        if len(pix):
            coord[p] = corr_log[pix, zarg[p, pix]].mean()
    mask_score = mask_score / np.log(2)
    coord = coord / np.log(m)
    return mask_score + coord, mask_score, coord


def prune_masks(poses, p2d, p3d, n3d, K, obj_diameter, res, dist_2d_min=0.1):
    """poseEstSurf.py:147-163."""
    dist_2d = np.linalg.norm(p2d[:, :3, None] - p2d[:, None, :3], axis=-1).max(axis=(1, 2))
    dist_2d_mask = dist_2d >= dist_2d_min * res
    z = poses[:, 2, 3]
    z_min = K[0, 0] * obj_diameter / (res * 20)
    z_max = K[0, 0] * obj_diameter / (res * 0.5)
    size_mask = (z_min < z) & (z < z_max)
    Rt = poses[:, :3, :3].transpose(0, 2, 1)
    n3d_cam = n3d @ Rt
    p3d_cam = p3d[:, :3] @ Rt + poses[:, None, :3, 3]
    normals_mask = np.all((n3d_cam * p3d_cam).sum(axis=-1) < 0, axis=-1)
    return dist_2d, dist_2d_mask, size_mask, normals_mask


def estimate_pose(mask_lgts, query_img, obj_pts, obj_normals, obj_keys, obj_diameter, K, max_poses=10000,
                  max_pose_evaluations=1000, down_sample_scale=3, alpha=1.5, dist_2d_min=0.1,
                  pose_batch_size=500, max_pool=True, do_prune=True, poses=None, seed=0):
    """Returns (R, t, pose_scores, mask_scores, coord_scores, dist_2d, size_mask, normals_mask) with
    R/t/scores as f32 torch-CPU tensors, plus a dict of intermediates for stage-wise parity tests."""
    K = np.asarray(K, np.float64).copy()
    K[:2, 2] += 0.5
    K[:2] /= down_sample_scale
    K[:2, 2] -= 0.5
    mlp, nmlp, mprob, queries, res = prepare(mask_lgts, query_img, down_sample_scale, max_pool)
    n, m = res * res, obj_keys.shape[0]
    corr_log, corr = corr_matrices(queries, obj_keys, mprob, res, max_pool)
    inter = dict(mask_log_prob=mlp, neg_mask_log_prob=nmlp, mask_prob=mprob, queries=queries, corr_log=corr_log,
                 res=res, K=K)
    dist_2d = size_mask = normals_mask = None
    if poses is None:
        corr_idx = sample(corr, mprob, alpha, max_poses, seed)
        p2d_idx, p3d_idx = corr_idx // m, corr_idx % m
        p2d = np.stack([p2d_idx % res, p2d_idx // res], axis=-1).astype(np.float64)     # (x, y)
        pts_np = obj_pts.numpy().astype(np.float64)
        p3d = pts_np[p3d_idx]
        n3d = np.asarray(obj_normals)[p3d_idx[:, :3]]
        pk = picks(max_poses, seed)
        poses = np.zeros((max_poses, 3, 4))
        pmask = np.zeros(max_poses, bool)
        for i in range(max_poses):
            sols = p3p_sorted(p3d[i], p2d[i], K) if len(set(corr_idx[i].tolist())) == 4 else []
            if sols:
                j = int((int(pk[i]) * len(sols)) >> 32)
                poses[i, :, :3], poses[i, :, 3] = sols[j]
                pmask[i] = True
        inter.update(corr_idx=corr_idx, poses_all=poses.copy(), poses_mask=pmask.copy())
        poses, p2d, p3d, n3d = [a[pmask] for a in (poses, p2d, p3d, n3d)]
        dist_2d, d2m, size_mask, normals_mask = prune_masks(poses, p2d, p3d, n3d, K, obj_diameter, res, dist_2d_min)
        if do_prune:
            poses = poses[d2m & size_mask & normals_mask]
    poses = poses[:max_pose_evaluations]
    R = torch.from_numpy(poses[:, :3, :3]).float()
    t = torch.from_numpy(poses[:, :3, 3]).float()
    Kt = torch.from_numpy(K).float()
    ps, ms, cs = [], [], []
    for l in range(0, len(R), pose_batch_size):
        a, b, c = batch_score(R[l:l + pose_batch_size], t[l:l + pose_batch_size], Kt, obj_pts.float(), res, mlp, nmlp,
                              corr_log)
        ps.append(a), ms.append(b), cs.append(c)
    cat = lambda xs: torch.cat(xs) if xs else torch.empty(0)
    return (R, t, cat(ps), cat(ms), cat(cs), dist_2d, size_mask, normals_mask), inter
