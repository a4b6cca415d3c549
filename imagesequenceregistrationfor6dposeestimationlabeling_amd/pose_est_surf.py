"""estimate_pose() with the reference's signature (poseEstSurf.py:11-15, returns :258-261), the device
stages in csrc/estimate_pose.hip + isr_corr_logsoftmax.

What runs on the device: mask / query pooling, the (n x m) log-softmax correspondence matrix (or its per-pixel
variant for avg_queries=False) and its 3x3 spatial max-pool, inversion sampling (no 4.4e8-element cumsum), the
10 000-iteration cv2.solveP3P Python loop (one thread per sample), the pruning masks and the ordered selection
of the poses to score (poseEstSurf.py:147-177), and batch_score's scatter_min z-buffer.  The host sees one
integer (how many poses survive: it sizes the returned tensors) and the three mask arrays the signature
returns as NumPy.  `seed` is an addition: the reference draws torch.rand on the device and an unseeded
np.random.randint.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import ops
from ._capi import IsrError, check, current_stream, lib, ptr, require_cuda
from .registration import SOLVEPNP_AP3P, _dev


def _k_scaled(K, scale):
    K = np.asarray(K, np.float64).copy()         # the reference copies K too (poseEstSurf.py:42)
    K[:2, 2] += 0.5
    K[:2] /= scale
    K[:2, 2] -= 0.5
    return K


def _kptr(K):
    arr = (ctypes.c_double * 9)(*np.asarray(K, np.float64).reshape(9).tolist())
    return arr, ctypes.cast(arr, ctypes.c_void_p)


def prepare(mask_lgts: torch.Tensor, query_img: torch.Tensor, scale: int = 3, max_pool: bool = True):
    """isr_ep_prepare -> mask_log_prob (n,), neg_mask_log_prob (n,), mask_prob (n,), queries (n,e), res."""
    dev = require_cuda(mask_lgts, query_img)
    ml = mask_lgts.to(torch.float32).contiguous()
    qi = query_img.to(torch.float32).contiguous()
    r, e = ml.shape[0], qi.shape[-1]
    if ml.shape != (r, r) or qi.shape != (r, r, e):
        raise ValueError(f"mask_lgts {tuple(ml.shape)} / query_img {tuple(qi.shape)} must be (r,r) / (r,r,e)")
    res = r // scale
    n = res * res
    mlp = torch.empty(n, dtype=torch.float32, device=dev)
    nmlp = torch.empty(n, dtype=torch.float32, device=dev)
    mp = torch.empty(n, dtype=torch.float32, device=dev)
    q = torch.empty((n, e), dtype=torch.float32, device=dev)
    ws = ops.workspace(dev, 8 * n + 1024, "ep")
    with torch.cuda.device(dev):
        rc = lib().isr_ep_prepare(ptr(ml), ptr(qi), r, e, scale, int(bool(max_pool)), ptr(mlp), ptr(nmlp), ptr(mp),
                                  ptr(q), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_ep_prepare")
    return mlp, nmlp, mp, q, res


def pool_corr(corr_log: torch.Tensor, res: int) -> torch.Tensor:
    dev = require_cuda(corr_log)
    out = torch.empty_like(corr_log)
    with torch.cuda.device(dev):
        rc = lib().isr_ep_pool_corr(ptr(corr_log), res, corr_log.shape[1], ptr(out), current_stream(dev))
    check(rc, "isr_ep_pool_corr")
    return out


def corr_matrices(queries: torch.Tensor, obj_keys: torch.Tensor, res: int, max_pool: bool = True):
    """isr_ep_corr_matrices (avg_queries=True, poseEstSurf.py:70, 97-107): the (n, m) log-softmax matrix and — in the
    same pass over the output — its 3 x 3 spatially max-pooled twin (None when max_pool is False)."""
    dev = require_cuda(queries, obj_keys)
    q, k = queries.to(torch.float32).contiguous(), obj_keys.to(torch.float32).contiguous()
    n, e = q.shape
    m = k.shape[0]
    if n != res * res:
        raise ValueError(f"corr_matrices: {n} query rows for a {res} x {res} grid")
    raw = torch.empty((n, m), dtype=torch.float32, device=dev)
    pooled = torch.empty((n, m), dtype=torch.float32, device=dev) if max_pool else None
    L = lib()
    ws = ops.workspace(dev, L.isr_corr_logsoftmax_workspace_bytes(n, m, e, 1), "corr_lsm")
    with torch.cuda.device(dev):
        rc = L.isr_ep_corr_matrices(ptr(q), ptr(k), int(res), m, e, ptr(raw), ptr(pooled), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_ep_corr_matrices")
    return raw, pooled


def patch_corr(query_img: torch.Tensor, obj_keys: torch.Tensor, scale: int = 3):
    """isr_ep_patch_corr (avg_queries=False, poseEstSurf.py:72-96) -> corr_centre (n, m), corr_blockmax (n, m), res."""
    dev = require_cuda(query_img, obj_keys)
    qi, ok = query_img.to(torch.float32).contiguous(), obj_keys.to(torch.float32).contiguous()
    r, e = qi.shape[0], qi.shape[-1]
    m = ok.shape[0]
    res = r // scale
    centre = torch.empty((res * res, m), dtype=torch.float32, device=dev)
    bmax = torch.empty((res * res, m), dtype=torch.float32, device=dev)
    L = lib()
    ws = ops.workspace(dev, L.isr_ep_patch_corr_workspace_bytes(r, m, e), "corr_lsm")
    with torch.cuda.device(dev):
        rc = L.isr_ep_patch_corr(ptr(qi), ptr(ok), r, e, int(scale), m, ptr(centre), ptr(bmax), ptr(ws), ws.numel(),
                                 current_stream(dev))
    check(rc, "isr_ep_patch_corr")
    return centre, bmax, res


def sample(corr_log: torch.Tensor, mask_prob: torch.Tensor, alpha: float, n_samples: int, seed: int) -> torch.Tensor:
    dev = require_cuda(corr_log, mask_prob)
    n, m = corr_log.shape
    idx = torch.empty((n_samples, 4), dtype=torch.int64, device=dev)
    L = lib()
    ws = ops.workspace(dev, L.isr_ep_sample_workspace_bytes(n, m), "ep_sample")
    with torch.cuda.device(dev):
        rc = L.isr_ep_sample(ptr(corr_log), ptr(mask_prob), n, m, float(alpha), int(n_samples),
                             seed & 0xFFFFFFFFFFFFFFFF, ptr(idx), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_ep_sample")
    return idx


class DescriptorGrid:
    """What the matrix-free stages read instead of an (n, m) matrix: the grid of query descriptors, the log-sum-exp of
    each grid pixel's logits over all keys (K1's exact-f32 path) and the keys.  Element (grid pixel g, key k) =
    <q[g], keys[k]> - lse[g], bit for bit the entry corr_matrices / patch_corr would have written.
    avg_queries=True : the pooled queries, one grid pixel per output pixel (win = 1, poseEstSurf.py:70);
    avg_queries=False: the crop's own pixels, win = scale grid pixels per output pixel (poseEstSurf.py:72-96)."""

    def __init__(self, q: torch.Tensor, keys: torch.Tensor, res: int, pitch: int, win: int, lse: torch.Tensor = None):
        require_cuda(q, keys)
        self.q = q.to(torch.float32).contiguous().view(-1, q.shape[-1])
        self.keys = keys.to(torch.float32).contiguous()
        self.res, self.pitch, self.win, self.e = int(res), int(pitch), int(win), int(self.q.shape[1])
        if self.keys.shape[1] != self.e or self.pitch < self.res * self.win or self.q.shape[0] < self.pitch * (self.res * self.win - 1) + self.res * self.win:
            raise ValueError(f"DescriptorGrid: q {tuple(self.q.shape)} keys {tuple(self.keys.shape)} res {res} pitch {pitch} win {win}")
        # lse: the rows' log-sum-exps when the caller already has them (one K1 launch over a block of images: a row's
        # result does not depend on the launch it rides in)
        self.lse = self.row_lse(self.q, self.keys) if lse is None else lse.to(torch.float32).contiguous()
        if self.lse.shape != (self.q.shape[0],):
            raise ValueError(f"DescriptorGrid: lse {tuple(self.lse.shape)} for {self.q.shape[0]} grid pixels")

    @staticmethod
    def row_lse(q_rows: torch.Tensor, keys: torch.Tensor) -> torch.Tensor:
        return ops.corr_lse(q_rows, keys)      # K1, lse only (the bits of corr_argmax(..., want_lse=True)[2])

    @classmethod
    def pooled(cls, queries: torch.Tensor, keys: torch.Tensor, res: int, lse: torch.Tensor = None):
        return cls(queries, keys, res, res, 1, lse)

    @classmethod
    def per_pixel(cls, query_img: torch.Tensor, keys: torch.Tensor, scale: int, lse: torch.Tensor = None):
        return cls(query_img, keys, query_img.shape[0] // scale, query_img.shape[1], scale, lse)


def sample_direct(grid: DescriptorGrid, mask_prob: torch.Tensor, alpha: float, n_samples: int, seed: int) -> torch.Tensor:
    """isr_ep_sample_direct: sample()'s indices without the matrix."""
    dev = require_cuda(mask_prob)
    n, m = grid.res * grid.res, grid.keys.shape[0]
    idx = torch.empty((n_samples, 4), dtype=torch.int64, device=dev)
    L = lib()
    ws = ops.workspace(dev, L.isr_ep_sample_workspace_bytes(n, m), "ep_sample")
    with torch.cuda.device(dev):
        rc = L.isr_ep_sample_direct(ptr(grid.q), ptr(grid.lse), grid.pitch, grid.e, grid.win, grid.res, ptr(mask_prob),
                                    ptr(grid.keys), m, float(alpha), int(n_samples), seed & 0xFFFFFFFFFFFFFFFF, ptr(idx),
                                    ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_ep_sample_direct")
    return idx


def sample_weights(corr_log: torch.Tensor, mask_prob: torch.Tensor, alpha: float) -> torch.Tensor:
    """isr_ep_sample_weights: the (n, m) f64 weights the sampler adds (validation aid)."""
    dev = require_cuda(corr_log, mask_prob)
    n, m = corr_log.shape
    w = torch.empty((n, m), dtype=torch.float64, device=dev)
    L = lib()
    ws = ops.workspace(dev, L.isr_ep_sample_workspace_bytes(n, m), "ep_sample")
    with torch.cuda.device(dev):
        rc = L.isr_ep_sample_weights(ptr(corr_log.contiguous()), ptr(mask_prob), n, m, float(alpha), ptr(w), ptr(ws), ws.numel(),
                                     current_stream(dev))
    check(rc, "isr_ep_sample_weights")
    return w


def p3p_samples(corr_idx: torch.Tensor, res: int, m: int, obj_pts: torch.Tensor, K, seed: int):
    dev = require_cuda(corr_idx, obj_pts)
    S = corr_idx.shape[0]
    poses = torch.empty((S, 3, 4), dtype=torch.float64, device=dev)
    ok = torch.empty(S, dtype=torch.uint8, device=dev)
    keep, kp = _kptr(K)
    with torch.cuda.device(dev):
        rc = lib().isr_ep_p3p(ptr(corr_idx), res, m, ptr(obj_pts), kp, S, seed & 0xFFFFFFFFFFFFFFFF, ptr(poses), ptr(ok),
                              current_stream(dev))
    check(rc, "isr_ep_p3p")
    return poses, ok


def prune(corr_idx, poses, ok, obj_pts, obj_normals, res: int, m: int, K00: float, obj_diameter: float,
          dist_2d_min: float = 0.1, do_prune: bool = True, max_eval: int = 1000):
    """isr_ep_prune (poseEstSurf.py:147-177) -> dist_2d (S) f32, size_mask, normals_mask, keep (S) u8,
    keep_idx (S) i32 (first n_keep valid, ascending), n_keep (1) i32, Rt32 (max_eval, 3, 4) f32 — all on the device."""
    dev = require_cuda(corr_idx, poses, ok, obj_pts, obj_normals)
    S = corr_idx.shape[0]
    normals = obj_normals.to(torch.float64).contiguous()
    dist = torch.empty(S, dtype=torch.float32, device=dev)
    sm = torch.empty(S, dtype=torch.uint8, device=dev)
    nm = torch.empty(S, dtype=torch.uint8, device=dev)
    keep = torch.empty(S, dtype=torch.uint8, device=dev)
    kidx = torch.empty(S, dtype=torch.int32, device=dev)
    nk = torch.empty(1, dtype=torch.int32, device=dev)
    Rt32 = torch.empty((max_eval, 3, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().isr_ep_prune(ptr(corr_idx), ptr(poses.contiguous()), ptr(ok), ptr(obj_pts), ptr(normals), S, int(res), int(m),
                                float(K00), float(obj_diameter), float(dist_2d_min), int(bool(do_prune)), int(max_eval),
                                ptr(dist), ptr(sm), ptr(nm), ptr(keep), ptr(kidx), ptr(nk), ptr(Rt32), current_stream(dev))
    check(rc, "isr_ep_prune")
    return dist, sm, nm, keep, kidx, nk, Rt32


def zbuf_score(obj_pts, R, t, K, res, mask_log_prob, neg_mask_log_prob, corr_log):
    """batch_score (poseEstSurf.py:182-237) for R (B,3,3), t (B,3) f32 on the device -> 3 x (B,) f32."""
    dev = require_cuda(obj_pts, R, t)
    B, m = R.shape[0], obj_pts.shape[0]
    Rt = torch.cat([R.to(torch.float32), t.to(torch.float32)[:, :, None]], dim=2).contiguous()
    out = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(3)]
    L = lib()
    ws = ops.workspace(dev, L.isr_zbuf_score_workspace_bytes(B, res), "zbuf")
    keep, kp = _kptr(K)
    with torch.cuda.device(dev):
        rc = L.isr_zbuf_score(ptr(obj_pts), m, ptr(Rt), B, kp, res, ptr(mask_log_prob), ptr(neg_mask_log_prob),
                              ptr(corr_log), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(ws), ws.numel(),
                              current_stream(dev))
    check(rc, "isr_zbuf_score")
    return out


def zbuf_score_direct(obj_pts, R, t, K, res, mask_log_prob, neg_mask_log_prob, grid: DescriptorGrid, max_pool: bool = True):
    """isr_zbuf_score_direct: zbuf_score()'s three scores on the [3x3 max-pooled] matrix, without the matrix."""
    dev = require_cuda(obj_pts, R, t)
    B, m = R.shape[0], obj_pts.shape[0]
    Rt = torch.cat([R.to(torch.float32), t.to(torch.float32)[:, :, None]], dim=2).contiguous()
    out = [torch.empty(B, dtype=torch.float32, device=dev) for _ in range(3)]
    L = lib()
    ws = ops.workspace(dev, L.isr_zbuf_score_workspace_bytes(B, res), "zbuf")
    keep, kp = _kptr(K)
    with torch.cuda.device(dev):
        rc = L.isr_zbuf_score_direct(ptr(obj_pts), m, ptr(Rt), B, kp, res, ptr(mask_log_prob), ptr(neg_mask_log_prob),
                                     ptr(grid.q), ptr(grid.lse), grid.pitch, grid.e, grid.win, int(bool(max_pool)),
                                     ptr(grid.keys), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(ws), ws.numel(),
                                     current_stream(dev))
    check(rc, "isr_zbuf_score_direct")
    return out


class _Call:
    """One estimate_pose call split at its only host round trip (the number of surviving poses sizes the outputs):
    front() launches everything up to the ordered selection of the poses to score, back(n_keep) scores them and
    assembles the reference's return tuple.  estimate_pose runs the two halves back to back; estimate_poses runs the
    fronts of a block of images on several streams, reads all the counts at once, then runs the backs."""

    def __init__(self, mask_lgts, query_img, obj_pts, obj_normals, obj_keys, obj_diameter, K, max_poses=10000,
                 max_pose_evaluations=1000, down_sample_scale=3, alpha=1.5, dist_2d_min=0.1, pnp_method=SOLVEPNP_AP3P,
                 pose_batch_size=500, max_pool=True, avg_queries=True, do_prune=True, visualize=False, poses=None,
                 debug=False, returnPoints=False, *, seed=0, materialize=False):
        del pnp_method
        if visualize:
            raise IsrError("estimate_pose(visualize=True) needs cv2.imshow; not available")
        self.mask_lgts, self.query_img = _dev(mask_lgts, torch.float32), _dev(query_img, torch.float32)
        self.obj_pts = _dev(obj_pts, torch.float32).contiguous()
        self.obj_keys = _dev(obj_keys, torch.float32).contiguous()
        self.dev = self.mask_lgts.device
        self.obj_normals, self.obj_diameter = obj_normals, obj_diameter
        self.Ks = _k_scaled(K, down_sample_scale)
        self.max_poses, self.max_eval, self.scale, self.alpha = max_poses, max_pose_evaluations, down_sample_scale, alpha
        self.dist_2d_min, self.batch, self.max_pool, self.avg_queries = dist_2d_min, pose_batch_size, max_pool, avg_queries
        self.do_prune, self.poses, self.debug, self.returnPoints = do_prune, poses, debug, returnPoints
        self.seed, self.materialize = seed, materialize
        self.nk_d = self.res = None

    def pool(self):
        """:47-69; returns the rows whose log-sum-exps the matrix-free route needs (None on the materialised route)."""
        self.mlp, self.nmlp, self.mprob, self.queries, self.res = prepare(self.mask_lgts, self.query_img, self.scale, self.max_pool)
        if self.materialize:
            return None
        return self.queries if self.avg_queries else self.query_img.reshape(-1, self.query_img.shape[-1])

    def front(self, lse=None):
        m = self.obj_keys.shape[0]
        if self.res is None:
            self.pool()
        res, mprob, queries = self.res, self.mprob, self.queries
        self.grid = self.corr_log = None
        if not self.materialize:
            self.grid = (DescriptorGrid.pooled(queries, self.obj_keys, res, lse) if self.avg_queries
                         else DescriptorGrid.per_pixel(self.query_img, self.obj_keys, self.scale, lse))
        elif self.avg_queries:
            corr_raw, self.corr_log = corr_matrices(queries, self.obj_keys, res, self.max_pool)   # (n, m) f32 each, :70 and :97-107
            if self.corr_log is None:
                self.corr_log = corr_raw
        else:
            # :72-96: per-pixel log-softmax; block-centre values feed the sampler, block maxima the scores
            corr_raw, corr_blk, _ = patch_corr(self.query_img, self.obj_keys, self.scale)
            self.corr_log = pool_corr(corr_blk, res) if self.max_pool else corr_blk
        if self.poses is not None:
            return
        self.corr_idx = (sample_direct(self.grid, mprob, self.alpha, self.max_poses, self.seed) if self.grid is not None
                         else sample(corr_raw, mprob, self.alpha, self.max_poses, self.seed))
        poses_d, self.ok = p3p_samples(self.corr_idx, res, m, self.obj_pts, self.Ks, self.seed)
        # normals_scaled.npy is float64 (:121); a device tensor is taken as it is (a NumPy array is 1.9 MB of upload per call)
        normals_d = (self.obj_normals.to(self.dev, torch.float64) if torch.is_tensor(self.obj_normals)
                     else _dev(np.asarray(self.obj_normals, np.float64)))
        (self.dist_d, self.sm_d, self.nm_d, _, self.kidx_d, self.nk_d, self.Rt32) = prune(
            self.corr_idx, poses_d, self.ok, self.obj_pts, normals_d, res, m, self.Ks[0, 0], self.obj_diameter, self.dist_2d_min,
            self.do_prune, self.max_eval)

    def back(self, n_keep=None):
        dev, res, m = self.dev, self.res, self.obj_keys.shape[0]
        dist_2d = size_mask = normals_mask = None
        p3dCp = p2dCp = None
        if self.poses is None:
            n_poses = min(n_keep, self.max_eval)
            # the reference's arrays have one entry per SOLVED sample (poses_mask, :145), in sample order
            pmask = self.ok.cpu().numpy().astype(bool)
            dist_2d = self.dist_d.cpu().numpy()[pmask]
            size_mask = self.sm_d.cpu().numpy().astype(bool)[pmask]
            normals_mask = self.nm_d.cpu().numpy().astype(bool)[pmask]
            R, t = self.Rt32[:n_poses, :, :3].contiguous(), self.Rt32[:n_poses, :, 3].contiguous()
            if self.returnPoints:
                kept = self.kidx_d[:n_keep].long()
                ci = self.corr_idx[kept]
                p2dCp = torch.stack([(ci // m) % res, (ci // m) // res], dim=-1).float().cpu().numpy()   # img_pts[p2d_idx].float()
                p3dCp = self.obj_pts[ci % m].cpu().numpy()
        else:
            poses = np.asarray(self.poses)[slice(None, self.max_eval)]
            n_poses = len(poses)
            R = torch.from_numpy(np.ascontiguousarray(poses[:, :3, :3])).float().to(dev)
            t = torch.from_numpy(np.ascontiguousarray(poses[:, :3, 3])).float().to(dev)
        if self.debug:
            print('n_poses', n_poses)
        pose_scores = torch.empty(n_poses, device=dev)
        mask_scores = torch.empty(n_poses, device=dev)
        coord_scores = torch.empty(n_poses, device=dev)
        # the reference batches to bound its (poses, n + 1) scatter buffers; the matrix-free scorer has nothing to bound
        batch = 65535 if self.grid is not None else self.batch
        for l in range(0, n_poses, batch):
            Rl, tl = R[l:l + batch], t[l:l + batch]
            ps, ms, cs = (zbuf_score_direct(self.obj_pts, Rl, tl, self.Ks, res, self.mlp, self.nmlp, self.grid, self.max_pool)
                          if self.grid is not None
                          else zbuf_score(self.obj_pts, Rl, tl, self.Ks, res, self.mlp, self.nmlp, self.corr_log))
            pose_scores[l:l + batch] = ps
            mask_scores[l:l + batch] = ms
            coord_scores[l:l + batch] = cs
        if self.returnPoints:
            return R, t, pose_scores, mask_scores, coord_scores, dist_2d, size_mask, normals_mask, p3dCp, p2dCp
        return R, t, pose_scores, mask_scores, coord_scores, dist_2d, size_mask, normals_mask


def estimate_pose(mask_lgts, query_img, obj_pts, obj_normals, obj_keys, obj_diameter, K, max_poses=10000,
                  max_pose_evaluations=1000, down_sample_scale=3, alpha=1.5, dist_2d_min=0.1,
                  pnp_method=SOLVEPNP_AP3P, pose_batch_size=500, max_pool=True, avg_queries=True, do_prune=True,
                  visualize=False, poses=None, debug=False, returnPoints=False, *, seed=0, materialize=False):
    """poseEstSurf.py:11-261.  mask_lgts (r,r), query_img (r,r,e), obj_pts (m,3), obj_keys (m,e) on the
    device (host arrays are uploaded); obj_normals a NumPy (m,3) array as in the reference (:121) or a device tensor.
    Returns R (n_poses,3,3) f32, t (n_poses,3) f32, pose_scores, mask_scores, coord_scores (device),
    dist_2d, size_mask, normals_mask (NumPy, pre-prune length) [+ p3dCp, p2dCp if returnPoints].
    materialize=False (default): the (n, m) correspondence matrices of :70-107 are never formed — the sampler and the
    scorer compute the elements they need from the descriptors (DescriptorGrid); True: the round-2/3 route through the
    arrays (corr_matrices / patch_corr + pool_corr).  Both return the same bits."""
    # isr_estimate_pose scores every surviving pose in ONE scorer launch (pose = blockIdx.y: at most 65 535); a larger
    # max_pose_evaluations (or do_prune=False with a large max_poses) takes the staged route below, which scores in
    # pose_batch_size batches as the reference does (poseEstSurf.py:225-237)
    if not materialize and poses is None and not returnPoints and not visualize and int(max_pose_evaluations) <= 65535:
        return _estimate_pose_one_call(mask_lgts, query_img, obj_pts, obj_normals, obj_keys, obj_diameter, K, max_poses,
                                       max_pose_evaluations, down_sample_scale, alpha, dist_2d_min, max_pool, avg_queries,
                                       do_prune, debug, seed)
    call = _Call(mask_lgts, query_img, obj_pts, obj_normals, obj_keys, obj_diameter, K, max_poses, max_pose_evaluations,
                 down_sample_scale, alpha, dist_2d_min, pnp_method, pose_batch_size, max_pool, avg_queries, do_prune, visualize,
                 poses, debug, returnPoints, seed=seed, materialize=materialize)
    call.front()
    return call.back(None if call.nk_d is None else int(call.nk_d.item()))     # the one host round trip: it sizes the outputs


def _estimate_pose_one_call(mask_lgts, query_img, obj_pts, obj_normals, obj_keys, obj_diameter, K, max_poses, max_eval, scale,
                            alpha, dist_2d_min, max_pool, avg_queries, do_prune, debug, seed):
    """isr_estimate_pose: the stages of the matrix-free route issued by ONE library call (the same launches in the same
    order — the same bits as the stage-by-stage composition), one read-back for the per-sample arrays."""
    mask_lgts, query_img = _dev(mask_lgts, torch.float32).contiguous(), _dev(query_img, torch.float32).contiguous()
    dev = mask_lgts.device
    pts, keys = _dev(obj_pts, torch.float32).contiguous(), _dev(obj_keys, torch.float32).contiguous()
    normals = (obj_normals.to(dev, torch.float64) if torch.is_tensor(obj_normals)
               else _dev(np.asarray(obj_normals, np.float64))).contiguous()
    r, e, m, S = mask_lgts.shape[0], query_img.shape[-1], keys.shape[0], int(max_poses)
    if mask_lgts.shape != (r, r) or query_img.shape != (r, r, e) or keys.shape[1] != e or pts.shape != (m, 3) or normals.shape != (m, 3):
        raise ValueError(f"estimate_pose: mask_lgts {tuple(mask_lgts.shape)} query_img {tuple(query_img.shape)} obj_pts "
                         f"{tuple(pts.shape)} obj_normals {tuple(normals.shape)} obj_keys {tuple(keys.shape)}")
    L = lib()
    f32 = torch.empty(max_eval * 15, dtype=torch.float32, device=dev)          # Rt32 (max_eval, 12) | three score vectors
    Rt32 = f32[:max_eval * 12].view(max_eval, 3, 4)
    ps, ms, cs = (f32[max_eval * (12 + i):max_eval * (13 + i)] for i in range(3))
    per = torch.empty(S * 7, dtype=torch.uint8, device=dev)                    # dist_2d (S) f32 | size | normals | solved (S) u8
    ws = ops.workspace(dev, L.isr_estimate_pose_workspace_bytes(r, e, m, int(scale), S, int(max_eval), int(bool(avg_queries))),
                       "estimate_pose")
    keep, kp = _kptr(np.asarray(K, np.float64))
    n_poses, n_keep = ctypes.c_int32(0), ctypes.c_int32(0)
    base = per.data_ptr()
    with torch.cuda.device(dev):
        rc = L.isr_estimate_pose(ptr(mask_lgts), ptr(query_img), r, e, ptr(pts), ptr(normals), ptr(keys), m, float(obj_diameter), kp,
                                 S, int(max_eval), int(scale), float(alpha), float(dist_2d_min), int(bool(max_pool)),
                                 int(bool(avg_queries)), int(bool(do_prune)), seed & 0xFFFFFFFFFFFFFFFF, ptr(Rt32), ptr(ps), ptr(ms),
                                 ptr(cs), base, base + 4 * S, base + 5 * S, base + 6 * S, ctypes.byref(n_poses),
                                 ctypes.byref(n_keep), ptr(ws), ws.numel(), current_stream(dev))
    check(rc, "isr_estimate_pose")
    n = n_poses.value
    if debug:
        print('n_poses', n)
    host = per.cpu().numpy()
    pmask = host[6 * S:7 * S].astype(bool)                                      # one entry per SOLVED sample (:145), in order
    dist_2d = host[:4 * S].view(np.float32)[pmask]
    size_mask, normals_mask = host[4 * S:5 * S].astype(bool)[pmask], host[5 * S:6 * S].astype(bool)[pmask]
    R, t = Rt32[:n, :, :3].contiguous(), Rt32[:n, :, 3].contiguous()
    return R, t, ps[:n].clone(), ms[:n].clone(), cs[:n].clone(), dist_2d, size_mask, normals_mask


def estimate_poses(mask_lgts, query_imgs, obj_pts, obj_normals, obj_keys, obj_diameter, Ks, *, seeds=None, n_streams=4,
                   **kwargs):
    """estimate_pose for a block of crops of one object (the reference's per-image loop, inference.py:163, 325-331):
    mask_lgts (B, r, r), query_imgs (B, r, r, e), Ks one (3, 3) or (B, 3, 3).  Image b's result is estimate_pose's on image b
    with seed seeds[b] (default b), bit for bit — the same kernels, with ONE K1 launch for the log-sum-exps of the whole
    block, the per-image chains issued on n_streams side streams so that one image's small kernels fill the tails of
    another's, and ONE host round trip for all the survivor counts instead of one per image.  Returns a list of B tuples."""
    mask_lgts, query_imgs = _dev(mask_lgts, torch.float32), _dev(query_imgs, torch.float32)
    B = mask_lgts.shape[0]
    if mask_lgts.ndim != 3 or query_imgs.ndim != 4 or query_imgs.shape[0] != B:
        raise ValueError(f"estimate_poses: mask_lgts {tuple(mask_lgts.shape)} / query_imgs {tuple(query_imgs.shape)} must be (B,r,r) / (B,r,r,e)")
    if B == 0:
        return []
    dev = mask_lgts.device
    obj_pts_d, obj_keys_d = _dev(obj_pts, torch.float32).contiguous(), _dev(obj_keys, torch.float32).contiguous()
    normals_d = (obj_normals.to(dev, torch.float64) if torch.is_tensor(obj_normals) else _dev(np.asarray(obj_normals, np.float64)))
    Ks = np.asarray(Ks, np.float64)
    seeds = list(range(B)) if seeds is None else list(seeds)
    main = torch.cuda.current_stream(dev)
    streams = _side_streams(dev, max(1, min(n_streams, B)))
    # pooling of every image, then ONE K1 launch for the log-sum-exps of all their descriptor rows (a row's result does not
    # depend on the launch it rides in; B x 5 476 rows run at the kernel's full rate, one image's 5 476 do not fill the chip)
    calls = [_Call(mask_lgts[b], query_imgs[b], obj_pts_d, normals_d, obj_keys_d, obj_diameter, Ks if Ks.ndim == 2 else Ks[b],
                   seed=seeds[b], **kwargs) for b in range(B)]
    rows = [c.pool() for c in calls]
    lses = [None] * B
    if B and rows[0] is not None:
        lse_all = DescriptorGrid.row_lse(torch.cat(rows), obj_keys_d)
        lses = list(torch.split(lse_all, [r.shape[0] for r in rows]))
    for st in streams:
        st.wait_stream(main)
    for b, c in enumerate(calls):
        with torch.cuda.stream(streams[b % len(streams)]):
            c.front(lses[b])
    counts = None
    if calls and calls[0].nk_d is not None:
        for st in streams:
            main.wait_stream(st)
        counts = torch.cat([c.nk_d for c in calls]).cpu().tolist()          # the one round trip of the block
        for st in streams:
            st.wait_stream(main)
    out = []
    for b, c in enumerate(calls):
        with torch.cuda.stream(streams[b % len(streams)]):
            out.append(c.back(None if counts is None else counts[b]))
    for st in streams:
        main.wait_stream(st)
    for res in out:
        for x in res:
            if torch.is_tensor(x):
                x.record_stream(main)
    return out


_streams = {}


def _side_streams(dev, n):
    pool = _streams.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(dev))
    return pool[:n]
