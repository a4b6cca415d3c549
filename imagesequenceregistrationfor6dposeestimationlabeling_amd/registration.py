"""Drop-in mirror of the reference's hot-path functions, backed by libisr_hip.so.

Same names, positional order and return conventions as the reference scripts (there is no
package in the reference: the functions live inside inference.py / finalposes.py / choosePose.py
/ verfication.py / icp.py; citations per function).  Arrays go in as NumPy / torch exactly as the
reference passes them; all arithmetic of the four inner loops runs in the HIP kernels.  There is
no CPU path: without a GPU and the built library every call raises.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._capi import IsrError

# cv2 flag values, so call sites written against OpenCV keep working
SOLVEPNP_P3P = 2
SOLVEPNP_AP3P = 5

_state = {"surface_pts": None, "device": None}


def device() -> torch.device:
    """The HIP device this process registers on (LOCAL_RANK-aware via torch.cuda.current_device)."""
    if not torch.cuda.is_available():
        raise IsrError("no HIP device visible: the registration hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _dev(x, dtype=None) -> torch.Tensor:
    """Host array / tensor -> device tensor (a copy is plumbing, not a fallback)."""
    if isinstance(x, torch.Tensor):
        t = x if x.is_cuda else x.to(device())
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).to(device())
    return t if dtype is None else t.to(dtype)


def _pose12(R, T) -> np.ndarray:
    R = np.asarray(R, np.float64).reshape(3, 3)
    T = np.asarray(T, np.float64).reshape(3)
    return np.concatenate([R, T[:, None]], axis=1)


# ----------------------------------------------------------------------------- a1 getCors
def getCors(queries, feats, leaves=1):
    """inference.py:142-149 (= finalposes.py:38-45 = choosePose.py:35-42).

    cMat = log_softmax(queries @ feats.T, -1); vals, idx = topk(cMat, leaves)
    returns (idx[...,0].cpu(), vals) for leaves == 1 — idx a CPU LongTensor (P,), vals (P,1) on
    the device — else (idx.cpu() (P,leaves), vals (P,leaves)).
    bf16 inputs run the bf16 MFMA kernel, anything else the exact-f32 route (f16 planes on the matrix cores, indices of
    the f32 fmaf chain); the (P,N) matrix is never materialised for leaves <= 8."""
    q, f = _dev(queries), _dev(feats)
    if leaves == 1:
        idx, logp = ops.corr_argmax(q, f)
        return idx.to(torch.int64).cpu(), logp[:, None]
    # leaves > 1 (no call site of the reference passes one).  Up to 8: isr_corr_topk keeps each query's list in registers
    # while the keys stream by (round 3 materialised the (P, N) log-softmax matrix and ran torch.topk on it: 1.8 GB at the
    # reference's 5 625 x 80 000); more than 8 leaves still take that route.
    if leaves <= 8:
        idx, vals = ops.corr_topk(q, f, leaves)
        return idx.to(torch.int64).cpu(), vals
    cMat = ops.corr_logsoftmax(q, f)
    vals, idx = torch.topk(cMat, k=leaves, dim=-1)
    return idx.cpu(), vals


# --------------------------------------------------------------------- a2 / a3 filter + assembly
def masked_queries(imfeatsfull, cropMask, down_sample: int = 3, n_feat: int = 12):
    """inference.py:248-279 as one call: `imfeats[:, ::ds, ::ds]`, `inputMask[::ds, ::ds]`,
    `maskIds = torch.where(inputMask)`, `maskedfeats = imfeats[0][maskIds]`, `ep2d[:,0] = maskIds[1]`,
    `ep2d[:,1] = maskIds[0]`.  imfeatsfull (1, H, W, >= n_feat) on the device, cropMask (H, W[, 3])
    uint8 array or tensor.  Returns (maskedfeats (P, n_feat) f32 on the device, ep2d (P, 2) float64
    NumPy) — the objects the reference builds; P comes to the host here, as it does there.  The fused
    driver sequence.register_crop keeps it on the device."""
    feat = _dev(imfeatsfull, torch.float32)
    mask = torch.as_tensor(np.asarray(cropMask) if not isinstance(cropMask, torch.Tensor) else cropMask)
    mask = mask.to(feat.device)
    Q, pix, n_dev = ops.prep_queries(feat, mask, c0=0, D=n_feat, step=down_sample, dtype="f32")
    n = int(n_dev.item())
    return Q[:n], pix[:n].double().cpu().numpy()


def crop_inputs(rgb, mask, camparams, out_size: int = 224, useMask: bool = True, down_sample: int = 3, bbox=None):
    """inference.py:196-232 and 260-263 for one image or a batch: the bounding box of the visible mask
    (cv2.boundingRect(mask[:, :, 0])), the crop affine M and the crop's camera matrix (formats.crop_affine /
    crop_camera, host f64 as in the reference), cv2.warpAffine of image and mask, the useMask blanking and
    normalize() — box, warp, blanking and normalisation on the device (isr_mask_bbox, isr_crop_normalize).
    rgb (H, W, 3) or (B, H, W, 3) uint8 as cv2.imread delivers it (array or device tensor), mask likewise with 1
    or 3 channels, camparams (3, 3) or (B, 3, 3).  Returns (inputIM (B, 3, r, r) f32 on the device — the encoder's
    input of inference.py:232 —, cropMask (B, r, r) uint8 on the device = cropMask[:, :, 0], camMat (B, 3, 3) f64 host
    after the ::down_sample scaling, M (B, 2, 3)); a single image gives B = 1.  The only host round trip is the
    box (4 ints per image), which the reference's cv2 calls make too."""
    from . import formats
    r_t, m_t = _dev(rgb), _dev(mask)
    if r_t.ndim == 3:
        r_t, m_t = r_t[None], m_t[None]
    B = r_t.shape[0]
    K = np.broadcast_to(np.asarray(camparams, np.float64), (B, 3, 3))
    boxes = ops.mask_bbox(m_t).cpu().numpy() if bbox is None else np.asarray(bbox, np.int64).reshape(B, 4)
    M = np.stack([formats.crop_affine(tuple(int(v) for v in boxes[b]), out_size) for b in range(B)])
    cam = np.stack([formats.crop_camera(K[b], tuple(int(v) for v in boxes[b]), out_size, down_sample=down_sample)
                    for b in range(B)])
    inputIM, cropMask = ops.crop_normalize(r_t, m_t, M, out_size, useMask)
    return inputIM, cropMask, cam, M


def filter_top(in1, frac=0.8, min_n=500):
    """inference.py:282-290: threshold at the reference's order statistic, keep strictly above.
    in1 (P,1) or (P,) log-probs on the device.  Returns nidx as a NumPy int64 array (the
    reference's torch.where(...)[0].cpu().numpy())."""
    x = _dev(in1, torch.float32).reshape(-1)
    keep, M, _ = ops.select_top(x, frac, min_n)
    return keep[: int(M.item())].to(torch.int64).cpu().numpy()


# ---------------------------------------------------------------------------------- a5 pnp
def pnp(h3d, h2d, cam, itr=100, reperr=2, flag=SOLVEPNP_P3P, gtR=None, gtT=None, spts=None, ret=None,
        *, seed=0, refine_iters=10, confidence=0.99):
    """inference.py:123-134 (= finalposes.py:20-30 = choosePose.py:23-33).

    cv2.solvePnPRansac(h3d, h2d, cam, None, iterationsCount=itr, reprojectionError=reperr,
    flags=P3P) -> (Rodrigues(rvec) (3,3) f64, tvec (3,) f64, inlier indices (k,) int32), or the
    reference's int sentinel (1, 1, 1) after printing its message.  gtR/gtT/spts/ret are accepted
    and ignored, as in the reference.  `itr` is the maximum number of hypotheses; as with cv2's default
    confidence=0.99 (the reference passes none) scoring stops once the best inlier count gives that
    confidence (confidence=1 scores all).  `seed` keys the Philox sampler (OpenCV's RNG is internal)."""
    del flag, gtR, gtT, spts, ret
    p3d, p2d = _dev(h3d, torch.float32), _dev(h2d, torch.float32)
    if p3d.ndim != 2 or p3d.shape[1] != 3 or p2d.shape != (p3d.shape[0], 2):
        raise ValueError(f"pnp: h3d {tuple(p3d.shape)} / h2d {tuple(p2d.shape)} must be (M,3)/(M,2)")
    if p3d.shape[0] == 0:
        print("pose could not be estimated with these correspondences")
        return 1, 1, 1
    r = ops.pnp_ransac(p3d, p2d, np.asarray(cam, np.float64), H=int(itr), reperr=float(reperr),
                       seed=int(seed), refine_iters=int(refine_iters), confidence=float(confidence))
    if int(r.status.item()) != 1:
        print("pose could not be estimated with these correspondences")
        return 1, 1, 1
    pose = r.pose.cpu().numpy()
    n = int(r.n_inl.item())
    return pose[:, :3].copy(), pose[:, 3].copy(), r.inl_idx[:n].cpu().numpy()


# ------------------------------------------------------------------------------ a8 / a9 metrics
def set_surface_points(surface_pts):
    """The reference's ADDS reads the module global surfacePointsScaled (inference.py:119,
    choosePose.py:21 — a NameError there unless --posesEst ran).  Set it explicitly here."""
    _state["surface_pts"] = _dev(surface_pts, torch.float32)


def ADD(verts, gtR1, gtT1, R1, T1):
    """inference.py:116-117: mean || (V gtR^T + gtT) - (V R^T + T) ||."""
    v = _dev(verts, torch.float32)
    Ta = _dev(_pose12(gtR1, gtT1)[None])
    Tb = _dev(_pose12(R1, T1)[None])
    return float(ops.add_metric(v, Ta, Tb).item())


def ADDS(verts, gtR1, gtT1, R1, T1, surface_pts=None):
    """inference.py:118-120 / choosePose.py:20-22:
    KDTree(surfacePointsScaled R^T + T, leaf_size=2).query(V gtR^T + gtT, k=1)[0].mean()
    — targets are the predicted-pose surface points, queries the GT-pose CAD vertices."""
    sp = _state["surface_pts"] if surface_pts is None else _dev(surface_pts, torch.float32)
    if sp is None:
        raise NameError("name 'surfacePointsScaled' is not defined (call set_surface_points first)")
    v = _dev(verts, torch.float32)
    r = ops.nn_batched(v, sp, _dev(_pose12(gtR1, gtT1)[None]), _dev(_pose12(R1, T1)[None]))
    return float(r.sum_d.item()) / v.shape[0]


# ------------------------------------------------------------------- a10 / a12 relative poses
def compute_rel_poses(R1, t1, R2, t2):
    """choosePose.py:43-51: (R1^T R2, t2 - t1) — host 3x3 arithmetic exactly as the reference."""
    R1, R2 = np.asarray(R1), np.asarray(R2)
    return np.dot(R1.T, R2), np.asarray(t2) - np.asarray(t1)


def calculate_relative_pose(R1, T1, R2, T2):
    """verfication.py:9-19: Rel = [R2|T2] inv([R1|T1]) -> (Rel[:3,:3], Rel[:3,3])."""
    RT1 = np.vstack([np.hstack((np.asarray(R1), np.asarray(T1).reshape(-1, 1))), [0, 0, 0, 1]])
    RT2 = np.vstack([np.hstack((np.asarray(R2), np.asarray(T2).reshape(-1, 1))), [0, 0, 0, 1]])
    Rel = np.dot(RT2, np.linalg.inv(RT1))
    return Rel[:3, :3], Rel[:3, -1]


def relative_pose_table(RList, TList, mode="choose", rows=None):
    """The n x n table of choosePose.py:98-107 (mode 'choose') or of calculate_relative_pose
    (mode 'verif'), built on the device: (rows, n, 4, 4) f64 NumPy, bottom row [0,0,0,1]."""
    R, t = _dev(np.asarray(RList, np.float64)), _dev(np.asarray(TList, np.float64))
    n = R.shape[0]
    i0, i1 = (0, n) if rows is None else rows
    tab = ops.rel_pose_table(R, t, 0 if mode == "choose" else 1, i0, i1).cpu().numpy()
    out = np.zeros((i1 - i0, n, 4, 4))
    out[:, :, :3, :] = tab
    out[:, :, 3, 3] = 1.0
    return out


# ------------------------------------------------------------------------- a11 best-image vote
def vote_error_rows(model_verts, surface_pts, gt_rel, pred_rel, diameter, chunk=4096):
    """choosePose.py:121-138 for a block of rows: error[i][j] = ADDS(modelVerts, gt_rel[i][j],
    pred_rel[i][j]) < 0.1 * diameter.  gt_rel / pred_rel (rows, n, 4, 4) or (rows, n, 3, 4).
    Returns (error (rows,n) f64 of 0/1, adds (rows,n) f64)."""
    v, sp = _dev(model_verts, torch.float32), _dev(surface_pts, torch.float32)
    g = np.asarray(gt_rel, np.float64)[..., :3, :].reshape(-1, 12)
    p = np.asarray(pred_rel, np.float64)[..., :3, :].reshape(-1, 12)
    rows, n = np.asarray(gt_rel).shape[:2]
    adds = np.empty(rows * n)
    for s in range(0, rows * n, chunk):
        r = ops.nn_batched(v, sp, _dev(g[s:s + chunk]), _dev(p[s:s + chunk]))
        adds[s:s + chunk] = (r.sum_d / v.shape[0]).cpu().numpy()
    adds = adds.reshape(rows, n)
    return (adds < 0.1 * diameter).astype(np.float64), adds


def choose_image(error, top=50):
    """choosePose.py:144-145: image_id = argmax(row sums), top-50 = argsort(-row sums)[:50].
    NumPy's default argsort leaves tie order unspecified; here ties keep the lower index."""
    s = np.sum(error, axis=1)
    return int(np.argmax(s)), np.argsort(-s, kind="stable")[:top]


# ----------------------------------------------------------------------------- a13 Chamfer
def chamfer(pc_a, pc_b, Ta=None, Tb=None):
    """verfication.py:97-101 / icp.py:113-117: (mean NN dist(a->b) + mean NN dist(b->a)) / 2.
    Ta / Tb: optional (3,4) rigid transforms applied on the fly."""
    a, b = _dev(pc_a, torch.float32), _dev(pc_b, torch.float32)
    ta = None if Ta is None else _dev(np.asarray(Ta, np.float64).reshape(1, 12))
    tb = None if Tb is None else _dev(np.asarray(Tb, np.float64).reshape(1, 12))
    ab = ops.nn_batched(a, b, ta, tb)
    ba = ops.nn_batched(b, a, tb, ta)
    return 0.5 * (float(ab.sum_d.item()) / a.shape[0] + float(ba.sum_d.item()) / b.shape[0])


def chamfer_pairs(pc1, R_pred, R_rel_gt):
    """The loop of verfication.py:61-102 for all consecutive pairs at once.
    pcgt = pc1 . R1pred^T . R_rel ; pcpred = pc1 . R2pred  (row-vector products as written there:
    rotation only, translations commented out at :83-85).  R_pred (n,3,3) predicted rotations,
    R_rel_gt (n-1,3,3) GT relative rotations.  Returns chamferdis (n-1,) f64 on the device."""
    pc = _dev(pc1, torch.float32)
    Rp = np.asarray(R_pred, np.float64)
    Rr = np.asarray(R_rel_gt, np.float64)
    n1 = Rp.shape[0] - 1
    Tg = np.zeros((n1, 3, 4))
    Tp = np.zeros((n1, 3, 4))
    # x -> (x^T R1^T R_rel)^T = R_rel^T R1 x ;  x -> (x^T R2)^T = R2^T x
    Tg[:, :, :3] = np.einsum("nji,njk->nik", Rr, Rp[:-1])
    Tp[:, :, :3] = np.transpose(Rp[1:], (0, 2, 1))
    tg, tp = _dev(Tg.reshape(n1, 12)), _dev(Tp.reshape(n1, 12))
    ab = ops.nn_batched(pc, pc, tp, tg)     # pcpred -> pcgt
    ba = ops.nn_batched(pc, pc, tg, tp)     # pcgt -> pcpred
    return 0.5 * (ab.sum_d + ba.sum_d) / pc.shape[0]


def choose_best(chamferdis):
    """verfication.py:105-106: min and list.index(min) — the FIRST minimum."""
    c = chamferdis.cpu().numpy() if isinstance(chamferdis, torch.Tensor) else np.asarray(chamferdis)
    i = int(np.argmin(c))
    return i, float(c[i])


# ------------------------------------------------------------------------------ a14 / a15 ICP
def _eval(src, tgt, threshold, T):
    """One NN(radius) pass of source under T against target; ONE device->host copy brings back
    {sum d, sum d^2} and the 16 Kabsch sums (slot 15 = inlier count)."""
    dev = src.device
    Tq = torch.from_numpy(np.ascontiguousarray(np.asarray(T, np.float64)[:3, :].reshape(1, 12))).to(dev)
    buf = torch.empty(18, dtype=torch.float64, device=dev)       # [sum_d, sum_d2, cov(16)]
    Nq, Nt = src.shape[0], tgt.shape[0]
    L = ops.lib()
    ws = ops.workspace(dev, L.isr_nn_batched_workspace_bytes(Nq, Nt, 1), "nn")
    base = buf.data_ptr()
    with torch.cuda.device(dev):
        rc = L.isr_nn_batched(ops.ptr(src), Nq, ops.ptr(tgt), Nt, ops.ptr(Tq), None, 1, float(threshold), base, base + 8,
                              None, None, None, base + 16, ops.ptr(ws), ws.numel(), ops.current_stream(dev))
    ops.check(rc, "isr_nn_batched")
    h = buf.cpu().numpy()
    cov = h[2:]
    n = int(cov[15])
    fitness = n / Nq
    rmse = float(np.sqrt(h[1] / n)) if n else 0.0
    return fitness, rmse, n, cov


def evaluate_registration(source, target, threshold, init=None):
    """icp.py:97-99 open3d evaluate_registration -> (fitness, inlier_rmse):
    fitness = #source points with a target within `threshold` / |source|, rmse over those."""
    src, tgt = _dev(source, torch.float32), _dev(target, torch.float32)
    T = np.eye(4) if init is None else np.asarray(init, np.float64)
    f, r, _, _ = _eval(src, tgt, threshold, T)
    return f, r


MORTON_MIN_ROWS = 256     # icp_point_to_point sorts its clouds only above this many rows (one target tile of the searches)


def morton_order(points: torch.Tensor) -> torch.Tensor:
    """Row permutation that walks a cloud along a Morton (Z-order) curve: 10 bits per axis on a cubic lattice over the
    bounding box, stable on equal codes.  Consecutive rows of the permuted cloud are a compact patch — what the per-wave
    tile cull of the ICP searches (csrc/nn_batched.hip) needs; nothing else depends on the order."""
    p = points.to(torch.float32)
    if p.shape[0] <= 1:                                  # nothing to order (and min / max of an empty cloud raise)
        return torch.arange(p.shape[0], device=p.device)
    lo = p.min(0).values
    ext = (p.max(0).values - lo).max().clamp_min(1e-30)
    q = ((p - lo) * (1023.0 / ext)).to(torch.int64).clamp_(0, 1023)

    def spread(x):
        x = (x | (x << 16)) & 0x30000FF
        x = (x | (x << 8)) & 0x300F00F
        x = (x | (x << 4)) & 0x30C30C3
        return (x | (x << 2)) & 0x9249249

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code, stable=True)


def icp_point_to_point(source, target, threshold, init=None, max_iter=30, rel_fitness=1e-6,
                       rel_rmse=1e-6, spatial_order=True):
    """icp.py:101-103 open3d registration_icp(source, target, threshold, init,
    TransformationEstimationPointToPoint()) with the library defaults (30 iterations, relative
    fitness / rmse 1e-6).  The loop — NN(radius) + Kabsch sums, stopping rule, rigid update — runs
    on the device without host round trips (isr_icp_point_to_point); one copy brings back T and
    the final (fitness, inlier_rmse).  Returns (T (4,4) f64, fitness, inlier_rmse).
    spatial_order: both clouds are handed over in Morton order (morton_order).  Open3D's result does not depend on the
    order of the points and neither does this one beyond the last bits of the f64 sums; the searches skip, wave by
    wave, the target tiles that lie beyond the wave's bound (and beyond the radius), which needs rows that are
    neighbours in space."""
    src, tgt = _dev(source, torch.float32).contiguous(), _dev(target, torch.float32).contiguous()
    # the cull works on 256-row target tiles: a cloud of one tile has nothing to skip, and an empty one goes to the C entry
    # as it is (which reports the argument error)
    if spatial_order and min(src.shape[0], tgt.shape[0]) > MORTON_MIN_ROWS:
        src, tgt = src[morton_order(src)].contiguous(), tgt[morton_order(tgt)].contiguous()
    dev = src.device
    T0 = np.eye(4) if init is None else np.asarray(init, np.float64)
    buf = torch.empty(20, dtype=torch.float64, device=dev)        # T (16) | result (4)
    buf[:16] = torch.from_numpy(np.ascontiguousarray(T0.reshape(16))).to(dev)
    L = ops.lib()
    ws = ops.workspace(dev, L.isr_icp_workspace_bytes(src.shape[0], tgt.shape[0]), "icp")
    base = buf.data_ptr()
    with torch.cuda.device(dev):
        rc = L.isr_icp_point_to_point(ops.ptr(src), src.shape[0], ops.ptr(tgt), tgt.shape[0], float(threshold),
                                      int(max_iter), float(rel_fitness), float(rel_rmse), base, base + 128,
                                      ops.ptr(ws), ws.numel(), ops.current_stream(dev))
    ops.check(rc, "isr_icp_point_to_point")
    h = buf.cpu().numpy()
    return h[:16].reshape(4, 4).copy(), float(h[16]), float(h[17])


def final_chamfer(source, target, T, cad_points):
    """icp.py:110-117: pred_obj_full = transform(source) U target; Chamfer against the CAD cloud."""
    src, tgt, cad = _dev(source, torch.float32), _dev(target, torch.float32), _dev(cad_points, torch.float32)
    Tm = _dev(np.asarray(T, np.float64)[:3, :].reshape(1, 12))
    # merged cloud -> CAD: two query sets against the same targets
    a = ops.nn_batched(src, cad, Tm, None)
    b = ops.nn_batched(tgt, cad, None, None)
    mean_ab = (float(a.sum_d.item()) + float(b.sum_d.item())) / (src.shape[0] + tgt.shape[0])
    # CAD -> merged cloud: nearest over both parts = min of the two nearest distances
    c = ops.nn_batched(cad, src, None, Tm, want_dist=True)
    d = ops.nn_batched(cad, tgt, None, None, want_dist=True)
    mean_ba = float(torch.minimum(c.nn_d, d.nn_d).mean().item())
    return 0.5 * (mean_ab + mean_ba)
