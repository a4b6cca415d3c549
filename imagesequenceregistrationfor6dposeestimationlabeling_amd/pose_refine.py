"""refine_pose() with the reference's signature (pose_refine.py:21-104).

The caller-supplied objects are used exactly as the reference uses them: `renderer.render(obj_idx,
K_crop, R, t[:,None])` for the visible object coordinates, `neural_radiance_field.batched_customForward`
for their key descriptors, `obj_.scale / .offset / .diameter`.  What runs in HIP kernels: the
log-sum-exp denominator image (K1's `lse` output over the sampled keys — no (H*W x 10 960) matrix)
and the bilinear objective with its analytic translation gradient (isr_refine_objective) — the
reference builds both with torch autograd and a cv2.Rodrigues round trip per evaluation.  BFGS stays
scipy.optimize.minimize on the host, as in the reference.  As there, only the translation is
optimised (the objective's rotation is a constant, pose_refine.py:73-76): R is returned unchanged.
`optimize_rotation=True` (off by default) is the evidently intended variant: the 6-vector is
(rotation vector, translation), the device returns d score / d R as well and the host chains it with the
Rodrigues Jacobian.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
from scipy.optimize import minimize

from . import ops
from ._capi import check, current_stream, lib, ptr, require_cuda
from .registration import _dev


INTERPOLATION = {"bilinear": 0, "nearest": 1, "bicubic": 2}     # F.grid_sample's modes -> ISR_INTERP_*


class RefineObjective:
    """score(t), grad(t) of pose_refine.py:70-91 for a fixed rotation, evaluated on the device."""

    def __init__(self, coord_obj: torch.Tensor, keys_masked: torch.Tensor, query_img: torch.Tensor,
                 denom_img: torch.Tensor, K_crop, R, interpolation: str = "bilinear"):
        if interpolation not in INTERPOLATION:
            raise ValueError(f"interpolation={interpolation!r}: F.grid_sample knows {sorted(INTERPOLATION)}")
        self.mode = INTERPOLATION[interpolation]
        self.dev = require_cuda(coord_obj, keys_masked, query_img, denom_img)
        self.X = coord_obj.to(torch.float32).contiguous()
        self.keys = keys_masked.to(torch.float32).contiguous()
        self.q = query_img.to(torch.float32).contiguous()
        self.den = denom_img.to(torch.float32).reshape(query_img.shape[0], query_img.shape[1]).contiguous()
        self.K = (ctypes.c_double * 9)(*np.asarray(K_crop, np.float64).reshape(9).tolist())
        self.R = np.asarray(R, np.float64).reshape(3, 3)
        self.out = torch.empty(13, dtype=torch.float64, device=self.dev)
        self.out_host = torch.empty(13, dtype=torch.float64).pin_memory()
        self.ws = ops.workspace(self.dev, 1 << 16, "refine_obj")
        self.n_launch = 0                 # device evaluations so far
        self._last = None                 # (pose bytes, full) -> outputs of the last evaluation

    def _eval(self, t, R=None, full=False):
        """One launch per distinct pose: BFGS asks for the value and then for the gradient at the same point
        (`fun` and `jac` are separate callables in the reference, pose_refine.py:93-101) — the kernel returns both."""
        Rm = self.R if R is None else np.asarray(R, np.float64).reshape(3, 3)
        Rt = np.concatenate([Rm, np.asarray(t, np.float64).reshape(3, 1)], axis=1).reshape(12)
        tag = (Rt.tobytes(), bool(full))
        if self._last is not None and self._last[0] == tag:
            return self._last[1]
        rt = (ctypes.c_double * 12)(*Rt.tolist())
        N, e = self.keys.shape
        fn = lib().isr_refine_objective_full if full else lib().isr_refine_objective
        with torch.cuda.device(self.dev):
            rc = fn(ptr(self.X), ptr(self.keys), N, e, ptr(self.q), ptr(self.den), self.q.shape[0], self.mode,
                    ctypes.cast(self.K, ctypes.c_void_p), ctypes.cast(rt, ctypes.c_void_p), ptr(self.out), ptr(self.ws),
                    self.ws.numel(), current_stream(self.dev))
            check(rc, "isr_refine_objective")
            self.out_host.copy_(self.out, non_blocking=True)
            torch.cuda.current_stream(self.dev).synchronize()
        self.n_launch += 1
        res = self.out_host.numpy()[: 13 if full else 4].copy()
        self._last = (tag, res)
        return res

    def with_rotation(self, pose, return_grad=False):
        """The 6-vector is (rotation vector, t): score, or its gradient (d/d rvec by the Rodrigues Jacobian)."""
        pose = np.asarray(pose, np.float64)
        R, dR = rodrigues(pose[:3])
        o = self._eval(pose[3:], R, full=True)
        if return_grad:
            return np.concatenate([np.einsum("jk,ijk->i", o[4:].reshape(3, 3), dR), o[1:4]])
        return float(o[0])

    def __call__(self, pose, return_grad=False):
        """pose: the reference's 6-vector (rvec ignored, t = pose[3:])."""
        o = self._eval(np.asarray(pose, np.float64)[3:])
        if return_grad:
            return np.concatenate([np.zeros(3), o[1:]])       # autograd leaves the unused rvec slots at 0
        return float(o[0])


def rodrigues(rvec):
    """cv2.Rodrigues(rvec) -> (R (3,3), dR (3,3,3) with dR[i] = d R / d rvec_i) in f64 (Gallego & Yezzi 2015:
    dR/dr_i = (r_i [r]x + [r x (I - R) e_i]x) R / |r|^2; [e_i]x at r = 0)."""
    r = np.asarray(rvec, np.float64).reshape(3)

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    th2 = float(r @ r)
    if th2 < 1e-24:
        return np.eye(3) + skew(r), np.stack([skew(e) for e in np.eye(3)])
    th = np.sqrt(th2)
    Kx = skew(r / th)
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
    dR = np.stack([(r[i] * skew(r) + skew(np.cross(r, (np.eye(3) - R) @ np.eye(3)[i]))) @ R / th2 for i in range(3)])
    return R, dR


def denominator_image(query_img: torch.Tensor, keys_sampled: torch.Tensor) -> torch.Tensor:
    """pose_refine.py:56: logsumexp(query_img @ keys_sampled.T, -1) -> (H, W, 1): an lse-only call of K1."""
    H, W, e = query_img.shape
    lse = ops.corr_lse(query_img.reshape(H * W, e).to(torch.float32), keys_sampled.to(torch.float32))
    return lse.reshape(H, W, 1)


def refine_pose(R, t, query_img, renderer, obj_idx, K_crop, obj_, neural_radiance_field, keys_verts,
                interpolation='bilinear', n_samples_denom=10960, method='BFGS', *, generator=None,
                optimize_rotation=False):
    """pose_refine.py:21-104.  Returns (R, t (3,), result.fun).  optimize_rotation=True also refines R (returned
    as Rodrigues(result.x[:3])); the default keeps the reference's behaviour (R constant, returned unchanged)."""
    if interpolation not in INTERPOLATION:
        raise ValueError(f"interpolation={interpolation!r}: F.grid_sample knows {sorted(INTERPOLATION)}")
    query_img = _dev(query_img, torch.float32)
    h, w, _ = query_img.shape
    assert h == w
    dev = query_img.device
    t = np.asarray(t, np.float64).reshape(3)
    coord_img = renderer.render(obj_idx, K_crop, R, np.expand_dims(t, axis=1))
    coord_img = coord_img.cpu().numpy() if isinstance(coord_img, torch.Tensor) else np.asarray(coord_img)
    mask = coord_img[..., 3] == 1.
    coord_norm_masked = torch.from_numpy(np.ascontiguousarray(coord_img[..., :3][mask])).to(dev)
    coord_masked = coord_norm_masked * obj_.scale + torch.from_numpy(np.asarray(obj_.offset)).to(dev)
    coord_nerf = torch.from_numpy((coord_masked.cpu().numpy() * 1.8 / obj_.diameter).astype("float32")).to(dev)
    feat = neural_radiance_field.batched_customForward(coord_nerf).detach().clone()
    keys_masked = feat[..., :feat.shape[-1] - 1]                                   # drop the silhouette channel
    keys_verts = _dev(keys_verts, torch.float32)
    perm = torch.randperm(len(keys_verts), device=dev, generator=generator)[:n_samples_denom]
    denom_img = denominator_image(query_img, keys_verts[perm])
    obj = RefineObjective(coord_masked.float(), keys_masked.float(), query_img, denom_img, K_crop, R, interpolation)
    if optimize_rotation:
        from scipy.spatial.transform import Rotation
        rvec = Rotation.from_matrix(np.asarray(R, np.float64)).as_rotvec()
        pose = np.array([rvec[0], rvec[1], rvec[2], t[0], t[1], t[2]], dtype=np.float64)
        result = minimize(fun=obj.with_rotation, x0=pose, jac=lambda p: obj.with_rotation(p, return_grad=True), method=method)
        return rodrigues(result.x[:3])[0], result.x[3:], result.fun
    pose = np.array([0, 0, 0, t[0], t[1], t[2]], dtype=np.float64)
    result = minimize(fun=obj, x0=pose, jac=lambda p: obj(p, return_grad=True), method=method)
    return R, result.x[3:], result.fun
