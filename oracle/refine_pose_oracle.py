"""TEST INFRASTRUCTURE, NOT PRODUCT — the objective of refine_pose (pose_refine.py:58-91) as the
reference writes it: torch F.grid_sample + autograd on the CPU (pinned: these are the reference's
own torch calls; cv2.Rodrigues is replaced by passing R directly — the reference feeds it a constant
rvec, so only t is differentiated there too)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def sample(img, p_img_norm, interpolation="bilinear"):
    samples = F.grid_sample(img.permute(2, 0, 1)[None], p_img_norm[None, None], align_corners=False,
                            padding_mode="border", mode=interpolation)
    return samples[0, :, 0].T


def objective(t, R, coord_masked, keys_masked, query_img, denom_img, K_crop, return_grad=False, interpolation="bilinear",
              dtype=torch.float32):
    """pose_refine.py:70-91 with pose[3:] = t.  All tensors torch-CPU, f32 as in the reference; dtype=torch.float64
    evaluates the same statements in double (the f32 autograd gradient of the bicubic mode carries rounding noise from
    points that project far outside the image: their coefficient derivatives cancel only to ~1e-7 and are multiplied
    by (c_x - u) / z, which is huge there)."""
    res = query_img.shape[0]
    coord_masked, keys_masked, query_img, denom_img = (x.to(dtype) for x in (coord_masked, keys_masked, query_img, denom_img))
    tt = torch.tensor(np.asarray(t, np.float64), dtype=dtype, requires_grad=return_grad)
    Rt = torch.cat((torch.from_numpy(np.asarray(R, np.float64)).to(dtype), tt[:, None]), dim=1)
    P = torch.from_numpy(np.asarray(K_crop, np.float64)).to(dtype) @ Rt
    X = torch.cat((coord_masked, torch.ones(len(coord_masked), 1, dtype=dtype)), dim=1)
    p_img = X @ P.T
    p_img = p_img[..., :2] / p_img[..., 2:]
    p_norm = (p_img + 0.5) * (2 / res) - 1
    q = sample(query_img, p_norm, interpolation)
    log_nom = (keys_masked * q).sum(dim=-1)
    log_den = sample(denom_img, p_norm, interpolation)[:, 0]
    score = -(log_nom.mean() - log_den.mean()) / 2
    if return_grad:
        score.backward()
        g = tt.grad if tt.grad is not None else torch.zeros(3)
        return score.item(), g.detach().numpy().astype(np.float64)
    return score.item()


def rodrigues_torch(rvec: torch.Tensor) -> torch.Tensor:
    """Rotation matrix of a rotation vector, differentiable (what cv2.Rodrigues computes; the reference wraps it in
    an autograd Function, pose_refine.py:7-18)."""
    th = torch.sqrt((rvec * rvec).sum())
    k = rvec / th
    Kx = torch.stack([torch.stack([torch.zeros(()), -k[2], k[1]]), torch.stack([k[2], torch.zeros(()), -k[0]]),
                      torch.stack([-k[1], k[0], torch.zeros(())])]).to(rvec.dtype)
    return torch.eye(3, dtype=rvec.dtype) + torch.sin(th) * Kx + (1 - torch.cos(th)) * (Kx @ Kx)


def objective_with_rotation(pose6, coord_masked, keys_masked, query_img, denom_img, K_crop):
    """The objective of pose_refine.py:70-91 with Rt = [Rodrigues(pose[:3]) | pose[3:]] (the evidently intended
    variant: the reference builds R from a constant), value and 6-gradient by torch autograd in f64."""
    res = query_img.shape[0]
    pose = torch.tensor(np.asarray(pose6, np.float64), dtype=torch.float64, requires_grad=True)
    Rt = torch.cat((rodrigues_torch(pose[:3]), pose[3:, None]), dim=1)
    P = torch.from_numpy(np.asarray(K_crop, np.float64)) @ Rt
    X = torch.cat((coord_masked.double(), torch.ones(len(coord_masked), 1, dtype=torch.float64)), dim=1)
    p_img = X @ P.T
    p_img = p_img[..., :2] / p_img[..., 2:]
    p_norm = (p_img + 0.5) * (2 / res) - 1
    q = sample(query_img.double(), p_norm)
    score = -((keys_masked.double() * q).sum(dim=-1).mean() - sample(denom_img.double(), p_norm)[:, 0].mean()) / 2
    score.backward()
    return score.item(), pose.grad.numpy().copy()


def denominator_image(query_img, keys_sampled):
    """pose_refine.py:56."""
    return torch.logsumexp(query_img @ keys_sampled.T, dim=-1, keepdim=True)
