"""Synthetic stand-ins for the BOP inputs: no dataset, checkpoint or `dep/`
package ships with the reference, so tests and bench.py drive the hot path from these generators.
NumPy only, seeded; nothing here is on the measured path."""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def _unit_sphere(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def bumpy_ellipsoid(rng, n, radii=(60.0, 40.0, 25.0)):
    """Asymmetric bumpy ellipsoid, r * (1 + 0.15 sin(3 theta) cos(2 phi)) — config 1 (Ruapc-like)."""
    d = _unit_sphere(rng, n)
    theta = np.arccos(np.clip(d[:, 2], -1, 1))
    phi = np.arctan2(d[:, 1], d[:, 0])
    s = 1.0 + 0.15 * np.sin(3 * theta) * np.cos(2 * phi)
    return (d * np.asarray(radii) * s[:, None]).astype(np.float32)


def tless_like(rng, n):
    """Box 60x40x30 with a cylinder boss (r=12, h=15) on top: discrete-symmetric solid — config 2."""
    hx, hy, hz = 30.0, 20.0, 15.0
    areas = np.array([4 * hy * hz, 4 * hy * hz, 4 * hx * hz, 4 * hx * hz, 4 * hx * hy, 4 * hx * hy,
                      2 * np.pi * 12 * 15, np.pi * 144])
    face = rng.choice(len(areas), size=n, p=areas / areas.sum())
    u, v = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    p = np.zeros((n, 3))
    for f, (ax, sg) in enumerate([(0, 1), (0, -1), (1, 1), (1, -1), (2, 1), (2, -1)]):
        m = face == f
        h = [hx, hy, hz]
        o = [a for a in range(3) if a != ax]
        p[m, ax] = sg * h[ax]
        p[m, o[0]] = u[m] * h[o[0]]
        p[m, o[1]] = v[m] * h[o[1]]
    m = face == 6
    ang = np.pi * u[m]
    p[m] = np.stack([12 * np.cos(ang), 12 * np.sin(ang), hz + 7.5 * (v[m] + 1)], 1)
    m = face == 7
    rr = 12 * np.sqrt(0.5 * (u[m] + 1))
    ang = np.pi * v[m]
    p[m] = np.stack([rr * np.cos(ang), rr * np.sin(ang), np.full(m.sum(), hz + 15.0)], 1)
    return p.astype(np.float32)


def revolution(rng, n):
    """Cylinder r=30, h=60 with a cone cap: continuous-symmetry object — config 4."""
    part = rng.uniform(size=n) < 0.7
    ang = rng.uniform(0, 2 * np.pi, n)
    z = np.where(part, rng.uniform(-30, 30, n), 0.0)
    r = np.full(n, 30.0)
    tcone = rng.uniform(0, 1, n)
    z = np.where(part, z, 30 + 25 * tcone)
    r = np.where(part, r, 30 * (1 - tcone))
    return np.stack([r * np.cos(ang), r * np.sin(ang), z], 1).astype(np.float32)


def revolution_keys(rng, pts, D=64, tau=5.0, azimuth=0.25, noise=1e-3):
    """Descriptors for the continuous-symmetry object of config 4: random Fourier features of the PROFILE
    coordinates (height z, radius rho) — constant along the azimuth — plus a weak azimuth-dependent part
    (amplitude `azimuth`) and a tiny noise, normalised to |k| = tau.  Keys of points on the same parallel
    are nearly identical: top-2 margins are small by construction (the exact-recheck path gets work) and a
    correspondence's azimuth is only weakly determined (RANSAC sees a low inlier ratio)."""
    z, rho = pts[:, 2] / 30.0, np.hypot(pts[:, 0], pts[:, 1]) / 30.0
    phi = np.arctan2(pts[:, 1], pts[:, 0])
    na = 8
    W = rng.normal(0, 2.5, size=(2, D - na))
    b = rng.uniform(0, 2 * np.pi, D - na)
    prof = np.cos(np.stack([z, rho], 1) @ W + b)
    az = np.concatenate([np.cos(np.outer(phi, np.arange(1, na // 2 + 1))), np.sin(np.outer(phi, np.arange(1, na // 2 + 1)))], 1)
    k = np.concatenate([prof, azimuth * np.sqrt((D - na) / na) * az], 1) + noise * rng.normal(size=(len(pts), D))
    return (tau * k / np.linalg.norm(k, axis=1, keepdims=True)).astype(np.float32)


def diameter(pts, sample=2000, rng=None):
    rng = rng or np.random.default_rng(0)
    s = pts[rng.choice(len(pts), min(sample, len(pts)), replace=False)].astype(np.float64)
    d = np.linalg.norm(s[:, None] - s[None], axis=-1)
    return float(d.max())


def camera(width=640, height=480, f=None):
    f = f or 1.2 * width
    return np.array([[f, 0, (width - 1) / 2.0], [0, f, (height - 1) / 2.0], [0, 0, 1.0]])


def random_poses(rng, n, tz=700.0, t_sigma=20.0):
    R = Rotation.random(n, random_state=int(rng.integers(1 << 31))).as_matrix()
    t = np.array([0.0, 0.0, tz]) + rng.normal(0, t_sigma, (n, 3))
    return R, t


def perturb_pose(rng, R, t, rot_deg, trans):
    ax = _unit_sphere(rng, 1)[0]
    dR = Rotation.from_rotvec(ax * np.deg2rad(rot_deg) * rng.uniform(0.3, 1.0)).as_matrix()
    return dR @ R, t + rng.uniform(-trans, trans, 3)


def project(K, R, t, X):
    Xc = X.astype(np.float64) @ R.T + t
    p = Xc @ K.T
    return p[:, :2] / p[:, 2:3]


def unit_keys(rng, N, D, tau=8.0):
    """Key descriptors with |k| = tau (so a planted query's own key wins the dot product)."""
    K = rng.normal(size=(N, D))
    return (tau * K / np.linalg.norm(K, axis=1, keepdims=True)).astype(np.float32)


def image_case(rng, keys, pts, Kcam, R, t, P, sigma=0.35, noise_px=0.5, outlier_frac=0.3):
    """One second-sequence image: P query descriptors + their pixel coordinates.
    gt_geo[p] is the key whose 3-D point really projects to pixel p; a fraction `outlier_frac` of
    the descriptors is planted on a different, uniformly drawn key (wrong 2D-3D match)."""
    N = len(keys)
    gt_geo = rng.integers(N, size=P)
    gt_match = gt_geo.copy()
    out = rng.uniform(size=P) < outlier_frac
    gt_match[out] = rng.integers(N, size=int(out.sum()))
    Q = keys[gt_match] + sigma * rng.normal(size=(P, keys.shape[1])).astype(np.float32)
    pix = project(Kcam, R, t, pts[gt_geo]) + noise_px * rng.normal(size=(P, 2))
    return Q.astype(np.float32), pix.astype(np.float32), gt_match, gt_geo


def pnp_case(rng, pts, Kcam, R, t, M, noise_px=0.5, outlier_frac=0.3):
    """Correspondences for the RANSAC stage alone: p3d (M,3) f32, p2d (M,2) f32, inlier flags."""
    idx = rng.integers(len(pts), size=M)
    p2d = project(Kcam, R, t, pts[idx]) + noise_px * rng.normal(size=(M, 2))
    out = rng.uniform(size=M) < outlier_frac
    idx3 = idx.copy()
    idx3[out] = rng.integers(len(pts), size=int(out.sum()))
    return pts[idx3].astype(np.float32), p2d.astype(np.float32), ~out


def split_halves(rng, cloud, n_each, overlap=5.0):
    """Upper / lower halves of an object with an overlap band (two NeRF reconstructions)."""
    up = cloud[cloud[:, 2] > -overlap]
    lo = cloud[cloud[:, 2] < overlap]
    up = up[rng.choice(len(up), n_each, replace=len(up) < n_each)]
    lo = lo[rng.choice(len(lo), n_each, replace=len(lo) < n_each)]
    return up.astype(np.float32), lo.astype(np.float32)


def rot_angle(Ra, Rb):
    """Rotation angle between two rotations, stable near 0: |Ra - Rb|_F = 2 sqrt(2) sin(angle / 2)."""
    f = np.linalg.norm(np.asarray(Ra, np.float64) - np.asarray(Rb, np.float64))
    return float(2.0 * np.arcsin(min(1.0, f / (2.0 * np.sqrt(2.0)))))


def crop_scene(seed=7, r=224, e=12, m=80000, f=700.0):
    """estimate_pose's inputs at the reference's own size (poseEstSurf.py:11-15 as called from inference.py: a 224 x 224
    crop, 12-D descriptors, m = 80 000 surface points, genFeat.py:201): mask logits of a rendered bumpy ellipsoid, a query
    image whose object pixels carry the (noisy) keys of the surface points that project there."""
    rng = np.random.default_rng(seed)
    pts = bumpy_ellipsoid(rng, m)
    nrm = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    keys = unit_keys(rng, m, e, tau=6.0)
    R, t = random_poses(rng, 1, tz=420.0, t_sigma=5.0)
    R, t = R[0], t[0]
    K = np.array([[f, 0, r / 2 - 0.5], [0, f, r / 2 - 0.5], [0, 0, 1]])
    uv = project(K, R, t, pts)
    cam = pts.astype(np.float64) @ R.T + t
    vis = (nrm @ R.T * cam).sum(1) < 0
    ui, vi = np.rint(uv[:, 0]).astype(int), np.rint(uv[:, 1]).astype(int)
    ok = np.nonzero(vis & (ui >= 0) & (ui < r) & (vi >= 0) & (vi < r))[0]
    ok = ok[np.argsort(-cam[ok, 2])]                                 # nearest written last
    mask_lgts = np.full((r, r), -6.0, np.float32)
    query = (0.3 * rng.normal(size=(r, r, e))).astype(np.float32)
    mask_lgts[vi[ok], ui[ok]] = 6.0
    query[vi[ok], ui[ok]] = keys[ok] + 0.2 * rng.normal(size=(len(ok), e)).astype(np.float32)
    return dict(pts=pts, normals=nrm, keys=keys, R=R, t=t, K=K, mask_lgts=mask_lgts, query=query,
                diameter=diameter(pts), r=r, e=e, m=m)


def crop_batch(dev, n=128, seed=0, N=80000, D=12, H=224, ds=3, f=600.0):
    """n crops at the reference's own per-image shape (inference.py:163, 248-293; genFeat.py:201): network outputs
    (n, H, H, D + 1) f32 on the device whose every ds-th pixel inside the object mask carries the (noisy, 25 % wrong) key of the
    surface point that projects there, masks (n, H, H, 3) u8, the crop camera, the true poses.  Returns a dict; tensors on dev."""
    import torch
    rng = np.random.default_rng(seed)
    S1 = (H + ds - 1) // ds
    pts = tless_like(rng, N)
    keys = unit_keys(rng, N, D, tau=6.0)
    Kc = camera(S1, S1, f=f)          # the crop is cut to the object box x 1.2 (inference.py:203-206): the object fills it
    R, t = random_poses(rng, n, t_sigma=3.0)
    pts_d, keys_d = torch.from_numpy(pts).to(dev), torch.from_numpy(keys).to(dev)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    feats = torch.empty((n, H, H, D + 1), dtype=torch.float32, device=dev)
    masks = torch.zeros((n, H, H, 3), dtype=torch.uint8, device=dev)
    counts = []
    Kd = torch.from_numpy(Kc).to(dev)
    for i in range(n):
        Rt = torch.from_numpy(np.concatenate([R[i], t[i][:, None]], 1)).to(dev)
        Xc = pts_d.double() @ Rt[:, :3].T + Rt[:, 3]
        p = Xc @ Kd.T
        uv = p[:, :2] / p[:, 2:3]
        px = torch.round(uv).long()
        ok = (px[:, 0] >= 0) & (px[:, 0] < S1) & (px[:, 1] >= 0) & (px[:, 1] < S1) & ((uv - px).abs().max(1).values < 0.45)
        owner = torch.full((S1 * S1,), -1, dtype=torch.long, device=dev)
        sel = torch.nonzero(ok)[:, 0]
        sel = sel[torch.randperm(len(sel), device=dev, generator=g)]
        owner[px[sel, 1] * S1 + px[sel, 0]] = sel                     # one surface point per lattice pixel
        hit = torch.nonzero(owner >= 0)[:, 0]
        rows, cols = hit // S1, hit % S1
        src = owner[hit]
        wrong = torch.rand(len(src), device=dev, generator=g) < 0.25
        src_f = torch.where(wrong, torch.randint(N, (len(src),), device=dev, generator=g), src)
        feats[i] = 0.3 * torch.randn(H, H, D + 1, device=dev, generator=g)
        feats[i, rows * ds, cols * ds, :D] = keys_d[src_f] + 0.2 * torch.randn(len(src), D, device=dev, generator=g)
        masks[i, rows * ds, cols * ds] = 255
        counts.append(len(src))
    return dict(feats=feats, masks=masks, Kc=Kc, R=R, t=t, pts=pts_d, keys=keys_d, counts=counts, S1=S1, D=D, ds=ds)

