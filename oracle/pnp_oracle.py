"""TEST INFRASTRUCTURE, NOT PRODUCT — NumPy restatement of the PnP + RANSAC stage.

Reference call site: pnp() = cv2.solvePnPRansac(h3d, h2d, cam, None, iterationsCount=itr,
reprojectionError=reperr, flags=cv2.SOLVEPNP_P3P) + cv2.Rodrigues, inference.py:123-134 / :293.

PARITY UNPINNED: OpenCV (opencv-python, unpinned in requirements.txt:1-15) is not installed in
this image and the reference holds no test vector for this call, so neither OpenCV's sampler nor
its refit can be observed.  This module states the algorithm the build owns (DESIGN.md "K2"):

  * hypothesis h draws 4 indices i_j = floor(x_j * M / 2^32) from Philox4x32-10(key=seed,
    counter=(h,0,0,0)) — the published Random123 generator (Salmon et al., SC'11), pinned below
    by its known-answer vectors;
  * P3P on the first three points.  This oracle uses Grunert's formulation (depth ratios u, v;
    resultant quartic in v solved with numpy.roots) and a Kabsch alignment — deliberately a
    DIFFERENT solver from the device kernel (degenerate-conic / Lambda-twist family), so the two
    check each other; solutions agree to ~1e-9;
  * the root with the smallest reprojection error on the 4th point wins (z > 0 required);
  * inliers: z > 0 and |proj - obs|^2 <= reperr^2, evaluated division-free in f32
    (oracle/isr_oracle.c:orc_ransac_score); best = most inliers, lowest h on ties;
  * adaptive termination = cv2.solvePnPRansac's `confidence` (default 0.99, which the reference's
    call leaves in force) [rule from the RANSAC literature / OpenCV docs, from memory]: hypotheses are
    scored in stages [0,32), [32,96), [96,224), ...; after b hypotheses with best count c of M the
    loop stops when (1 - (c/M)^4)^b <= 1 - confidence (stop_rule below: multiplications only);
  * refit: Gauss-Newton on the reprojection error over the best hypothesis' inliers (f64), then one
    round of local optimisation: inliers of the refitted pose (same f32 test), refit on those; the
    reported inlier set is that of the refitted pose.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

from . import cbind

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter (..., 4) uint32, key (2,) uint32 -> (..., 4) uint32.  Random123 philox4x32-10."""
    c = np.array(counter, dtype=np.uint64) & _MASK
    c = c.reshape(-1, 4).copy()
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c = np.stack([n0, n1, n2, n3], axis=1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32).reshape(np.shape(counter))


def sample_indices(H: int, M: int, seed: int) -> np.ndarray:
    """(H, 4) int32 correspondence indices of every hypothesis."""
    ctr = np.zeros((H, 4), np.uint32)
    ctr[:, 0] = np.arange(H, dtype=np.uint32)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.uint64)
    return ((r * np.uint64(M)) >> np.uint64(32)).astype(np.int32)


def _kabsch(X, Y):
    """R, t with Y ~ R X + t for 3 (or more) points, proper rotation."""
    cx, cy = X.mean(0), Y.mean(0)
    Hm = (X - cx).T @ (Y - cy)
    U, _, Vt = np.linalg.svd(Hm)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1.0, 1.0, d]) @ U.T
    return R, cy - R @ cx


def p3p_grunert(X, uv, K):
    """All P3P solutions for 3 object points X (3,3) seen at pixels uv (3,2): list of (R, t)."""
    Ki = np.linalg.inv(K)
    y = (Ki @ np.concatenate([uv, np.ones((3, 1))], axis=1).T).T
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    a12 = np.sum((X[0] - X[1]) ** 2)
    a13 = np.sum((X[0] - X[2]) ** 2)
    a23 = np.sum((X[1] - X[2]) ** 2)
    if min(a12, a13, a23) <= 0:
        return []
    b12, b13, b23 = y[0] @ y[1], y[0] @ y[2], y[1] @ y[2]
    # lam2 = u lam1, lam3 = v lam1.  Two quadratics in u with polynomial-in-v coefficients
    # (numpy poly1d convention: highest power first):
    #   A2 u^2 + A1 u + A0(v) = 0   from a13 (1+u^2-2 b12 u) = a12 (1+v^2-2 b13 v)
    #   B2 u^2 + B1(v) u + B0(v) = 0 from a23 (1+u^2-2 b12 u) = a12 (u^2+v^2-2 b23 u v)
    P = np.poly1d
    A2, A1 = P([a13]), P([-2 * a13 * b12])
    A0 = P([-a12, 2 * a12 * b13, a13 - a12])
    B2 = P([a23 - a12])
    B1 = P([2 * a12 * b23, -2 * a23 * b12])
    B0 = P([-a12, 0.0, a23])
    res = (A2 * B0 - A0 * B2) ** 2 - (A2 * B1 - A1 * B2) * (A1 * B0 - A0 * B1)
    sols = []
    for v in np.roots(res.coeffs):
        if abs(v.imag) > 1e-7 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        den = (A1 * B2 - A2 * B1)(v)
        if den == 0:
            continue
        u = (A2 * B0 - A0 * B2)(v) / den
        if u <= 0:
            continue
        q = 1 + u * u - 2 * b12 * u
        if q <= 0:
            continue
        l1 = np.sqrt(a12 / q)
        lam = np.array([l1, u * l1, v * l1])
        # Newton polish on the three distance equations
        for _ in range(3):
            r = np.array([lam[0] ** 2 + lam[1] ** 2 - 2 * b12 * lam[0] * lam[1] - a12,
                          lam[0] ** 2 + lam[2] ** 2 - 2 * b13 * lam[0] * lam[2] - a13,
                          lam[1] ** 2 + lam[2] ** 2 - 2 * b23 * lam[1] * lam[2] - a23])
            J = 2 * np.array([[lam[0] - b12 * lam[1], lam[1] - b12 * lam[0], 0],
                              [lam[0] - b13 * lam[2], 0, lam[2] - b13 * lam[0]],
                              [0, lam[1] - b23 * lam[2], lam[2] - b23 * lam[1]]])
            try:
                lam = lam - np.linalg.solve(J, r)
            except np.linalg.LinAlgError:
                break
        if np.any(lam <= 0):
            continue
        R, t = _kabsch(X, lam[:, None] * y)
        sols.append((R, t))
    return sols


def project(K, R, t, X):
    Xc = X @ R.T + t
    p = Xc @ K.T
    return p[:, :2] / p[:, 2:3], Xc[:, 2]


def hypothesis(p3d, p2d, K, idx4):
    """Best P3P root of one 4-sample (or None): [R|t] (3,4) f64.  Samples that repeat a
    correspondence are rejected."""
    if len(set(int(i) for i in idx4)) < 4:
        return None
    X = p3d[idx4[:3]].astype(np.float64)
    uv = p2d[idx4[:3]].astype(np.float64)
    best, be = None, np.inf
    for R, t in p3p_grunert(X, uv, K):
        pr, z = project(K, R, t, p3d[idx4[3:4]].astype(np.float64))
        if z[0] <= 0:
            continue
        e = np.sum((pr[0] - p2d[idx4[3]].astype(np.float64)) ** 2)
        if e < be:
            be, best = e, np.concatenate([R, t[:, None]], axis=1)
    return best


def hypotheses(p3d, p2d, K, H, seed):
    M = len(p3d)
    S = sample_indices(H, M, seed)
    Rt = np.tile(np.eye(3, 4), (H, 1, 1))
    ok = np.zeros(H, np.uint8)
    if M >= 4:
        for h in range(H):
            b = hypothesis(p3d, p2d, K, S[h])
            if b is not None:
                Rt[h], ok[h] = b, 1
    return Rt, ok, S


def _rodrigues(w):
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * (Kx @ Kx)


GN_STOP_ROT2, GN_STOP_TRANS2 = 1e-18, 1e-14      # csrc/ransac.hip: kGnStopRot2, kGnStopTrans2


def refine(p3d, p2d, K, Rt, sel, iters=10):
    """Gauss-Newton on the reprojection error over correspondences sel (bool or index)."""
    X = p3d[sel].astype(np.float64)
    uv = p2d[sel].astype(np.float64)
    R, t = Rt[:, :3].copy(), Rt[:, 3].copy()
    if len(X) < 4:          # csrc/ransac.hip:gn_solve keeps the pose when fewer than 4 correspondences remain
        return np.array(Rt, np.float64, copy=True)
    for _ in range(iters):
        Xc = X @ R.T + t
        p = Xc @ K.T
        pr = p[:, :2] / p[:, 2:3]
        r = (pr - uv).reshape(-1)
        ipz = 1.0 / p[:, 2]
        a = (K[0][None] - pr[:, 0:1] * K[2][None]) * ipz[:, None]
        b = (K[1][None] - pr[:, 1:2] * K[2][None]) * ipz[:, None]
        J = np.zeros((len(X), 2, 6))
        J[:, 0, :3] = np.cross(Xc, a)      # a . (-[Xc]x) = Xc x a
        J[:, 1, :3] = np.cross(Xc, b)
        J[:, 0, 3:] = a
        J[:, 1, 3:] = b
        J = J.reshape(-1, 6)
        try:
            dx = np.linalg.solve(J.T @ J, -J.T @ r)
        except np.linalg.LinAlgError:
            break
        Q = _rodrigues(dx[:3])
        R, t = Q @ R, Q @ t + dx[3:]
        # csrc/ransac.hip:gn_solve's stopping rule (kGnStopRot2, kGnStopTrans2): the step just applied was below 1e-9 rad and
        # 1e-7 of |t| — quadratic convergence makes the next one ~1e-18
        if float(dx[:3] @ dx[:3]) < GN_STOP_ROT2 and float(dx[3:] @ dx[3:]) < GN_STOP_TRANS2 * (float(t @ t) + 1.0):
            break
    return np.concatenate([R, t[:, None]], axis=1)


def stop_rule(c: int, M: int, b: int, one_minus_conf: float) -> bool:
    """csrc/ransac.hip:ransac_stop, operation for operation (IEEE doubles, binary powering)."""
    if c < 4 or M <= 0 or not (one_minus_conf > 0.0):
        return False
    w = np.float64(c) / np.float64(M)
    w2 = w * w
    base, q, e = np.float64(1.0) - w2 * w2, np.float64(1.0), int(b)
    while e > 0:
        if e & 1:
            q = q * base
        base = base * base
        e >>= 1
    return bool(q <= one_minus_conf)


def evaluated_hypotheses(n_inl, ok, M: int, confidence: float) -> int:
    """How many hypotheses the staged loop scores: the first stage boundary 32 (2^k - 1) at which
    stop_rule holds for the best count so far, else all."""
    H = len(n_inl)
    if confidence >= 1.0:
        return H
    lo, ln = 0, 32
    while lo + ln < H:
        b = lo + ln
        c = int(np.max(np.where(ok[:b].astype(bool), n_inl[:b], 0))) if b else 0
        if stop_rule(c, M, b, 1.0 - confidence):
            return b
        lo, ln = b, ln * 2
    return H


def unpack_mask(mask, M):
    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")[:M]
    return bits.astype(bool)


def pnp_ransac(p3d, p2d, K, H=500, reperr=2.0, seed=0, refine_iters=10, confidence=0.99):
    """-> dict(status, Rt (3,4), inliers (k,) i32, n_inl (H,) [0 past n_eval], best, n_eval, Rt_all, ok)."""
    p3d = np.ascontiguousarray(p3d, np.float32)
    p2d = np.ascontiguousarray(p2d, np.float32)
    K = np.asarray(K, np.float64)
    Rt, ok, S = hypotheses(p3d, p2d, K, H, seed)
    sc = cbind.ransac_score(p3d, p2d, K, Rt, ok, reperr)
    n_eval = evaluated_hypotheses(sc["n_inl"], ok, len(p3d), confidence)
    if n_eval < H:                        # the staged loop never scored the rest
        ok_e = ok.copy()
        ok_e[n_eval:] = 0
        sc = cbind.ransac_score(p3d, p2d, K, Rt, ok_e, reperr)
    best = sc["best"]
    inl = unpack_mask(sc["best_mask"], len(p3d))
    status = int(best >= 0 and sc["n_inl"][best] >= 4)
    pose = Rt[best] if best >= 0 else np.eye(3, 4)
    if status and refine_iters > 0:
        pose = refine(p3d, p2d, K, pose, inl, refine_iters)
        lo = cbind.ransac_score(p3d, p2d, K, pose.reshape(1, 12), np.ones(1, np.uint8), reperr)
        inl = unpack_mask(lo["best_mask"], len(p3d))
        pose = refine(p3d, p2d, K, pose, inl, refine_iters)
        # the reported inliers belong to the returned pose
        fin = cbind.ransac_score(p3d, p2d, K, pose.reshape(1, 12), np.ones(1, np.uint8), reperr)
        inl = unpack_mask(fin["best_mask"], len(p3d))
    return dict(status=status, Rt=pose, inliers=np.nonzero(inl)[0].astype(np.int32) if status else
                np.zeros(0, np.int32), n_inl=sc["n_inl"], best=best, n_eval=n_eval, Rt_all=Rt, ok=ok, samples=S)
