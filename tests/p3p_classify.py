"""Degeneracy measures for a P3P problem on which two solvers disagree (test helper, estimate_pose stage 4).

cv2.solveP3P is not available; the device solver (csrc/p3p_device.hpp, degenerate-conic form) and the oracle's (Grunert's
quartic + numpy.roots + Newton polish + Kabsch, oracle/pnp_oracle.py) are both the builder's.  On estimate_pose's samples —
three pixels of a 74 x 74 lattice, three surface points drawn from the correspondence distribution — the two return
different root SETS for a few percent of the problems.  Every such problem is accounted for by one of:
  sliver     the image triangle or the object triangle is (nearly) degenerate: twice its area over its longest edge squared
             is below SLIVER (two samples on one pixel, collinear pixels, two samples on one surface point ...): P3P is
             ill-posed there and the two algebraic forms break down differently;
  double     an unmatched root has a twin: another root (of either solver) within DOUBLE radians — a real pair about to
             merge into a complex one, which one solver still counts and the other does not;
  grazing    an unmatched root puts one of the three points at less than GRAZE of the farthest one's depth (a root at the
             edge of the positive-depth condition both solvers impose);
  invalid    an unmatched root does not reproject its own three points to INVALID_PX pixels: not a solution at all (the
             solver that returned it was wrong, which on a non-degenerate problem would be a bug: counted separately).
"""
import numpy as np

SLIVER, DOUBLE, GRAZE, INVALID_PX = 2e-2, 2e-2, 1e-3, 1e-4
MATCH_RAD, MATCH_MM = 1e-5, 1e-3


def _rot_angle(Ra, Rb):
    f = np.linalg.norm(np.asarray(Ra, np.float64) - np.asarray(Rb, np.float64))
    return float(2.0 * np.arcsin(min(1.0, f / (2.0 * np.sqrt(2.0)))))


def sliverness(P):
    """2 * area / (longest edge)^2 of the triangle P (3, 2 or 3): 0 for collinear or coincident points."""
    P = np.asarray(P, np.float64)
    if P.shape[1] == 2:
        P = np.concatenate([P, np.zeros((3, 1))], 1)
    e = [np.linalg.norm(P[i] - P[j]) for i, j in ((0, 1), (0, 2), (1, 2))]
    L = max(e)
    return float(np.linalg.norm(np.cross(P[1] - P[0], P[2] - P[0])) / (L * L)) if L > 0 else 0.0


def measures(X3, uv3, K, roots_a, roots_b):
    """roots_*: lists of (R, t).  Returns dict(sliver_img, sliver_obj, unmatched=[...per unmatched root: twin, graze, resid_px, who])."""
    X3, uv3 = np.asarray(X3, np.float64), np.asarray(uv3, np.float64)
    used_b = set()
    un = []
    pairs = []
    for ia, (Ra, ta) in enumerate(roots_a):
        hit = next((ib for ib, (Rb, tb) in enumerate(roots_b) if ib not in used_b and _rot_angle(Ra, Rb) < MATCH_RAD and
                    np.linalg.norm(np.asarray(ta) - np.asarray(tb)) < MATCH_MM), None)
        if hit is None:
            un.append(("a", ia))
        else:
            used_b.add(hit)
            pairs.append((ia, hit))
    un += [("b", ib) for ib in range(len(roots_b)) if ib not in used_b]
    allr = [("a", i, r) for i, r in enumerate(roots_a)] + [("b", i, r) for i, r in enumerate(roots_b)]
    out = []
    for who, i in un:
        R, t = (roots_a if who == "a" else roots_b)[i]
        # the twin may be a root of either solver (but not this root's own match: it has none, and not itself)
        twin = min([_rot_angle(R, r2[0]) for w2, i2, r2 in allr if not (w2 == who and i2 == i)] or [np.inf])
        Xc = X3 @ np.asarray(R, np.float64).T + np.asarray(t, np.float64)
        p = Xc @ np.asarray(K, np.float64).T
        resid = float(np.abs(p[:, :2] / p[:, 2:3] - uv3).max()) if np.all(Xc[:, 2] != 0) else np.inf
        graze = float(Xc[:, 2].min() / max(Xc[:, 2].max(), 1e-300))
        out.append(dict(who=who, twin=twin, graze=graze, resid_px=resid))
    return dict(sliver_img=sliverness(uv3), sliver_obj=sliverness(X3), unmatched=out, matched=pairs)


def explain(m):
    """'' when the root sets agree, else the class that accounts for the disagreement ('unexplained' if none does)."""
    if not m["unmatched"]:
        return ""
    if min(m["sliver_img"], m["sliver_obj"]) < SLIVER:
        return "sliver"
    cls = []
    for u in m["unmatched"]:
        if u["resid_px"] > INVALID_PX:
            cls.append("invalid")
        elif u["twin"] < DOUBLE:
            cls.append("double")
        elif u["graze"] < GRAZE:
            cls.append("grazing")
        else:
            cls.append("unexplained")
    for c in ("unexplained", "invalid", "grazing", "double"):
        if c in cls:
            return c
    return "unexplained"
