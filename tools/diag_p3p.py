"""Classify every disagreement between the device P3P (degenerate-conic solver, csrc/p3p_device.hpp) and the
oracle's (Grunert quartic + numpy.roots + Kabsch, oracle/pnp_oracle.py): python tools/diag_p3p.py
For each hypothesis: is the device pose a TRUE P3P solution (re-projects its three sample points to < 1e-6 px, proper
rotation, positive depths)?  Is it one of the oracle's roots?  Whose pick has the smaller 4th-point error?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, synth
from oracle import pnp_oracle as po

def scene(seed, M, kind):
    rng = np.random.default_rng(seed)
    pts = {"tless": synth.tless_like, "ell": synth.bumpy_ellipsoid, "rev": synth.revolution}[kind](rng, 4000)
    K = synth.camera()
    R, t = synth.random_poses(rng, 1)
    p3d, p2d, inl = synth.pnp_case(rng, pts, K, R[0], t[0], M, 0.5, 0.3)
    return K, p3d, p2d

dev = torch.device("cuda:0")
tot = {}
for seed, M, H, kind in ((2, 245760, 1000, "tless"), (1, 2000, 1000, "tless"), (3, 777, 1000, "ell"), (13, 20000, 1000, "rev")):
    K, p3d, p2d = scene(seed, M, kind)
    Rt, ok, smp = ops.p3p_hypotheses(torch.from_numpy(p3d).to(dev), torch.from_numpy(p2d).to(dev), K, H, seed=seed * 1000003, want_samples=True)
    torch.cuda.synchronize()
    Rt, ok, S = Rt.cpu().numpy(), ok.cpu().numpy(), smp.cpu().numpy()
    cls = {}
    for h in range(H):
        if len(set(S[h].tolist())) < 4:
            c = "repeated sample: both reject" if not ok[h] else "DEVICE accepted a repeated sample"
            cls[c] = cls.get(c, 0) + 1
            continue
        X = p3d[S[h][:3]].astype(np.float64); uv = p2d[S[h][:3]].astype(np.float64)
        X4 = p3d[S[h][3:4]].astype(np.float64); uv4 = p2d[S[h][3]].astype(np.float64)
        sols = [(R_, t_) for R_, t_ in po.p3p_grunert(X, uv, K)]
        e4 = []
        for R_, t_ in sols:
            pr, z = po.project(K, R_, t_, X4)
            e4.append(float(np.sum((pr[0] - uv4) ** 2)) if z[0] > 0 else np.inf)
        ob = int(np.argmin(e4)) if e4 and np.isfinite(min(e4)) else -1
        if not ok[h]:
            c = "both reject" if ob < 0 else "DEVICE missed: oracle has a valid root"
            cls[c] = cls.get(c, 0) + 1
            continue
        d = Rt[h]
        pr3, z3 = po.project(K, d[:, :3], d[:, 3], X)
        pr4, z4 = po.project(K, d[:, :3], d[:, 3], X4)
        valid = np.abs(pr3 - uv).max() < 1e-6 and z3.min() > 0 and z4[0] > 0 and abs(np.linalg.det(d[:, :3]) - 1) < 1e-9
        de4 = float(np.sum((pr4[0] - uv4) ** 2))
        match = [i for i, (R_, t_) in enumerate(sols) if synth.rot_angle(R_, d[:, :3]) < 1e-6 and np.linalg.norm(t_ - d[:, 3]) < 1e-4]
        if not valid:
            c = "DEVICE pose is not a P3P solution"
        elif ob >= 0 and match and match[0] == ob:
            c = "agree"
        elif ob < 0:
            c = "oracle found no valid root, device pose is a true solution"
        elif not match:
            c = ("oracle misses the device's (true) root; device pick is better" if de4 <= e4[ob] + 1e-9
                 else "DEVICE missed the oracle's better root")
        else:
            c = "same root set, different pick: " + ("4th-point errors tie" if abs(de4 - e4[ob]) < 1e-6 * max(1.0, de4) else
                                                      ("device pick better" if de4 < e4[ob] else "DEVICE pick worse"))
        cls[c] = cls.get(c, 0) + 1
    print(f"scene seed={seed} M={M} H={H} {kind}:")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        print(f"   {v:5d}  {k}")
        tot[k] = tot.get(k, 0) + v
print("all scenes:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"   {v:5d}  {k}")
