"""The on-disk hand-offs between the reference's stages, read and written with the
same names and layouts so the surrounding scripts keep working.  Pure host I/O (NumPy / json).

Directory convention of the reference: <UH>_<dataset>_obj_<objid>/  (inference.py:23, choosePose.py:95).
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np


def root_dir(UH, dataset: str, objid) -> Path:
    return Path(f"{UH}_{dataset}_obj_{objid}")


def load_model(UH, dataset: str, objid, base: Path | str = "."):
    """genFeat.py:223-228 outputs: (vert1_scaled (N,3) f32, feat1_scaled (N,D) f32, normals (N,3) f64|None)."""
    d = Path(base) / root_dir(UH, dataset, objid) / f"{objid}poseEst"
    pts = np.load(d / "vert1_scaled.npy").astype(np.float32)
    feats = np.load(d / "feat1_scaled.npy").astype(np.float32)
    nrm = d / "normals_scaled.npy"
    return pts, feats, (np.load(nrm) if nrm.exists() else None)


def is_failure(R, t=None) -> bool:
    """pnp()'s failure sentinel is the int triple (1, 1, 1) (inference.py:130-134)."""
    return isinstance(R, (int, np.integer)) or np.ndim(R) == 0


def save_poses(R_list, t_list, UH, dataset: str, objid, base: Path | str = ".", failed: str = "nan"):
    """finalposes.py:234-238 / choosePose.py:303-309 write np.save(pred_R), np.save(pred_t) from Python
    lists — ragged (dtype=object) as soon as one pnp failed.  Here failed images are stored as NaN
    poses (`failed="nan"`, keeps (n,3,3)/(n,3) float64 and the image indexing) or dropped
    (`failed="drop"`); returns the boolean mask of successful images."""
    ok = np.array([not is_failure(R) for R in R_list], bool)
    n = len(R_list)
    R = np.full((n, 3, 3), np.nan)
    t = np.full((n, 3), np.nan)
    for i in range(n):
        if ok[i]:
            R[i], t[i] = np.asarray(R_list[i], np.float64), np.asarray(t_list[i], np.float64).reshape(3)
    if failed == "drop":
        R, t = R[ok], t[ok]
    d = Path(base) / root_dir(UH, dataset, objid)
    d.mkdir(parents=True, exist_ok=True)
    np.save(d / f"{objid}pred_R.npy", R)
    np.save(d / f"{objid}pred_t.npy", t)
    return ok


def load_poses(UH, dataset: str, objid, base: Path | str = "."):
    """choosePose.py:95-96, icp.py:57-58: (pred_R (n,3,3), pred_t (n,3)); tolerates the reference's
    ragged object arrays by mapping sentinel entries to NaN poses."""
    d = Path(base) / root_dir(UH, dataset, objid)
    R = np.load(d / f"{objid}pred_R.npy", allow_pickle=True)
    t = np.load(d / f"{objid}pred_t.npy", allow_pickle=True)
    if R.dtype == object:
        Rn = np.full((len(R), 3, 3), np.nan)
        tn = np.full((len(R), 3), np.nan)
        for i, (a, b) in enumerate(zip(R, t)):
            if not is_failure(a):
                Rn[i], tn[i] = np.asarray(a, np.float64), np.asarray(b, np.float64).reshape(3)
        R, t = Rn, tn
    return np.asarray(R, np.float64), np.asarray(t, np.float64)


def write_top_choices(top_indices, UH, dataset: str, objid, base: Path | str = ".") -> Path:
    """choosePose.py:147-150: one image index per line, best first."""
    p = Path(base) / root_dir(UH, dataset, objid) / f"{objid}top_50_choices.txt"
    p.parent.mkdir(parents=True, exist_ok=True)
    p.write_text("".join(f"{int(i)}\n" for i in top_indices))
    return p


def read_top_choices(UH, dataset: str, objid, base: Path | str = "."):
    """icp.py:37-39."""
    p = Path(base) / root_dir(UH, dataset, objid) / f"{objid}top_50_choices.txt"
    return [int(line.strip()) for line in p.read_text().splitlines() if line.strip()]


def save_vote(error, agreed, UH, dataset: str, objid, base: Path | str = "."):
    """choosePose.py:141-142."""
    d = Path(base) / root_dir(UH, dataset, objid)
    d.mkdir(parents=True, exist_ok=True)
    np.save(d / f"{objid}agreedposes.npy", np.asarray(agreed))
    np.save(d / f"{objid}error.npy", np.asarray(error, np.float64))


def save_relative_poses(table, kind: str, UH, dataset: str, objid, base: Path | str = "."):
    """choosePose.py:111,114: <objid>{gt,pred}_relative_poses.npy, (n,n,4,4) float64."""
    assert kind in ("gt", "pred")
    d = Path(base) / root_dir(UH, dataset, objid)
    d.mkdir(parents=True, exist_ok=True)
    np.save(d / f"{objid}{kind}_relative_poses.npy", np.asarray(table, np.float64))


def write_pred6d_json(R, t, image_ids, path: Path | str):
    """verfication.py:48-52 READS pred6d.json ({img_id: [{"R":[9], "T":[3]}]}) but no script in the
    reference writes it (README.md:92-94 says inference.py does; it does not).  This is the missing
    producer; failed (NaN) poses are skipped."""
    out = {}
    for i, Ri, ti in zip(image_ids, R, t):
        if np.all(np.isfinite(Ri)):
            out[str(int(i))] = [{"R": np.asarray(Ri, np.float64).reshape(9).tolist(),
                                 "T": np.asarray(ti, np.float64).reshape(3).tolist()}]
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    Path(path).write_text(json.dumps(out))
    return out


def read_pred6d_json(path: Path | str):
    """-> (image ids sorted numerically, R (n,3,3), t (n,3)) exactly as verfication.py:50-52, 76-79 index it."""
    d = json.loads(Path(path).read_text())
    keys = sorted(d.keys(), key=lambda x: int(x))
    R = np.array([np.asarray(d[k][0]["R"], np.float64).reshape(3, 3) for k in keys])
    t = np.array([np.asarray(d[k][0]["T"], np.float64) for k in keys])
    return [int(k) for k in keys], R, t


def read_scene_gt(path: Path | str, obj_index: int = 0):
    """BOP scene_gt.json: {img_id: [{"cam_R_m2c":[9], "cam_t_m2c":[3], ...}]} -> ids, R (n,3,3), t (n,3)
    (inference.py:170-186, parsed ONCE instead of once per image)."""
    d = json.loads(Path(path).read_text())
    keys = sorted(d.keys(), key=lambda x: int(x))
    R = np.array([np.asarray(d[k][obj_index]["cam_R_m2c"], np.float64).reshape(3, 3) for k in keys])
    t = np.array([np.asarray(d[k][obj_index]["cam_t_m2c"], np.float64) for k in keys])
    return [int(k) for k in keys], R, t


def read_scene_camera(path: Path | str):
    """BOP scene_camera.json: {img_id: {"cam_K":[9], ...}} -> ids, K (n,3,3)."""
    d = json.loads(Path(path).read_text())
    keys = sorted(d.keys(), key=lambda x: int(x))
    return [int(k) for k in keys], np.array([np.asarray(d[k]["cam_K"], np.float64).reshape(3, 3) for k in keys])


def crop_affine(bbox_xywh, out_size: int = 224, pad: float = 1.2, even_size: bool = True) -> np.ndarray:
    """The 2x3 crop affine M of inference.py:203-219 (host, f64): the bounding box of the visible
    mask (cv2.boundingRect) is shrunk to even width / height (:203-206), then
    size = out/max(w,h)/pad;  M = size * [I | -centre];  M[:,2] += out/2.
    even_size=False skips the decrement for callers that already applied it."""
    x, y, w, h = bbox_xywh
    if even_size:
        if w % 2 != 0:
            w = w - 1
        if h % 2 != 0:
            h = h - 1
    size = out_size / max(w, h) / pad
    c = np.array([x + w / 2.0, y + h / 2.0])
    M = np.concatenate([np.eye(2), -c[:, None]], axis=1) * size
    M[:, 2] += out_size / 2.0
    return M


def crop_camera(K, bbox_xywh, out_size: int = 224, pad: float = 1.2, down_sample: int = 3,
                even_size: bool = True):
    """a4 — the camera-matrix arithmetic of inference.py:203-222 and :260-263 (host, f64):
    cam = [M; 0 0 1] @ K with M = crop_affine(bbox) (odd box sizes are decremented first, exactly as
    the reference does with cv2.boundingRect's output), then the pixel-centre-preserving division
    by the ::down_sample subsampling (camMat[:2,2] += .5; camMat[:2] /= ds; camMat[:2,2] -= .5)."""
    M = crop_affine(bbox_xywh, out_size, pad, even_size)
    cam = np.vstack([M, [0, 0, 1]]) @ np.asarray(K, np.float64)
    cam[:2, 2] += 0.5
    cam[:2] /= down_sample
    cam[:2, 2] -= 0.5
    return cam
