"""TEST INFRASTRUCTURE, NOT PRODUCT — NumPy restatement of the image front end of the per-image loop
(inference.py:135-141, 202, 224-232).

normalize() is the reference's own function (pinned: tests/golden/ref_normalize.npz was produced by executing
it).  PARITY UNPINNED: cv2.boundingRect and cv2.warpAffine are OpenCV calls, OpenCV is not in this image and
the reference ships no image fixture.  bounding_rect restates the documented contract (box of the non-zero
pixels, (0,0,0,0) if none).  warp_affine is the textbook bilinear warp — inverse map in f64, neighbours outside
the frame count as 0 (BORDER_CONSTANT 0), round half to even — in exactly the operation order of
csrc/crop_normalize.hip; OpenCV [from memory] additionally quantises the sample position to 1/32 px and the
weights to 2^-15, so its bytes may differ by a grey level at some pixels.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize(img: np.ndarray) -> np.ndarray:
    """inference.py:135-141."""
    mu, std = IMAGENET_MEAN, IMAGENET_STD
    if img.dtype == np.uint8:
        img = img / 255
    return (img - mu) / std


def bounding_rect(gray: np.ndarray) -> tuple[int, int, int, int]:
    """cv2.boundingRect(mask[:, :, 0]) (inference.py:202): x, y, w, h of the non-zero pixels."""
    ys, xs = np.nonzero(gray)
    if len(xs) == 0:
        return 0, 0, 0, 0
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def warp_affine(img: np.ndarray, M: np.ndarray, size: int) -> np.ndarray:
    """cv2.warpAffine(img, M, (size, size)) with the defaults INTER_LINEAR / BORDER_CONSTANT(0), as defined above.
    img (H, W, C) u8, M (2, 3) source -> crop."""
    H, W = img.shape[:2]
    M = np.asarray(M, np.float64)
    det = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    i00, i01, i10, i11 = M[1, 1] / det, -M[0, 1] / det, -M[1, 0] / det, M[0, 0] / det
    i02 = -(i00 * M[0, 2] + i01 * M[1, 2])
    i12 = -(i10 * M[0, 2] + i11 * M[1, 2])
    y, x = np.mgrid[0:size, 0:size].astype(np.float64)
    sx = (i00 * x + i01 * y) + i02
    sy = (i10 * x + i11 * y) + i12
    fx0, fy0 = np.floor(sx), np.floor(sy)
    fx, fy = sx - fx0, sy - fy0
    x0, y0 = fx0.astype(np.int64), fy0.astype(np.int64)

    def px(xx, yy):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.float64)
        return np.where(ok[..., None], v, 0.0)

    w00, w01, w10, w11 = (1.0 - fx) * (1.0 - fy), fx * (1.0 - fy), (1.0 - fx) * fy, fx * fy
    v = ((px(x0, y0) * w00[..., None] + px(x0 + 1, y0) * w01[..., None]) + px(x0, y0 + 1) * w10[..., None]) \
        + px(x0 + 1, y0 + 1) * w11[..., None]
    return np.rint(v).astype(np.uint8)


def crop_inputs(rgb: np.ndarray, mask: np.ndarray, M: np.ndarray, size: int = 224, use_mask: bool = True):
    """inference.py:224-232: -> inputIM (3, size, size) f32, cropMask[:, :, 0] (size, size) u8."""
    crop_rgb = warp_affine(rgb, M, size)
    crop_mask = warp_affine(mask if mask.ndim == 3 else mask[..., None], M, size)
    if use_mask:
        crop_rgb[crop_mask[:, :, 0] == 0] = 0
    return np.moveaxis(normalize(crop_rgb).astype("float32"), 2, 0), crop_mask[:, :, 0]
