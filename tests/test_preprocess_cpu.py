"""CPU: the oracle of the image front end (oracle/preprocess_oracle.py).  normalize is pinned by the vector
the reference's own function produced; bounding_rect and warp_affine restate OpenCV contracts (parity
unpinned) and are checked for the properties that define them."""
from pathlib import Path

import numpy as np

from oracle import preprocess_oracle as pp

G = Path(__file__).resolve().parent / "golden"


def test_normalize_matches_reference_output():
    g = np.load(G / "ref_normalize.npz")
    assert np.array_equal(pp.normalize(g["img"]), g["out"])


def test_bounding_rect_contract():
    m = np.zeros((40, 60), np.uint8)
    assert pp.bounding_rect(m) == (0, 0, 0, 0)
    m[7, 11] = 255
    assert pp.bounding_rect(m) == (11, 7, 1, 1)
    m[30, 3] = 1
    m[12, 50] = 9
    assert pp.bounding_rect(m) == (3, 7, 48, 24)


def test_warp_affine_integer_shift_and_border():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(30, 40, 3), dtype=np.uint8)
    M = np.array([[1.0, 0, -5], [0, 1.0, -3]])              # crop(x, y) = img(x + 5, y + 3)
    out = pp.warp_affine(img, M, 20)
    assert np.array_equal(out, img[3:23, 5:25])
    M = np.array([[1.0, 0, 4], [0, 1.0, 6]])                # crop(x, y) = img(x - 4, y - 6): the rim is outside -> 0
    out = pp.warp_affine(img, M, 20)
    assert (out[:6] == 0).all() and (out[:, :4] == 0).all() and np.array_equal(out[6:, 4:], img[:14, :16])
    M = np.array([[2.0, 0, 0], [0, 2.0, 0]])                # 2x magnification: odd crop pixels are midpoints
    out = pp.warp_affine(img, M, 20).astype(np.float64)
    mid = (img[:10, :9].astype(np.float64) + img[:10, 1:10]) / 2
    assert np.abs(out[0:20:2, 1:19:2] - mid).max() <= 0.5
