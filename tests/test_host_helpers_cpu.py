"""CPU: host-side helpers added in round 4 — the MIN-table rehearsal of shard.py and the P3P disagreement classifier."""
import numpy as np
import pytest
import torch

from imagesequenceregistrationfor6dposeestimationlabeling_amd import shard


def test_min_table_image_and_merge_equal_an_allreduce_min():
    """Eight ranks' contributions built by min_table_image and merged by merge_min_tables = the table an
    all_reduce(MIN) over the int64 images leaves: owned entries pass through bit for bit, unowned stay +inf, a NaN entry
    (a failed image's pose) cannot win, the merge order does not matter, an empty contribution may sit past the end."""
    rng = np.random.default_rng(3)
    n, size = 511, 8
    full = rng.uniform(0.0, 50.0, n)
    full[100] = np.nextafter(full[7], 0.0)              # less than one ulp below another rank's value
    full[400] = np.nan
    parts = []
    for r in range(size):
        lo, hi = shard.owned_pairs(n + 1, r, size)
        parts.append(shard.min_table_image(torch.from_numpy(full[lo:hi].copy()), lo, n))
    parts.append(shard.min_table_image(torch.empty(0, dtype=torch.float64), n + 5, n))      # a rank that owns nothing
    merged = shard.merge_min_tables(parts)
    back = shard.merge_min_tables(parts[::-1])
    assert torch.equal(merged.view(torch.int64), back.view(torch.int64))
    got = merged.numpy()
    ok = ~np.isnan(full)
    assert np.array_equal(got[ok].view(np.int64), full[ok].view(np.int64))
    assert np.isinf(got[400]) or np.isnan(got[400])
    i, v = shard.first_min(merged)
    assert i == int(np.nanargmin(full)) and v == np.nanmin(full)
    with pytest.raises(ValueError):
        shard.min_table_image(torch.ones(4, dtype=torch.float64), n - 2, n)


def _rot(axis, ang):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(np.asarray(axis, float) / np.linalg.norm(axis) * ang).as_matrix()


def test_p3p_classifier_on_constructed_root_sets():
    from tests import p3p_classify as pc
    K = np.array([[100.0, 0, 16], [0, 100.0, 16], [0, 0, 1]])
    X = np.array([[10.0, 0, 0], [0, 12.0, 0], [-8.0, -5.0, 3.0]])
    R0, t0 = _rot([1, 2, 3], 0.4), np.array([1.0, -2.0, 300.0])
    Xc = X @ R0.T + t0
    uv = (Xc @ K.T)[:, :2] / Xc[:, 2:3]
    root = (R0, t0)
    # the same sets: nothing to explain
    m = pc.measures(X, uv, K, [root], [root])
    assert pc.explain(m) == "" and m["sliver_img"] > 0.1 and m["sliver_obj"] > 0.1
    # a twin of the root within DOUBLE radians that one solver lacks: a merging pair (a hand-made twin does not reproject
    # the points exactly, and the classifier reports that first)
    twin = (_rot([0, 0, 1], 1e-3) @ R0, t0)
    m2 = pc.measures(X, uv, K, [root, twin], [root])
    assert len(m2["unmatched"]) == 1 and m2["unmatched"][0]["twin"] < pc.DOUBLE and pc.explain(m2) in ("double", "invalid")
    # an extra root that does not reproject its points and has no twin: invalid
    far = (_rot([1, 0, 0], 1.0) @ R0, t0 + np.array([30.0, 0, 0]))
    assert pc.explain(pc.measures(X, uv, K, [root, far], [root])) == "invalid"
    # collinear pixels: whatever the root sets, the problem is a sliver
    uvc = np.array([[1.0, 1.0], [2.0, 2.0], [3.0, 3.0]])
    assert pc.explain(pc.measures(X, uvc, K, [root], [])) == "sliver"
    assert pc.sliverness(uvc) == 0.0 and pc.sliverness(np.array([[0.0, 0], [1, 0], [0, 1]])) == pytest.approx(0.5)


def test_morton_order_is_a_stable_locality_preserving_permutation():
    """registration.morton_order (the row order the ICP's per-wave tile cull lives on): a permutation; consecutive rows
    of the permuted cloud are neighbours in space; rigid motions keep the locality; degenerate clouds are handled."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd.registration import morton_order
    g = torch.Generator().manual_seed(3)
    p = torch.rand(5000, 3, generator=g) * torch.tensor([60.0, 40.0, 45.0])
    o = morton_order(p)
    assert torch.equal(torch.sort(o).values, torch.arange(5000))
    step = lambda q: (q[1:] - q[:-1]).norm(dim=1).mean().item()
    assert step(p[o]) < 0.25 * step(p)
    # the order computed in the cloud's own frame still walks a rotated + shifted copy locally
    c, s = np.cos(0.7), np.sin(0.7)
    R = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    moved = p @ R.T + torch.tensor([5.0, -3.0, 800.0])
    assert abs(step(moved[o]) - step(p[o])) < 1e-3 * step(p[o])
    # a 256-row tile of an ordered SURFACE cloud (20 000 points on an ellipsoid shell) is a compact box — the typical one
    # a fifth of the object's extent (tiles that straddle a jump of the Z curve are longer)
    u = torch.randn(20000, 3, generator=g)
    shell = u / u.norm(dim=1, keepdim=True) * torch.tensor([30.0, 20.0, 22.0])
    so = shell[morton_order(shell)]
    tiles = so[: 78 * 256].reshape(78, 256, 3)
    ext = (tiles.max(1).values - tiles.min(1).values).max(1).values
    assert ext.median() < 0.25 * 60.0
    rnd = shell[: 78 * 256].reshape(78, 256, 3)
    assert (rnd.max(1).values - rnd.min(1).values).max(1).values.median() > 0.9 * 60.0
    # stability on equal codes, one point, all points equal
    assert torch.equal(morton_order(torch.zeros(7, 3)), torch.arange(7))
    assert torch.equal(morton_order(torch.tensor([[1.0, 2.0, 3.0]])), torch.tensor([0]))
    q = torch.tensor([[0.0, 0, 0], [1, 1, 1], [0, 0, 0], [1, 1, 1]])
    assert morton_order(q).tolist() == [0, 2, 1, 3]


def test_morton_order_of_empty_and_single_row_clouds():
    """min / max of an empty cloud raise inside torch; the order of 0 or 1 rows is the identity, and
    icp_point_to_point leaves clouds of at most one search tile unsorted."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import registration as reg
    assert reg.morton_order(torch.zeros(0, 3)).tolist() == []
    assert reg.morton_order(torch.zeros(1, 3)).tolist() == [0]
    assert reg.MORTON_MIN_ROWS == 256


def test_cloud_key_tells_axis_permuted_and_row_swapped_clouds_apart():
    """the distance-field cache of vote_rows was keyed on two sums that a permutation of the coordinates inside
    the rows left unchanged.  The key is now a position-weighted hash of the bit patterns."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd.sequence import cloud_key
    rng = np.random.default_rng(11)
    p = torch.from_numpy(rng.normal(size=(500, 3)).astype(np.float32))
    k0 = cloud_key(p)
    assert cloud_key(p.clone()) == k0
    assert cloud_key(p[:, [1, 2, 0]].contiguous()) != k0          # cyclic axis rotation: same sum, same row-weighted sum
    assert cloud_key(p[:, [1, 0, 2]].contiguous()) != k0
    q = p.clone(); q[[3, 4]] = q[[4, 3]]
    assert cloud_key(q) != k0
    q = p.clone(); q[17, 1] = torch.nextafter(q[17, 1], torch.tensor(1e9))
    assert cloud_key(q) != k0
    assert cloud_key(p.double()) != k0
    # a non-contiguous view hashes as its contents
    big = torch.zeros(500, 6); big[:, :3] = p
    assert cloud_key(big[:, :3]) == k0


def test_the_stress_scripts_parse():
    """tests/stress_gpu.py and tools/stress_corr_screened.py are scripts for the GPU box, not collected by pytest: at least keep them
    syntactically alive here."""
    import ast
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for rel in ("tests/stress_gpu.py", "tools/stress_corr_screened.py"):
        ast.parse((root / rel).read_text(), filename=rel)
