"""GPU parity: the network-output -> K1 hand-off (isr_prep_queries, isr_select_top_dev,
sequence.register_crop) against the literal torch expressions of inference.py:248-290 on torch-CPU."""
import numpy as np
import pytest
import torch

from imagesequenceregistrationfor6dposeestimationlabeling_amd import synth

pytestmark = pytest.mark.gpu


def _reference(feat, mask, ds, n_feat):
    """inference.py:248-279, literally, on the CPU."""
    imfeatsfull = feat[None] if feat.ndim == 3 else feat
    imfeats = imfeatsfull[..., 0:n_feat]
    inputMask = mask[:, :, 0] if mask.ndim == 3 else mask
    imfeats = imfeats[:, ::ds, ::ds]
    inputMask = inputMask[::ds, ::ds]
    maskIds = torch.where(inputMask)
    maskedfeats = imfeats[0][maskIds]
    ep2d = np.zeros((maskedfeats.shape[0], 2))
    ep2d[:, 0] = maskIds[1].numpy()
    ep2d[:, 1] = maskIds[0].numpy()
    return maskedfeats, ep2d


def _blob_mask(rng, H, W, ch=3):
    yy, xx = np.mgrid[:H, :W]
    m = ((yy - H * 0.55) ** 2 / (H * 0.3) ** 2 + (xx - W * 0.45) ** 2 / (W * 0.35) ** 2) < 1.0
    m &= rng.random((H, W)) > 0.1                      # holes
    m = (m * 255).astype(np.uint8)
    return np.repeat(m[:, :, None], ch, axis=2) if ch else m


@pytest.mark.parametrize("H,W,C,ds,ch", [(224, 224, 13, 3, 3), (225, 223, 13, 3, 3), (64, 80, 12, 1, 0), (50, 70, 20, 4, 1)])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "bf16_log2"])
def test_prep_queries_matches_torch(cuda0, H, W, C, ds, ch, dtype):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(H * 7 + W + ds)
    feat = torch.from_numpy(rng.normal(0, 2, (1, H, W, C)).astype(np.float32))
    mask = torch.from_numpy(_blob_mask(rng, H, W, ch))
    mf, ep2d = _reference(feat, mask, ds, 12)
    Q, pix, n_dev = ops.prep_queries(feat.to(cuda0), mask.to(cuda0), c0=0, D=12, step=ds, dtype=dtype)
    n = int(n_dev.item())
    assert n == mf.shape[0] and n > 0
    S = ((H + ds - 1) // ds) * ((W + ds - 1) // ds)
    assert Q.shape[0] == S and pix.shape == (S, 2)
    assert np.array_equal(pix[:n].cpu().numpy().astype(np.float64), ep2d)
    Qh = Q.cpu()
    if dtype == "f32":
        assert Qh.shape[1] == 12 and torch.equal(Qh[:n], mf)
    else:
        want = ops.prescale_queries_log2(mf) if dtype == "bf16_log2" else mf.bfloat16()
        assert Qh.shape[1] == 16
        assert torch.equal(Qh[:n, :12].view(torch.int16), want.view(torch.int16))      # same roundings, bit for bit
        assert (Qh[:, 12:].float() == 0).all()
    assert (Qh[n:].float() == 0).all()


def test_prep_queries_empty_and_full_mask(cuda0):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(3)
    feat = torch.from_numpy(rng.normal(0, 1, (30, 33, 12)).astype(np.float32)).to(cuda0)
    for fill, want in ((0, 0), (255, 10 * 11)):
        mask = torch.full((30, 33), fill, dtype=torch.uint8, device=cuda0)
        Q, pix, n_dev = ops.prep_queries(feat, mask, step=3, dtype="f32")
        assert int(n_dev.item()) == want
    assert torch.equal(Q.cpu(), feat.cpu()[::3, ::3].reshape(-1, 12))


@pytest.mark.parametrize("P,n", [(5625, 4000), (5625, 5625), (5625, 300), (5625, 1), (5625, 0), (100000, 64123)])
def test_select_top_dev_equals_host_count(cuda0, P, n):
    """isr_select_top_dev on the first n of P capacity values = isr_select_top on those n values
    (and = the reference's sort/where expressions, which test_gpu_select.py pins for the host count)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    rng = np.random.default_rng(P + n)
    x = torch.from_numpy(rng.normal(-3, 2, P).astype(np.float32)).to(cuda0)
    x[n:] = 1.0e3                                       # padding values that would win if they were counted
    n_dev = torch.tensor([n], dtype=torch.int32, device=cuda0)
    keep, M, thr = ops.select_top(x, n_dev=n_dev)
    if n == 0:
        assert int(M.item()) == 0
        return
    keep_h, M_h, thr_h = ops.select_top(x[:n].contiguous())
    assert int(M.item()) == int(M_h.item())
    m = int(M.item())
    assert torch.equal(keep[:m], keep_h[:m])
    if n >= 2:
        assert float(thr.item()) == float(thr_h.item())


def test_register_crop_equals_register_image(cuda0):
    """The device-side hand-off feeds the same chain: a crop registered from the network output gives
    the pose of registering the (host-)compacted descriptors, and the planted pose is recovered."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, sequence
    rng = np.random.default_rng(11)
    N, D, H, W, ds = 4000, 12, 224, 224, 3
    pts = synth.tless_like(rng, N)
    keys = synth.unit_keys(rng, N, D, tau=6.0)
    R, t = synth.random_poses(rng, 1)
    Kc = synth.camera(75, 75, f=400.0)                              # the object fills the 75 x 75 sub-sampled crop
    proj = synth.project(Kc, R[0], t[0], pts)
    from scipy.spatial import cKDTree
    gy, gx = np.mgrid[:75, :75]
    d, nn = cKDTree(proj).query(np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64))
    inside = (d < 0.6).reshape(75, 75)                              # sub-sampled pixels that see a surface point
    rows, cols = np.nonzero(inside)
    nn = nn.reshape(75, 75)[rows, cols]
    mask = (rng.random((H, W)) < 0.5).astype(np.uint8) * 255        # pixels off the ::3 lattice: must be ignored
    mask[::ds, ::ds] = 0
    mask[rows * ds, cols * ds] = 255
    mask = np.repeat(mask[:, :, None], 3, axis=2)
    feat = rng.normal(0, 0.3, (1, H, W, 13)).astype(np.float32)
    wrong = rng.random(len(nn)) < 0.25
    nn_feat = np.where(wrong, rng.integers(N, size=len(nn)), nn)
    feat[0, rows * ds, cols * ds, :12] = keys[nn_feat] + 0.2 * rng.normal(size=(len(nn), D)).astype(np.float32)
    model = sequence.SequenceModel(keys=torch.from_numpy(keys).to(cuda0), pts=torch.from_numpy(pts).to(cuda0))
    res, n_dev = sequence.register_crop(model, torch.from_numpy(feat).to(cuda0), torch.from_numpy(mask).to(cuda0), Kc,
                                        n_feat=12, down_sample=ds, itr=300, seed=4)
    n = int(n_dev.item())
    assert n == len(rows)
    mf, ep2d = _reference(torch.from_numpy(feat), torch.from_numpy(mask), ds, 12)
    ref = sequence.register_image(model, mf.to(cuda0), torch.from_numpy(ep2d.astype(np.float32)).to(cuda0), Kc,
                                  itr=300, seed=4)
    torch.cuda.synchronize()
    assert torch.equal(res.idx[:n], ref.idx) and torch.equal(res.logp[:n], ref.logp)
    m = int(ref.M.item())
    assert int(res.M.item()) == m and torch.equal(res.keep[:m], ref.keep[:m])
    assert int(res.status.item()) == 1 and torch.equal(res.pose, ref.pose)
    # the planted geometry is a nearest-projection assignment: correct to about a pixel
    pose = res.pose.cpu().numpy()
    assert synth.rot_angle(pose[:, :3], R[0]) < 0.05


def _crop_case(rng, pts, keys, Kc, R, t, H, W, ds, kind):
    """One crop as the network would deliver it: features planted at the ::ds lattice pixels that see a surface
    point.  kind: 'object' (ragged mask), 'empty' (no masked pixel), 'full' (every lattice pixel masked)."""
    from scipy.spatial import cKDTree
    N, D = keys.shape
    S1 = (H + ds - 1) // ds
    proj = synth.project(Kc, R, t, pts)
    gy, gx = np.mgrid[:S1, :S1]
    d, nn = cKDTree(proj).query(np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64))
    inside = (d < 0.6).reshape(S1, S1)
    if kind == "empty":
        inside[:] = False
    elif kind == "full":
        inside[:] = True
    elif kind == "holes":
        inside &= rng.random((S1, S1)) > 0.3
    rows, cols = np.nonzero(inside)
    nn = nn.reshape(S1, S1)[rows, cols]
    mask = (rng.random((H, W)) < 0.5).astype(np.uint8) * 255        # pixels off the lattice: must be ignored
    mask[::ds, ::ds] = 0
    mask[rows * ds, cols * ds] = 255
    feat = rng.normal(0, 0.3, (H, W, 13)).astype(np.float32)
    wrong = rng.random(len(nn)) < 0.25
    nn_feat = np.where(wrong, rng.integers(N, size=len(nn)), nn)
    feat[rows * ds, cols * ds, :D] = keys[nn_feat] + 0.2 * rng.normal(size=(len(nn), D)).astype(np.float32)
    return feat, np.repeat(mask[:, :, None], 3, axis=2), len(rows)


@pytest.mark.parametrize("dtype,n_streams", [("f32", 1), ("bf16_log2", 1), ("bf16_log2", 3)])
def test_register_crops_equals_register_crop_per_image(cuda0, dtype, n_streams):
    """The batched per-image loop (isr_prep_queries_batch + ONE K1 launch per group + one filter / RANSAC chain):
    every image bit-identical to the single-image chain, on ragged masks including an empty and a full one, with
    per-image cameras and seeds, across a group boundary (7 images, groups of 4)."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, sequence
    rng = np.random.default_rng(17)
    N, D, H, W, ds = 4000, 12, 224, 224, 3
    pts = synth.tless_like(rng, N)
    keys = synth.unit_keys(rng, N, D, tau=6.0)
    kinds = ["object", "holes", "empty", "object", "full", "holes", "object"]
    n = len(kinds)
    R, t = synth.random_poses(rng, n)
    cams = np.stack([synth.camera(75, 75, f=380.0 + 10.0 * i) for i in range(n)])
    cases = [_crop_case(rng, pts, keys, cams[i], R[i], t[i], H, W, ds, kinds[i]) for i in range(n)]
    feats = torch.from_numpy(np.stack([c[0] for c in cases])).to(cuda0)
    masks = torch.from_numpy(np.stack([c[1] for c in cases])).to(cuda0)
    if dtype == "f32":
        model = sequence.SequenceModel(keys=torch.from_numpy(keys).to(cuda0), pts=torch.from_numpy(pts).to(cuda0))
    else:
        model = sequence.SequenceModel(keys=torch.from_numpy(keys).bfloat16().to(cuda0), pts=torch.from_numpy(pts).to(cuda0),
                                       log2_queries=True)
    seeds = [100 + 3 * i for i in range(n)]
    # the raw batched hand-off first: image b's rows equal the single-image call's
    Qb, pixb, nb = ops.prep_queries_batch(feats, masks, D=12, step=ds, dtype=dtype)
    for b in range(n):
        Q1, pix1, n1 = ops.prep_queries(feats[b], masks[b], D=12, step=ds, dtype=dtype)
        assert int(nb[b].item()) == int(n1.item()) == cases[b][2]
        assert torch.equal(Qb[b].view(torch.int16 if dtype != "f32" else torch.int32),
                           Q1.view(torch.int16 if dtype != "f32" else torch.int32))
        assert torch.equal(pixb[b], pix1)
    res, n_dev = sequence.register_crops(model, feats, masks, cams, n_feat=12, down_sample=ds, itr=300, seeds=seeds,
                                         refine_iters=6, group=4, n_streams=n_streams)
    torch.cuda.synchronize()
    assert n_dev.cpu().tolist() == [c[2] for c in cases]
    for b in range(n):
        ref, n1 = sequence.register_crop(model, feats[b:b + 1], masks[b], cams[b], n_feat=12, down_sample=ds, itr=300,
                                         seed=seeds[b], refine_iters=6)
        torch.cuda.synchronize()
        k = cases[b][2]
        assert torch.equal(res[b].idx[:k], ref.idx[:k]) and torch.equal(res[b].logp[:k], ref.logp[:k]), b
        m = int(ref.M.item())
        assert int(res[b].M.item()) == m and torch.equal(res[b].keep[:m], ref.keep[:m]), b
        assert int(res[b].status.item()) == int(ref.status.item()) == (0 if kinds[b] == "empty" else 1), b
        assert int(res[b].n_eval.item()) == int(ref.n_eval.item())
        ni = int(ref.n_inl.item())
        assert int(res[b].n_inl.item()) == ni and torch.equal(res[b].inl_idx[:ni], ref.inl_idx[:ni]), b
        if kinds[b] != "empty":
            assert torch.equal(res[b].pose, ref.pose), b
        if kinds[b] in ("object", "holes"):
            assert synth.rot_angle(res[b].pose.cpu().numpy()[:, :3], R[b]) < 0.05


def test_register_frames_equals_register_frame(cuda0):
    """Raw frames -> poses, batched (crop front end for all frames, ONE encoder call, register_crops) against
    the single-frame driver: same crops, same camera matrices, bit-identical poses."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import sequence
    rng = np.random.default_rng(5)
    N, D = 3000, 12
    pts = synth.tless_like(rng, N)
    keys = synth.unit_keys(rng, N, D, tau=6.0)
    model = sequence.SequenceModel(keys=torch.from_numpy(keys).to(cuda0), pts=torch.from_numpy(pts).to(cuda0))
    n, Hf, Wf = 3, 240, 320
    rgbs = rng.integers(0, 256, (n, Hf, Wf, 3), dtype=np.uint8)
    masks = np.zeros((n, Hf, Wf, 3), np.uint8)
    for i in range(n):
        masks[i, 40 + 10 * i:150 + 7 * i, 60 + 5 * i:200 - 9 * i] = 255
    Kf = synth.camera(Wf, Hf)
    g = torch.Generator(device=cuda0).manual_seed(3)
    Wenc = torch.randn(13, 3, device=cuda0, generator=g)

    def encoder(x):      # a stand-in network: per-pixel linear map (B, 3, r, r) -> (B, 13, r, r), written elementwise
        w = Wenc[None, :, :, None, None]               # so that a pixel's features do not depend on the batch it is in
        return (x[:, None, 0] * w[:, :, 0] + x[:, None, 1] * w[:, :, 1]) + x[:, None, 2] * w[:, :, 2]

    res, n_dev, cam = sequence.register_frames(model, rgbs, masks, Kf, encoder, n_feat=12, itr=100, seeds=[7, 8, 9],
                                               refine_iters=4, group=2)
    torch.cuda.synchronize()
    for i in range(n):
        r1, n1, cam1 = sequence.register_frame(model, rgbs[i], masks[i], Kf, encoder, n_feat=12, itr=100, seed=7 + i,
                                               refine_iters=4)
        torch.cuda.synchronize()
        assert np.array_equal(cam[i], cam1) and int(n_dev[i].item()) == int(n1.item()) > 0
        k = int(n1.item())
        assert torch.equal(res[i].idx[:k], r1.idx[:k]) and torch.equal(res[i].logp[:k], r1.logp[:k])
        assert int(res[i].status.item()) == int(r1.status.item())
        if int(r1.status.item()):
            assert torch.equal(res[i].pose, r1.pose)
