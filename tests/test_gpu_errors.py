"""GPU: error behaviour of the C ABI — bad requests come back as negative codes with a message in
isr_last_error(), never as a crash or a silent wrong answer (include/isr_hip.h: ISR_ERR_*)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import _capi
    return _capi, _capi.lib()


def test_workspace_too_small_is_reported(cuda0):
    _capi, L = _lib()
    P, N, D = 300, 500, 64
    q = torch.zeros((P, D), dtype=torch.bfloat16, device=cuda0)
    k = torch.zeros((N, D), dtype=torch.bfloat16, device=cuda0)
    idx = torch.empty(P, dtype=torch.int32, device=cuda0)
    logp = torch.empty(P, dtype=torch.float32, device=cuda0)
    ws = torch.empty(64, dtype=torch.uint8, device=cuda0)
    rc = L.isr_corr_argmax(q.data_ptr(), k.data_ptr(), P, N, D, D, D, _capi.DTYPE_BF16, idx.data_ptr(), logp.data_ptr(),
                           None, ws.data_ptr(), ws.numel(), None)
    assert rc == -2 and b"workspace" in L.isr_last_error()
    rc = L.isr_nn_batched(q.data_ptr(), 10, k.data_ptr(), 10, None, None, 1, -1.0, logp.data_ptr(), None, None, None,
                          None, None, ws.data_ptr(), 8, None)
    assert rc == -2


def test_bad_arguments_are_reported(cuda0):
    _capi, L = _lib()
    q = torch.zeros((8, 24), dtype=torch.bfloat16, device=cuda0)
    out = torch.empty(8, dtype=torch.int32, device=cuda0)
    lp = torch.empty(8, dtype=torch.float32, device=cuda0)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=cuda0)
    # bf16 needs D in {16, 32, 64, 128}: the mirror pads, the raw ABI refuses
    rc = L.isr_corr_argmax(q.data_ptr(), q.data_ptr(), 8, 8, 24, 24, 24, _capi.DTYPE_BF16, out.data_ptr(), lp.data_ptr(), None,
                           ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"D=24" in L.isr_last_error()
    rc = L.isr_corr_argmax(None, q.data_ptr(), 8, 8, 16, 16, 16, _capi.DTYPE_BF16, out.data_ptr(), lp.data_ptr(), None,
                           ws.data_ptr(), ws.numel(), None)
    assert rc == -1
    rc = L.isr_corr_argmax(q.data_ptr(), q.data_ptr(), 8, 8, 16, 24, 24, 7, out.data_ptr(), lp.data_ptr(), None,
                           ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"dtype" in L.isr_last_error()
    # the reference indexes an empty sort and raises: so does the filter on P = 0
    rc = L.isr_select_top(lp.data_ptr(), 0, 0.8, 500, out.data_ptr(), out.data_ptr(), None, ws.data_ptr(), ws.numel(), None)
    assert rc == -1
    # ICP needs a positive correspondence radius
    rc = L.isr_icp_point_to_point(lp.data_ptr(), 2, lp.data_ptr(), 2, -1.0, 30, 1e-6, 1e-6, lp.data_ptr(), lp.data_ptr(),
                                  ws.data_ptr(), ws.numel(), None)
    assert rc == -1


def test_lse_only_and_topk_argument_errors(cuda0):
    """Round 4's entries: idx may be NULL only for an lse-only call (logp NULL, lse given); isr_corr_topk refuses k outside
    1 .. 8, D > 128, a missing lse and a short workspace; the mirrors raise ValueError before they call."""
    _capi, L = _lib()
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    P, N, D = 64, 100, 12
    q = torch.randn((P, D), device=cuda0)
    k = torch.randn((N, D), device=cuda0)
    idx = torch.empty((P, 3), dtype=torch.int32, device=cuda0)
    vals = torch.empty((P, 3), dtype=torch.float32, device=cuda0)
    lse = torch.empty(P, dtype=torch.float32, device=cuda0)
    ws = torch.empty(max(L.isr_corr_argmax_workspace_bytes(P, N, D, _capi.DTYPE_F32), L.isr_corr_topk_workspace_bytes(P, N)),
                     dtype=torch.uint8, device=cuda0)
    # idx NULL with logp given, or with nothing to write: refused
    rc = L.isr_corr_argmax(q.data_ptr(), k.data_ptr(), P, N, D, D, D, _capi.DTYPE_F32, None, vals.data_ptr(), lse.data_ptr(),
                           ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"lse-only" in L.isr_last_error()
    rc = L.isr_corr_argmax(q.data_ptr(), k.data_ptr(), P, N, D, D, D, _capi.DTYPE_F32, None, None, None, ws.data_ptr(), ws.numel(), None)
    assert rc == -1
    # the lse-only call itself is fine
    rc = L.isr_corr_argmax(q.data_ptr(), k.data_ptr(), P, N, D, D, D, _capi.DTYPE_F32, None, None, lse.data_ptr(), ws.data_ptr(),
                           ws.numel(), None)
    assert rc == 0
    ref = torch.logsumexp(q.double() @ k.double().T, dim=-1)
    assert float((lse.double() - ref).abs().max()) < 1e-4
    for bad_k in (0, 9):
        rc = L.isr_corr_topk(q.data_ptr(), k.data_ptr(), P, N, D, D, D, bad_k, lse.data_ptr(), idx.data_ptr(), vals.data_ptr(),
                             ws.data_ptr(), ws.numel(), None)
        assert rc == -1 and b"leaves" in L.isr_last_error()
    rc = L.isr_corr_topk(q.data_ptr(), k.data_ptr(), P, N, 129, 129, 129, 3, lse.data_ptr(), idx.data_ptr(), vals.data_ptr(),
                         ws.data_ptr(), ws.numel(), None)
    assert rc == -1
    rc = L.isr_corr_topk(q.data_ptr(), k.data_ptr(), P, N, D, D, D, 3, None, idx.data_ptr(), vals.data_ptr(), ws.data_ptr(),
                         ws.numel(), None)
    assert rc == -1
    rc = L.isr_corr_topk(q.data_ptr(), k.data_ptr(), P, N, D, D, D, 3, lse.data_ptr(), idx.data_ptr(), vals.data_ptr(),
                         ws.data_ptr(), 16, None)
    assert rc == -2
    with pytest.raises(ValueError):
        ops.corr_topk(q, k, 9)
    with pytest.raises(ValueError):
        ops.corr_topk(q, torch.randn((N, D + 1), device=cuda0), 2)
    with pytest.raises(ValueError):
        ops.corr_lse(q[:0], k)


def test_adds_bounds_argument_errors(cuda0):
    import ctypes
    _capi, L = _lib()
    f = torch.zeros(64, dtype=torch.float32, device=cuda0)
    T = torch.zeros(12, dtype=torch.float64, device=cuda0)
    out = torch.empty(2, dtype=torch.float64, device=cuda0)
    g = (ctypes.c_double * 3)(0.0, 0.0, 0.0)
    bb = (ctypes.c_float * 6)(0, 0, 0, 1, 1, 1)
    gp, bp = ctypes.cast(g, ctypes.c_void_p), ctypes.cast(bb, ctypes.c_void_p)
    ok = lambda **kw: L.isr_adds_bounds(kw.get("v", f.data_ptr()), kw.get("V", 4), T.data_ptr(), None, kw.get("B", 1), f.data_ptr(),
                                        kw.get("g", gp), kw.get("h", 0.5), kw.get("nx", 4), 4, 4, bp, out.data_ptr(),
                                        out.data_ptr() + 8, None)
    assert ok() == 0
    assert ok(v=None) == -1 and b"null" in L.isr_last_error()
    assert ok(g=None) == -1
    assert ok(V=0) == -1 and b"V=0" in L.isr_last_error()
    assert ok(B=0) == -1
    assert ok(h=0.0) == -1
    assert ok(nx=0) == -1
    torch.cuda.synchronize()


def test_mirror_raises_on_shape_errors(cuda0):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    with pytest.raises(ValueError):
        ops.corr_argmax(torch.zeros((4, 12), device=cuda0), torch.zeros((5, 13), device=cuda0))
    with pytest.raises(ValueError):
        ops.nn_batched(torch.zeros((4, 2), device=cuda0), torch.zeros((4, 3), device=cuda0))
    with pytest.raises(ValueError):
        ops.prep_queries(torch.zeros((8, 8, 12), device=cuda0), torch.zeros((9, 8), dtype=torch.uint8, device=cuda0))


def test_matrix_free_estimate_pose_entries_validate_their_arguments(cuda0):
    """isr_ep_sample_direct / isr_zbuf_score_direct: a descriptor grid narrower than res * win, more than 128 channels, a
    missing pointer and a short workspace come back as error codes; the host mirror refuses an lse of the wrong length."""
    _capi, L = _lib()
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_est_surf as pes
    res, e, m = 8, 12, 100
    q = torch.zeros((res * res, e), device=cuda0)
    lse = torch.zeros(res * res, device=cuda0)
    keys = torch.zeros((m, e), device=cuda0)
    mp = torch.full((res * res,), 0.5, device=cuda0)
    idx = torch.empty((16, 4), dtype=torch.int64, device=cuda0)
    ws = torch.empty(L.isr_ep_sample_workspace_bytes(res * res, m), dtype=torch.uint8, device=cuda0)
    args = lambda pitch=res, ee=e, win=1, wsb=ws.numel(), qp=q.data_ptr(): (
        qp, lse.data_ptr(), pitch, ee, win, res, mp.data_ptr(), keys.data_ptr(), m, 1.5, 16, 0, idx.data_ptr(), ws.data_ptr(), wsb, None)
    assert L.isr_ep_sample_direct(*args()) == 0
    torch.cuda.synchronize()
    assert int(idx.min()) >= 0 and int(idx.max()) < res * res * m
    assert L.isr_ep_sample_direct(*args(pitch=res - 1)) == -1 and b"g_pitch" in L.isr_last_error()
    assert L.isr_ep_sample_direct(*args(win=2)) == -1                      # an 8-wide grid cannot hold 8 blocks of 2
    assert L.isr_ep_sample_direct(*args(ee=129)) == -1
    assert L.isr_ep_sample_direct(*args(qp=None)) == -1
    assert L.isr_ep_sample_direct(*args(wsb=64)) == -2 and b"workspace" in L.isr_last_error()
    out = torch.empty(3, device=cuda0)
    Rt = torch.zeros((1, 12), device=cuda0)
    K = (_capi.C.c_double * 9)(100, 0, 4, 0, 100, 4, 0, 0, 1)
    pts = torch.zeros((m, 3), device=cuda0)
    rc = L.isr_zbuf_score_direct(pts.data_ptr(), m, Rt.data_ptr(), 1, _capi.C.cast(K, _capi.C.c_void_p), res, mp.data_ptr(),
                                 mp.data_ptr(), q.data_ptr(), lse.data_ptr(), res - 1, e, 1, 1, keys.data_ptr(), out.data_ptr(),
                                 out.data_ptr(), out.data_ptr(), None, 0, None)
    assert rc == -1 and b"g_pitch" in L.isr_last_error()
    with pytest.raises(ValueError):
        pes.DescriptorGrid.pooled(q, keys, res, lse[:-1])
    with pytest.raises(ValueError):
        pes.DescriptorGrid(q, keys, res, res, 2)


def test_estimate_pose_one_call_entry_validates_and_matches_the_mirror(cuda0):
    """isr_estimate_pose through bare ctypes (what another host language would bind): a short workspace and a missing pointer
    are refused; a good call returns the poses and scores the Python mirror returns."""
    import ctypes
    import numpy as np
    _capi, L = _lib()
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import pose_est_surf as pes, synth
    s = synth.crop_scene(seed=3, r=96, e=12, m=3000, f=400.0)
    ml, q = torch.from_numpy(s["mask_lgts"]).to(cuda0), torch.from_numpy(s["query"]).to(cuda0)
    pts, keys = torch.from_numpy(s["pts"]).to(cuda0), torch.from_numpy(s["keys"]).to(cuda0)
    nrm = torch.from_numpy(s["normals"]).double().to(cuda0)              # the ABI takes f64 normals (normals_scaled.npy)
    S, E = 2000, 100
    need = L.isr_estimate_pose_workspace_bytes(96, 12, 3000, 3, S, E, 1)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda0)
    Rt = torch.empty((E, 12), device=cuda0)
    sc = [torch.empty(E, device=cuda0) for _ in range(3)]
    dist = torch.empty(S, device=cuda0)
    u8 = [torch.empty(S, dtype=torch.uint8, device=cuda0) for _ in range(3)]
    K = (ctypes.c_double * 9)(*s["K"].reshape(9).tolist())
    n_poses, n_keep = ctypes.c_int32(-1), ctypes.c_int32(-1)

    def call(ws_bytes=need, mask_ptr=ml.data_ptr()):
        return L.isr_estimate_pose(mask_ptr, q.data_ptr(), 96, 12, pts.data_ptr(), nrm.data_ptr(), keys.data_ptr(), 3000,
                                   float(s["diameter"]), ctypes.cast(K, ctypes.c_void_p), S, E, 3, 1.5, 0.1, 1, 1, 1, 5,
                                   Rt.data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(), sc[2].data_ptr(), dist.data_ptr(),
                                   u8[0].data_ptr(), u8[1].data_ptr(), u8[2].data_ptr(), ctypes.byref(n_poses), ctypes.byref(n_keep),
                                   ws.data_ptr(), ws_bytes, None)
    assert call(ws_bytes=need // 2) == -2 and b"workspace" in L.isr_last_error()
    assert call(mask_ptr=None) == -1
    assert call() == 0
    torch.cuda.synchronize()
    ref = pes.estimate_pose(ml, q, pts, nrm, keys, s["diameter"], s["K"], max_poses=S, max_pose_evaluations=E, seed=5)
    n = n_poses.value
    assert n == ref[0].shape[0] and 0 < n <= E and n_keep.value >= n
    assert torch.equal(Rt[:n].view(n, 3, 4)[:, :, :3], ref[0]) and torch.equal(sc[0][:n], ref[2])
    solved = u8[2].cpu().numpy().astype(bool)
    assert np.array_equal(dist.cpu().numpy()[solved], ref[5])
