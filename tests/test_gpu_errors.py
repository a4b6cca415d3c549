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


def test_mirror_raises_on_shape_errors(cuda0):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops
    with pytest.raises(ValueError):
        ops.corr_argmax(torch.zeros((4, 12), device=cuda0), torch.zeros((5, 13), device=cuda0))
    with pytest.raises(ValueError):
        ops.nn_batched(torch.zeros((4, 2), device=cuda0), torch.zeros((4, 3), device=cuda0))
    with pytest.raises(ValueError):
        ops.prep_queries(torch.zeros((8, 8, 12), device=cuda0), torch.zeros((9, 8), dtype=torch.uint8, device=cuda0))
