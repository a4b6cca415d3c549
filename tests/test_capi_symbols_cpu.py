"""CPU: libisr_hip.so loads without a GPU and exports every symbol include/isr_hip.h declares;
argument errors are reported through the C ABI without touching a device."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "isr_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(isr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(hip_lib):
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import _capi
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in isr_hip.h but not exported"
        assert n in _capi.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_capi.SIGNATURES) == names, "ctypes table and header disagree"


def test_no_stub_left():
    assert not (ROOT / "imagesequenceregistrationfor6dposeestimationlabeling_amd" / "csrc" / "stubs_todo.hip").exists()


def test_abi_version_and_errors(hip_lib):
    assert hip_lib.isr_abi_version() == 5
    assert hip_lib.isr_nn_batched_workspace_bytes(0, 10, 1) == 0
    # null clouds -> ISR_ERR_ARG with a message, no device access
    rc = hip_lib.isr_nn_batched(None, 4, None, 4, None, None, 1, -1.0, None, None, None, None, None, None,
                                None, 0, None)
    assert rc == -1 and b"null" in hip_lib.isr_last_error()
    rc = hip_lib.isr_select_top(None, 0, 0.8, 500, None, None, None, None, 0, None)
    assert rc == -1


def test_product_never_imports_oracle():
    pkg = ROOT / "imagesequenceregistrationfor6dposeestimationlabeling_amd"
    for f in pkg.rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f"{f} imports the oracle"
    for f in (pkg / "csrc").glob("*"):
        if f.is_file() and f.suffix in (".hip", ".hpp"):
            assert "#include \"../../oracle" not in f.read_text()


def test_cpu_tensors_fail_loudly(hip_lib):
    import torch
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import ops, registration
    from imagesequenceregistrationfor6dposeestimationlabeling_amd._capi import IsrError
    with pytest.raises(IsrError):
        ops.corr_argmax(torch.zeros(4, 16), torch.zeros(8, 16))
    if not torch.cuda.is_available():
        with pytest.raises(IsrError):
            registration.getCors(torch.zeros(4, 12), torch.zeros(8, 12))
