import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The compiled C oracle (test infrastructure)."""
    from oracle import cbind
    cbind.build()
    return cbind


@pytest.fixture(scope="session")
def hip_lib():
    """libisr_hip.so, built in-tree if missing.  GPU tests fail loudly when it cannot load."""
    from imagesequenceregistrationfor6dposeestimationlabeling_amd import _capi, build
    if not _capi.LIB_PATH.exists():
        build.build_hip()
    return _capi.lib()


@pytest.fixture(scope="session")
def cuda0(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    return torch.device("cuda:0")
