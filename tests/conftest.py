import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")
    config.addinivalue_line("markers", "slow: the BASELINE configs at full size on one MI355X (minutes); opt-in: -m slow on the GPU box "
                                       "(not part of -m gpu; skipped without a GPU)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/libmsfm_oracle.so on first use."""
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library; built (cross-compiled) on demand so CPU-only runs can check the ABI."""
    import subprocess
    from monocularsfm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "monocularsfm_amd", "csrc"), "-s"])
    return _lib.load()


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """Two HIP runtimes share the test process on the GPU box (PyTorch's bundled one, the system one behind libmsfm_match.so), and
    torch's only comes up if it initialises FIRST (monocularsfm_amd/_lib.py, Context.__init__): the tests that use torch tensors
    (tests/test_gpu_exchange.py) must not depend on which test created the first context."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # noqa: BLE001
        pass
    yield


@pytest.fixture(scope="session")
def gpu_ctx(built_lib):
    """One GPU context for the whole session; fails loudly (no skip, no fallback) without a GPU."""
    from monocularsfm_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
