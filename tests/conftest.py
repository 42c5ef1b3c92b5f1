import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# PyTorch's wheel bundles its own ROCm runtime.  A process that initialises the system runtime first (libvsom_hip.so) and
# imports torch afterwards ends with RCCL bound to a runtime that never came up ("no ROCm-capable device is detected" in
# ncclCommInitAll: tests/test_gpu_group.py after a lazy `import torch` in tests/test_gpu_dist_ranks.py).  The whole suite
# never sees that -- collecting tests/test_dist_gloo.py imports torch before any GPU test runs -- but a selection of test
# files does: import it here, first, whatever is selected.
try:
    import torch  # noqa: F401
except Exception:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import vsom_amd
        return vsom_amd.capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _have_gpu()


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with -m gpu; when selected they must not silently skip
    pass
