import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch brings its own copy of the HIP runtime; libcuddle_hip.so is linked against the system's. Whichever is loaded
# first serves both (same soname), and torch only finds the GPU through its own: load it before any test can load
# the library (a test file that imported torch inside a test, after the library, got "no ROCm-capable device").
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        from vectordb_amd import _ffi
        return _ffi.lib().cuddle_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_available():
        pytest.fail("this test needs a GPU and the built libcuddle_hip.so (no CPU fallback exists)")
    return 0
