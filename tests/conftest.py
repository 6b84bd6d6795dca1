import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The host search calls into libasw_hip.so (native hypercube subdivision), so even the CPU
    suite needs the library: build it once when it is absent.  An existing library is never
    rebuilt from inside a test session (it may already be mapped by this process)."""
    from acousticswarms_speech_amd import native
    if not os.path.exists(native.LIB_PATH):
        native.build()
    if not os.path.exists(native.OPS_PATH):
        native.build_torch_ops()
