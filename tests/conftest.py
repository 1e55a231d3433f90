import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """libaether_hip.so built in-tree (fails loudly if absent: the GPU tests must exercise native code)."""
    from aether_amd import _lib
    from aether_amd.build import build_native

    if not _lib.lib_path().exists():
        build_native()
    return _lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from aether_amd import _lib

    _lib.check(_lib.load().aether_check_device(), "aether_check_device")
    return torch.device("cuda:0")
