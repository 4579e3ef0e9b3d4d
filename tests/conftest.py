import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hiplib():
    # torch ships its own HIP runtime: when it initialises AFTER libovplane_hip.so has brought up the system one, it no longer
    # finds a device (seen with `-k` selections that reach the RCCL tests first through this fixture).  Bring torch up first.
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from ov_plane_amd.build import build_lib

    build_lib()
    from ov_plane_amd import capi

    return capi
