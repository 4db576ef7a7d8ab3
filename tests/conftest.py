import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from mmpl_amd import _lib
    return _lib.load()


@pytest.fixture(autouse=True)
def _release_gpu_objects_between_tests(request):
    """GPU tests build pipelines whose hipGraphs, side streams and library handles are released by garbage collection -- which would
    otherwise strike at an arbitrary point of a LATER test (graph / handle destruction implies device-wide frees).  Collect at the test
    boundary instead, with the device idle."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
