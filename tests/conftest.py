import contextlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build what the tests load.  hipcc cross-compiles without a GPU; on the GPU box the
    # prebuilt .so travels with the snapshot and these are no-ops.
    from backtoreality_amd import build as _build
    import oracle as _oracle
    _build.build()
    _oracle.build()


@contextlib.contextmanager
def use_ext(ext):
    """Temporarily route pointnet2_utils through `ext` (the CPU oracle adapter in CPU tests)."""
    from backtoreality_amd.pointnet2 import pointnet2_utils
    old = pointnet2_utils._ext
    pointnet2_utils._ext = ext
    try:
        yield
    finally:
        pointnet2_utils._ext = old


@pytest.fixture
def oracle_ext():
    import oracle
    with use_ext(oracle.ext_cpu):
        yield oracle.ext_cpu


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=[1, 0, 2], ids=["fmad1", "fmad0", "fmad2"])
def distance_mode(request):
    """Runs a test once per rounding mode of the squared distance (include/btr_pointnet2.h,
    btr_distance_mode): the HIP library of that mode against the oracle of that mode."""
    import oracle
    from backtoreality_amd.pointnet2 import _ext
    _ext.set_fmad(request.param)
    oracle.set_fmad(request.param)
    try:
        yield request.param
    finally:
        _ext.set_fmad(1)
        oracle.set_fmad(1)
