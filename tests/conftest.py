import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def dev_lib():
    """The DEVELOPMENT build of the library (libcgan_hip_dev.so: the same sources with -DCGAN_DEV) for the duration of a
    test: the tests that run every kernel variant on the same cases force kernels through its ``cgan_debug_set_*`` knobs; the
    product library (what every other test runs on) exports none of them."""
    from climategan_amd import _lib

    lib = _lib.load_dev()
    yield lib
    _lib.use_product()
