import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    # -m gpu tests go through the C ABI on a real MI355X; on a box without a GPU they fail loudly (the HIP
    # path is the product, there is nothing to fall back to).  Everything else runs on CPU.
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_lib():
    from mysteryann_amd import capi

    capi.build_library()
    return capi.lib()
