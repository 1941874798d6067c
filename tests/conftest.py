import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device AND the built libcvhip.so: skip them (instead of failing every one of them) on a
    CPU-only box, so a plain `pytest tests` still shows real host-test regressions."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    has_lib = os.path.exists(os.path.join(ROOT, "cvpytorch_amd", "libcvhip.so"))
    if has_gpu and has_lib:
        return
    why = "no HIP GPU available" if not has_gpu else "cvpytorch_amd/libcvhip.so not built"
    skip = pytest.mark.skip(reason="gpu test: " + why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
