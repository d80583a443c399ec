import os
import sys

import pytest

# torch brings its own copy of the HIP runtime and asks the loader for it under another file name
# than librt_amd.so does; a process that uses both must load torch's first, or it ends up with
# two runtimes of which the second finds no GPU.  Some GPU tests hold device memory in torch
# tensors, so torch is loaded here, before any test module imports raytracing_amd.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
