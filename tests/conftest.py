import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def conv_math(request):
    """Run a test under the requested conv arithmetic (heatnet_amd.ops.CONV_MATH: "f32" | "bf16x3"), then restore."""
    from heatnet_amd import ops
    prev = ops.set_conv_math(request.param)
    yield request.param
    ops.set_conv_math(prev)


BOTH_MATHS = pytest.mark.parametrize("conv_math", ["f32", "bf16x3"], indirect=True)
