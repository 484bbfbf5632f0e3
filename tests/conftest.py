import os
import sys

import pytest

# The library ignores its XSI_* tuning variables and failure-injection hooks unless the process opts in (csrc/xsi_common.hpp);
# the tests force kernels and inject failures through them.
os.environ.setdefault("XSI_ENABLE_TUNING_ENV", "1")
os.environ.setdefault("XSI_ENABLE_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
