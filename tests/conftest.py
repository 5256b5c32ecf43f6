import os
import sys

import pytest

try:  # torch (when present) must load its HIP runtime before libpysparse_hip.so does
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suite tests the HIP product: the opt-in host mode (PSP_DEVICE=cpu, psp_cpu.hip) is only ever entered by the
    # child processes of tests/test_cpu_mode.py, never by the test process itself
    if os.environ.get("PSP_DEVICE"):
        raise pytest.UsageError("unset PSP_DEVICE: the tests must run on the HIP path (tests/test_cpu_mode.py starts "
                                "its own PSP_DEVICE=cpu children)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
