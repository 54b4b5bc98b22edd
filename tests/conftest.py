import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# a wait for the GPU that never ends becomes an error of the call (csrc/api.hip), and a test that hangs anyway ends the run after
# 15 minutes instead of occupying the GPU box until someone's outer limit
os.environ.setdefault("NSGPU_WAIT_TIMEOUT_S", "120")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.pluginmanager.hasplugin("timeout") and not config.getoption("timeout", None):
        config.option.timeout = 900
        config.option.timeout_method = "thread"


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
