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


def pytest_collection_modifyitems(config, items):
    # the run that needs most of the GPU's memory (cfg4's per-GPU share with the consensus graphs in HBM: 190 GB mapped at its peak) goes first,
    # while nothing else of the session holds HBM
    items.sort(key=lambda it: 0 if "cfg4_per_gpu_share" in it.nodeid and "[device]" in it.nodeid else 1)


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
