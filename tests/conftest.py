import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a rank stuck in a collective must fail its test, not stall the whole run: default per-test timeout when the
    # pytest-timeout plugin is there and the command line set none
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900


@pytest.fixture(scope="session")
def oracle():
    from oracle import gvoracle
    gvoracle.lib()
    return gvoracle
