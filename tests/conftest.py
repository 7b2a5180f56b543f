import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _reset_global_state():
    """Forward context and TP state are process globals; keep tests independent."""
    yield
    from minivllm_amd.utils.context import reset_context
    from minivllm_amd.parallel import group as pg
    reset_context()
    if not pg.is_initialized():
        pg.set_tp_for_testing(0, 1)
