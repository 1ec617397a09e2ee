import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(autouse=True)
def _order_every_level(monkeypatch):
    """Map ordering (csrc/pp_maporder.hip) is skipped for levels below MAP_ORDER_MIN_ROWS rows in production (no gain);
    the tests run on small clouds, so lower the threshold and exercise the ordered path at every level."""
    import sys
    me = sys.modules.get("panopticsegforlargescalepointcloud_amd.MinkowskiEngine")
    if me is None:
        try:
            from panopticsegforlargescalepointcloud_amd import MinkowskiEngine as me
        except Exception:
            me = None
    if me is not None:
        monkeypatch.setattr(me, "MAP_ORDER_MIN_ROWS", 2)
    yield
