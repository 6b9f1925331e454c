import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=[16384, 1], ids=["tile-gemm", "panel-gemm"])
def panel_rows(request, monkeypatch):
    """Golden / oracle step tests run twice: with the shipped row gate (small test shapes go to the tile GEMM) and with the gate at
    one row, so that every K <= 256 and K-streamed contraction -- LayerNorm epilogue, keep bits, weight planes -- of the SAME test is
    served by csrc/panel.hip and compared with the reference's fixtures directly (the headline configuration runs those kernels)."""
    from unast_amd import config, ops
    monkeypatch.setattr(config, "PANEL_MIN_ROWS", request.param)
    before = list(ops.PANEL_LAUNCHES)
    yield request.param
    served = [a - b for a, b in zip(ops.PANEL_LAUNCHES, before)]
    if request.param == 1:
        assert served[0] > 0 and served[1] > 0, "the panel kernels did not serve this test: %r" % (served,)
    else:
        assert served == [0, 0], "a small test shape reached the panel kernels through the shipped row gate: %r" % (served,)


@pytest.fixture(autouse=True)
def _fixed_sums_off_after_each_test():
    """config.DETERMINISTIC_SUMS (utils.set_deterministic(..., fixed_sums=True)) selects slower, order-fixed kernels; a test that fails
    while it is on must not leave the rest of the session on them (the parity tests are to pin the kernels the train step runs with)."""
    yield
    if "unast_amd.config" in sys.modules:
        sys.modules["unast_amd.config"].DETERMINISTIC_SUMS = os.environ.get("UNAST_DETERMINISTIC_SUMS", "0") == "1"
