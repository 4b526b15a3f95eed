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


@pytest.fixture(autouse=True)
def product_defaults(monkeypatch):
    """Every test runs the library as a user gets it: no environment switches (MPX_DETERMINISTIC would make mpx_create
    select the lane-mode fit kernel; the parity tests must exercise the default, cooperative end game included)."""
    for k in ("MPX_DETERMINISTIC", "MPX_FIT_NOPARK", "MPX_FIT_PARK_NFEV", "MPX_FIT_MAXFEV", "MPX_SACF_PAIR", "MPX_SACF_ABLATE", "MPX_HE_WG"):
        monkeypatch.delenv(k, raising=False)
