import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def dev_env(monkeypatch, **kv):
    """The library's diagnostic / tuning knobs are keys of ONE variable, IDELUCS_DEV="name=value,..." (csrc/dev_env.h; read at every use): set or (value None) drop
    some of them for the rest of the test."""
    import os
    cur = dict(item.split("=", 1) for item in os.environ.get("IDELUCS_DEV", "").split(",") if "=" in item)
    for k, v in kv.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    monkeypatch.setenv("IDELUCS_DEV", ",".join(f"{k}={v}" for k, v in cur.items()))
