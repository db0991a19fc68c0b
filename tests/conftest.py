import json
import pathlib
import sys

try:                     # parquet engine first: on the GPU box pandas could not import it once torch's
    import pyarrow       # distributed stack had been loaded by test collection  # noqa: F401
except ImportError:
    pass

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def unarr(d):
    data = [np.nan if v is None else v for v in d["data"]]
    return np.array(data, dtype=d["dtype"]).reshape(d["shape"])


@pytest.fixture(scope="session")
def golden_primitives():
    return json.loads((GOLDEN / "primitives.json").read_text())


@pytest.fixture(scope="session")
def golden_dense():
    return np.load(GOLDEN / "dense_10k.npz")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library; GPU tests call through it."""
    from review_recommender_amd import _lib
    return _lib.load()
