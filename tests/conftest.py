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


def pytest_sessionstart(session):
    """The product refuses a library that was built from other sources than the ones beside it (_lib.load: "stale library").
    A test session on a tree whose sources moved after the last build would fail in every test for that one reason: where
    hipcc is at hand the harness rebuilds first (content-addressed: a no-op on an up-to-date tree) and says so."""
    try:
        from review_recommender_amd import build as B
        for debug in (False, True):
            lib = B.DEBUG_LIB_PATH if debug else B.LIB_PATH
            if (lib.exists() or not debug) and B.needs_build(debug):
                print(f"[conftest] {lib.name} is missing or stale: rebuilding with hipcc", flush=True)
                B.build_library(debug=debug)
    except Exception as e:              # (no hipcc, a compile error: the tests will say what is wrong)
        print(f"[conftest] could not rebuild the HIP library: {e}", flush=True)


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
