"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol
include/rr_hip.h declares, fails loudly without a device, and nothing in the product
package touches the oracle."""
import ctypes
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG = ROOT / "review-recommender_amd"


def declared_symbols():
    header = (ROOT / "include" / "rr_hip.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", header)))


def test_library_exports_every_declared_symbol(hip):
    from review_recommender_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(hip, name), f"librr_hip.so lacks {name}"
    assert set(names) == set(_lib.PROTOTYPES), "ctypes prototypes and header disagree"


def test_fuse_params_layout_matches_header():
    from review_recommender_amd import _lib
    header = (ROOT / "include" / "rr_hip.h").read_text()
    start = header.index("typedef struct rr_fuse_params {") + len("typedef struct rr_fuse_params {")
    body = header[start:header.index("} rr_fuse_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        m = re.match(r"\s*(double|int32_t|int64_t)\s+(.*)", decl.strip(), flags=re.S)
        if m:
            fields += [(n.strip(), m.group(1)) for n in m.group(2).split(",")]
    ctype = {"double": ctypes.c_double, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64}
    assert [(n, ctype[t]) for n, t in fields] == list(_lib.FuseParams._fields_)


def test_no_device_means_a_loud_error_not_a_fallback(hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    from review_recommender_amd import _lib
    from review_recommender_amd.index import ProductIndex
    with pytest.raises(_lib.HipLibraryError):
        ProductIndex(np.zeros((4, 8), dtype=np.float32))
    n = ctypes.c_int(-1)
    assert hip.rr_device_count(ctypes.byref(n)) != 0 and b"hipGetDeviceCount" in hip.rr_last_error()


def test_product_never_imports_the_oracle():
    offenders = []
    for path in list(PKG.rglob("*.py")) + list(PKG.rglob("*.hip")) + list(PKG.rglob("*.h")) + \
            [ROOT / "review_recommender_amd.py"]:
        src = path.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "oracle/" in src and path.suffix == ".py":
            offenders.append(str(path))
    assert not offenders, offenders


def test_bench_uses_the_oracle_only_in_the_cpu_baseline_leg():
    src = (ROOT / "bench.py").read_text()
    uses = [m.start() for m in re.finditer(r"\boracle\b", src)]
    start = src.index("def cpu_baseline(")
    end = src.index("\ndef ", start + 1)
    assert uses and all(start <= u < end for u in uses if "import" in src[max(0, u - 12):u + 8]), \
        "oracle imports must live inside cpu_baseline()"
