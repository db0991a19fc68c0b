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
    """Every NAME bench.py binds from oracle/ (import, from-import, importlib, attribute access) lives inside
    cpu_baseline() or its helper _csr_oracle(), and the only callers of those two are cpu_baseline() and the N == 1
    reporting branch of main()."""
    import ast
    src = (ROOT / "bench.py").read_text()
    tree = ast.parse(src)
    allowed = {"cpu_baseline", "_csr_oracle"}
    spans = {n.name: (n.lineno, n.end_lineno) for n in tree.body if isinstance(n, ast.FunctionDef)}
    inside = lambda line: any(spans[f][0] <= line <= spans[f][1] for f in allowed)
    touches = []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
            touches.append(node.lineno)
        elif isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
            touches.append(node.lineno)
        elif isinstance(node, ast.Name) and node.id == "oracle":
            touches.append(node.lineno)
        elif isinstance(node, ast.Constant) and isinstance(node.value, str) and re.fullmatch(r"oracle(\.\w+)*", node.value):
            touches.append(node.lineno)                  # importlib.import_module("oracle...") / __import__
    assert touches and all(inside(l) for l in touches), [l for l in touches if not inside(l)]
    callers = {}
    for fn in (n for n in tree.body if isinstance(n, ast.FunctionDef)):
        for node in ast.walk(fn):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in allowed:
                callers.setdefault(node.func.id, set()).add(fn.name)
    assert callers.get("cpu_baseline") == {"main"} and callers.get("_csr_oracle") <= {"cpu_baseline"}, callers


def test_a_stale_library_is_refused_with_a_rebuild_message(tmp_path, monkeypatch):
    """_lib.load() compares librr_hip.so.buildid with the digest of the sources beside it: a library built from other
    sources says "stale library: ... rebuild" instead of failing later with an AttributeError on a missing symbol."""
    from review_recommender_amd import _lib, build
    assert build.BUILD_ID.read_text().strip() == build.library_digest()      # the shipped pair is consistent
    stale = tmp_path / "librr_hip.so.buildid"
    stale.write_text("0" * 64)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "BUILD_ID", stale)
    with pytest.raises(_lib.HipLibraryError, match="stale library.*rebuild"):
        _lib.load()
    monkeypatch.setattr(_lib, "BUILD_ID", tmp_path / "absent.buildid")
    with pytest.raises(_lib.HipLibraryError, match="stale library.*buildid missing"):
        _lib.load()


def test_committed_generated_loop_equals_the_generator_output():
    """csrc/rr_fltq_loop.inc is generator output kept in history: an edit of gen_fltq_loop.py that is not followed by a
    deliberate regeneration must fail here (and in build.check_generated), not silently change the shipped kernel."""
    import subprocess
    import sys
    csrc = PKG / "csrc"
    gen = subprocess.run([sys.executable, str(csrc / "gen_fltq_loop.py")], capture_output=True, text=True, check=True)
    assert gen.stdout == (csrc / "rr_fltq_loop.inc").read_text()


def test_cli_has_no_hub_backend_unless_asked(monkeypatch):
    """cli.py: sentence-transformers by model NAME is an explicit --allow-hub opt-in, not a fallback of the product path."""
    import sys
    from review_recommender_amd import cli
    args = cli.parse_args(["-q", "x"])
    assert args.allow_hub is False

    class Boom:
        def __getattr__(self, name):
            raise AssertionError("sentence_transformers touched without --allow-hub")
    monkeypatch.setitem(sys.modules, "sentence_transformers", Boom())
    assert cli._load_encoders(args) == (None, None)
