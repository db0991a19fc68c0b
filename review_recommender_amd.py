"""Import shim: ``import review_recommender_amd`` loads the package that lives in
the directory ``review-recommender_amd/`` (a hyphen is not importable)."""
import importlib.util
import pathlib
import sys

_dir = pathlib.Path(__file__).resolve().parent / "review-recommender_amd"
_spec = importlib.util.spec_from_file_location(
    "review_recommender_amd", _dir / "__init__.py",
    submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["review_recommender_amd"] = _mod
_spec.loader.exec_module(_mod)
