"""Builds librr_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

Every source is compiled to its own object (in parallel) and the objects are linked into one shared
library.  What decides whether anything is rebuilt is CONTENT, not mtime: the sha256 of a source, of
every header and of the flags is kept beside its object, and the digest of all of them beside the
library (`librr_hip.so.buildid`), so a stale library cannot hide behind a fresh timestamp and a fresh
checkout with a prebuilt library (the GPU box) does not rebuild.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import pathlib
import shutil
import subprocess

PKG_DIR = pathlib.Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = PKG_DIR / "build"
LIB_PATH = PKG_DIR / "librr_hip.so"
BUILD_ID = PKG_DIR / "librr_hip.so.buildid"
# the ablation harness (csrc/rr_debug.h) lives in a library of its own, built on demand for tools/ only
DEBUG_OBJ_DIR = PKG_DIR / "build_dbg"
DEBUG_LIB_PATH = PKG_DIR / "librr_hip_dbg.so"
DEBUG_BUILD_ID = PKG_DIR / "librr_hip_dbg.so.buildid"
DEBUG_FLAGS = ["-DRR_DEBUG_HARNESS"]
SOURCES = ["rr_api.hip", "rr_dense.hip", "rr_dense_bf16.hip", "rr_dense_x3.hip", "rr_dense_x3w.hip",
           "rr_dense_flt.hip", "rr_bm25.hip", "rr_fuse.hip", "rr_reviews.hip", "rr_ce.hip", "rr_ce_h2.hip"]
# -ffp-contract=off: the BM25 and fusion kernels reproduce numpy's one-rounding-per-
# operation arithmetic; fused multiply-adds are written out (__builtin_fmaf) where wanted.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]


def hipcc_path() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _headers(debug: bool = False):
    incs = [p for p in sorted(CSRC.glob("*.inc")) if debug or p.name != "rr_fltq_loop_abl.inc"]
    return sorted(CSRC.glob("*.h")) + incs + [PKG_DIR.parent / "include" / "rr_hip.h"]


FLTQ_ABLATIONS = "128,1,2,3,4,8,64,66,1025,1026,1027,1028,1029,1030"      # tools/fltq_ablate.py variants 3000 + n (rr_dense_flt.hip: RR_FLTQA_CASE)


def check_generated(debug: bool = False, regenerate: bool = False) -> None:
    """csrc/rr_fltq_loop.inc is generated (gen_fltq_loop.py) and COMMITTED: a copy that differs from what the generator
    prints must not build -- this raises instead of rewriting the shipped hand-scheduled loop behind the reader's back
    (`python -m review_recommender_amd.build --regenerate`, or RR_REGENERATE=1, writes the file on purpose).  The debug
    harness's timing ablations of that loop (rr_fltq_loop_abl.inc) are generated on demand and not committed."""
    gen = subprocess.run([os.sys.executable, str(CSRC / "gen_fltq_loop.py")], capture_output=True, text=True, check=True)
    inc = CSRC / "rr_fltq_loop.inc"
    if not inc.exists() or inc.read_text() != gen.stdout:
        if regenerate or os.environ.get("RR_REGENERATE") == "1":
            inc.write_text(gen.stdout)
        else:
            raise RuntimeError(f"{inc} differs from the output of gen_fltq_loop.py: regenerate it on purpose with "
                               "`python -m review_recommender_amd.build --regenerate` and commit the result")
    if debug:
        gen = subprocess.run([os.sys.executable, str(CSRC / "gen_fltq_loop.py"), "--abl", FLTQ_ABLATIONS],
                             capture_output=True, text=True, check=True)
        inc = CSRC / "rr_fltq_loop_abl.inc"
        if not inc.exists() or inc.read_text() != gen.stdout:
            inc.write_text(gen.stdout)


def _digest(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _flags(debug: bool):
    return FLAGS + (DEBUG_FLAGS if debug else [])


def _source_digest(src: str, debug: bool = False) -> str:
    return _digest([CSRC / src] + _headers(debug), " ".join(_flags(debug)))


def library_digest(debug: bool = False) -> str:
    return _digest([CSRC / s for s in SOURCES] + _headers(debug), " ".join(_flags(debug)))


def needs_build(debug: bool = False) -> bool:
    lib, bid = (DEBUG_LIB_PATH, DEBUG_BUILD_ID) if debug else (LIB_PATH, BUILD_ID)
    if not lib.exists() or not bid.exists():
        return True
    return bid.read_text().strip() != library_digest(debug)


def _compile_one(src: str, verbose: bool, debug: bool = False) -> pathlib.Path:
    obj_dir = DEBUG_OBJ_DIR if debug else OBJ_DIR
    obj = obj_dir / (src + ".o")
    tag = obj_dir / (src + ".sha256")
    want = _source_digest(src, debug)
    if obj.exists() and tag.exists() and tag.read_text().strip() == want:
        return obj
    cmd = [hipcc_path(), *_flags(debug), "-c", str(CSRC / src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n" + proc.stdout + proc.stderr)
    tag.write_text(want)
    return obj


def build_library(force: bool = False, verbose: bool = False, jobs: int = 0, debug: bool = False) -> pathlib.Path:
    """debug=True: librr_hip_dbg.so = the same sources + the ablation harness (-DRR_DEBUG_HARNESS), for tools/ only."""
    lib_path, build_id, obj_dir = ((DEBUG_LIB_PATH, DEBUG_BUILD_ID, DEBUG_OBJ_DIR) if debug
                                   else (LIB_PATH, BUILD_ID, OBJ_DIR))
    check_generated(debug)
    if not force and not needs_build(debug):
        return lib_path
    obj_dir.mkdir(exist_ok=True)
    if force:
        for f in obj_dir.glob("*.sha256"):
            f.unlink()
    jobs = jobs or min(len(SOURCES), max(1, (os.cpu_count() or 2) - 1), 8)
    with cf.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, verbose, debug), SOURCES))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(lib_path)]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + proc.stdout + proc.stderr)
    build_id.write_text(library_digest(debug))
    return lib_path


if __name__ == "__main__":
    import sys
    if "--regenerate" in sys.argv:
        check_generated(regenerate=True)
    print(build_library(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
