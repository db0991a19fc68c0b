"""Builds librr_hip.so (hand-written HIP for gfx950) in-tree with hipcc."""
from __future__ import annotations

import os
import pathlib
import shutil
import subprocess

PKG_DIR = pathlib.Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "librr_hip.so"
SOURCES = ["rr_api.hip", "rr_dense.hip", "rr_dense_bf16.hip", "rr_dense_x3.hip", "rr_dense_x3w.hip", "rr_dense_flt.hip", "rr_bm25.hip", "rr_fuse.hip", "rr_reviews.hip"]
# -ffp-contract=off: the BM25 and fusion kernels reproduce numpy's one-rounding-per-
# operation arithmetic; fused multiply-adds are written out (__builtin_fmaf) where wanted.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
         "-ffp-contract=off", "-Wno-unused-value"]


def hipcc_path() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    built = LIB_PATH.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "rr_common.h", CSRC / "rr_dense.h", CSRC / "rr_x3.h",
                                          PKG_DIR.parent / "include" / "rr_hip.h"]
    return any(d.stat().st_mtime > built for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> pathlib.Path:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path(), *FLAGS, *[str(CSRC / s) for s in SOURCES], "-o", str(LIB_PATH)]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
