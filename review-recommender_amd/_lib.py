"""ctypes binding of librr_hip.so (include/rr_hip.h).  No CPU fallback: if the
library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os
import threading

from .build import DEBUG_BUILD_ID, BUILD_ID, DEBUG_LIB_PATH, LIB_PATH, library_digest

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p
P = C.POINTER


class CopySeg(C.Structure):
    """struct rr_copy_seg (include/rr_hip.h)."""
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("row_bytes", C.c_int64), ("rows", C.c_int64),
                ("dst_pitch", C.c_int64), ("src_pitch", C.c_int64)]


class FuseParams(C.Structure):
    """struct rr_fuse_params (include/rr_hip.h)."""
    _fields_ = [("w_dense", c_f64), ("w_bm25", c_f64), ("w_rerank", c_f64),
                ("w_prior", c_f64), ("w_best", c_f64), ("prior_C", c_f64),
                ("min_reviews", c_i32), ("trust_sat", c_i32), ("apply_trust", c_i32),
                ("rerank_active", c_i32), ("rerank_k", c_i32), ("k", c_i32),
                ("n_candidates", c_i32), ("pool", c_i32), ("cand_per_rank", c_i32),
                ("bm25_f64", c_i32), ("cand_rank_stride_bytes", c_i64)]


class CEConfig(C.Structure):
    """struct rr_ce_config (include/rr_hip.h)."""
    _fields_ = [("hidden", c_i32), ("n_layers", c_i32), ("n_heads", c_i32), ("ffn", c_i32),
                ("vocab", c_i32), ("max_pos", c_i32), ("type_vocab", c_i32), ("n_labels", c_i32),
                ("ln_eps", c_f32), ("precision", c_i32)]


# name -> (restype, argtypes); every symbol include/rr_hip.h declares
PROTOTYPES = {
    "rr_last_error": (C.c_char_p, []),
    "rr_version": (C.c_int, []),
    "rr_device_count": (C.c_int, [P(C.c_int)]),
    "rr_index_create": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i64, P(c_vp)]),
    "rr_index_upload_rows": (C.c_int, [c_vp, c_i64, c_i64, c_vp]),
    "rr_index_upload_rows_f32": (C.c_int, [c_vp, c_i64, c_i64, c_vp, c_f32]),
    "rr_index_adopt_device": (C.c_int, [c_vp, c_vp]),
    "rr_index_dim_padded": (C.c_int, [c_vp, P(c_i32)]),
    "rr_index_l2_normalize": (C.c_int, [c_vp, c_f32]),
    "rr_index_set_meta": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "rr_index_destroy": (C.c_int, [c_vp]),
    "rr_dense_topk": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, P(c_i32)]),
    "rr_dense_topk_dev": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rr_index_last_scan_ms": (C.c_int, [c_vp, P(c_f32)]),
    "rr_index_scan_stats": (C.c_int, [c_vp, P(c_f64), P(c_i64)]),
    "rr_index_last_scan_info": (C.c_int, [c_vp, P(c_i32)]),
    "rr_index_select_trace": (C.c_int, [c_vp, P(c_i32)]),
    "rr_index_set_shadow": (C.c_int, [c_vp, c_i32]),
    "rr_index_set_scan_mode": (C.c_int, [c_vp, c_i32]),
    "rr_dense_scan_dev": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, P(c_i32), c_vp]),
    "rr_dense_select_dev": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rr_dense_scan_slot_dev": (C.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, P(c_i32), c_vp]),
    "rr_dense_select_slot_dev": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rr_dense_topk_slot_dev": (C.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rr_dense_select_part_dev": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rr_stream_create_cu_range": (C.c_int, [c_i32, c_i32, c_i32, P(c_vp)]),
    "rr_stream_destroy": (C.c_int, [c_vp]),
    "rr_index_set_scan_cus": (C.c_int, [c_vp, c_i32]),
    "rr_bm25_create": (C.c_int, [c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                 c_vp, c_vp, c_vp, c_f64, c_f64, c_f64, c_i64, P(c_vp)]),
    "rr_bm25_create_dev": (C.c_int, [c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_f64, c_f64, c_f64, c_i64, P(c_vp)]),
    "rr_bm25_destroy": (C.c_int, [c_vp]),
    "rr_bm25_get_scores": (C.c_int, [c_vp, c_vp, c_i32, c_vp]),
    "rr_bm25_scores_at": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp]),
    "rr_bm25_scores_at_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "rr_fuse_topk_dev": (C.c_int, [c_vp, P(FuseParams), c_i32] + [c_vp] * 9 + [c_vp] * 3 + [c_vp]),
    "rr_fuse_topk": (C.c_int, [c_vp, P(FuseParams), c_i32] + [c_vp] * 9 + [c_vp] * 3),
    "rr_index_gather_meta_dev": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "rr_copy_segments_dev": (C.c_int, [c_vp, c_i32, c_i32, c_vp]),
    "rr_index_matrix_changed": (C.c_int, [c_vp]),
    "rr_reviews_create": (C.c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_i32, c_f32, P(c_vp)]),
    "rr_reviews_destroy": (C.c_int, [c_vp]),
    "rr_reviews_best_dev": (C.c_int, [c_vp, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "rr_reviews_best_cut_dev": (C.c_int, [c_vp, c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "rr_ce_create": (C.c_int, [c_i32, P(CEConfig), P(c_vp), c_i32, P(c_vp)]),
    "rr_ce_destroy": (C.c_int, [c_vp]),
    "rr_ce_forward_dev": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "rr_ce_last_forward_ms": (C.c_int, [c_vp, P(c_f32)]),
    "rr_ce_range_status": (C.c_int, [c_vp, P(C.c_int32)]),
    "rr_ce_set_wide_range": (C.c_int, [c_vp, C.c_int32]),
    "rr_index_stream": (C.c_int, [c_vp, P(c_vp)]),
    "rr_index_synchronize": (C.c_int, [c_vp]),
}

# csrc/rr_debug.h: only in librr_hip_dbg.so (RR_DEBUG_HARNESS=1 in the environment; tools/ only, never the product)
DEBUG_PROTOTYPES = {
    "rr_debug_scan_x3w": (C.c_int, [c_vp, c_i32, c_i32, P(c_f32)]),
    "rr_debug_scan_flt": (C.c_int, [c_vp, c_i32, c_i32, P(c_f32)]),
    "rr_debug_fltq_compare": (C.c_int, [c_vp, P(c_i64)]),
    "rr_debug_ce_ffn_stamps": (C.c_int, [P(C.c_uint64)]),
    "rr_debug_ce_h2_stamps": (C.c_int, [P(C.c_uint64)]),
    "rr_debug_ce_h2_gemm": (C.c_int, [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, P(C.c_int32)]),
}

_lib = None
_lock = threading.Lock()


class HipLibraryError(RuntimeError):
    pass


def load():
    """Loads librr_hip.so once and binds every prototype.  Raises if it is absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        debug = os.environ.get("RR_DEBUG_HARNESS") == "1"
        path = DEBUG_LIB_PATH if debug else LIB_PATH
        if not path.exists():
            raise HipLibraryError(
                f"{path} is missing: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _refuse_stale(path, DEBUG_BUILD_ID if debug else BUILD_ID, debug)
        # torch ships its own HIP runtime (libamdhip64, SONAME .7).  Importing it first makes
        # the dynamic linker bind librr_hip.so to that same runtime; loaded the other way
        # round the process would hold two runtimes and the second one sees no device.
        import torch  # noqa: F401
        lib = C.CDLL(str(path))
        protos = dict(PROTOTYPES, **DEBUG_PROTOTYPES) if debug else PROTOTYPES
        for name, (res, args) in protos.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise HipLibraryError(f"stale library: {path.name} lacks the declared symbol {name}; rebuild it with "
                                      "`python __graft_entry__.py`") from None
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _refuse_stale(path, build_id, debug: bool) -> None:
    """A library built from other sources than the ones beside it (csrc/, include/rr_hip.h, the flags) must not be
    bound against today's prototypes: its buildid file holds the digest of what it was built from."""
    have = build_id.read_text().strip() if build_id.exists() else None
    if have != library_digest(debug):
        raise HipLibraryError(
            f"stale library: {path.name} was built from other sources than the ones in csrc/ "
            f"(buildid {'missing' if have is None else have[:12]} != {library_digest(debug)[:12]}); "
            "rebuild it with `python __graft_entry__.py`")


def check(rc: int, what: str = "") -> None:
    """Maps a non-zero status to ValueError (bad argument) or RuntimeError."""
    if rc == 0:
        return
    msg = load().rr_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(f"{what}: {msg}" if what else msg)
    raise HipLibraryError(f"{what}: {msg} (status {rc})" if what else f"{msg} (status {rc})")


def ptr(a):
    """Address of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array passed to the HIP library must be C-contiguous")
    return a.ctypes.data_as(c_vp)
