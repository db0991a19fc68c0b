"""Device-resident product index: the embedding matrix and the per-row metadata
the fusion kernel needs (product_emb.npy + product_emb_meta.parquet)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

MAX_POOL = 2048     # RR_MAX_POOL
MAX_BATCH = 1024    # RR_MAX_BATCH
DTYPES = {"f32": 0, "bf16": 1}   # RR_DTYPE_*


class ProductIndex:
    """``n_rows x dim`` float32 matrix on one GPU (one shard of the corpus).

    ``matrix`` may be a numpy array (copied to the device), or ``None`` with
    ``device_ptr`` naming caller-owned device memory that is already padded to
    ``dim_padded`` columns (used by bench.py, which generates 10M rows on the GPU).
    """

    def __init__(self, matrix: Optional[np.ndarray] = None, *, n_rows: Optional[int] = None,
                 dim: Optional[int] = None, device: int = 0, row_offset: int = 0,
                 device_ptr: Optional[int] = None, normalize: bool = False, eps: float = 1e-12,
                 keepalive=None, dtype: str = "f32"):
        lib = _lib.load()
        if dtype not in DTYPES:
            raise ValueError(f"dtype must be one of {sorted(DTYPES)}")
        self.dtype = dtype
        if matrix is not None and dtype != "f32":
            raise ValueError("pass fp32 rows through ProductIndex.from_rows(..., dtype='bf16'): "
                             "they are rounded once, after the fp32 normalisation")
        if matrix is not None:
            matrix = np.ascontiguousarray(matrix, dtype=np.float32)
            if matrix.ndim != 2:
                raise ValueError("embedding matrix must be 2-D (rows x dim)")
            n_rows, dim = matrix.shape
        if not n_rows or not dim:
            raise ValueError("an index needs at least one row and one column")
        self.n_rows, self.dim, self.device, self.row_offset = int(n_rows), int(dim), device, row_offset
        h = C.c_void_p()
        _lib.check(lib.rr_index_create(_lib.ptr(matrix), self.n_rows, self.dim, DTYPES[dtype], device,
                                       row_offset, C.byref(h)), "rr_index_create")
        self._h = h
        self._keepalive = keepalive
        padded = C.c_int32()
        _lib.check(lib.rr_index_dim_padded(h, C.byref(padded)))
        self.dim_padded = padded.value
        if device_ptr is not None:
            _lib.check(lib.rr_index_adopt_device(h, C.c_void_p(device_ptr)), "rr_index_adopt_device")
        if normalize:
            self.l2_normalize(eps)
        self.has_meta = False

    @classmethod
    def from_rows(cls, rows, dim: Optional[int] = None, chunk_rows: int = 262_144, **kw) -> "ProductIndex":
        """Builds the index from any (N, dim) array-like (e.g. a memory-mapped product_emb.npy),
        uploading ``chunk_rows`` rows at a time so the host never holds a second copy."""
        n = rows.shape[0]
        dim = dim or rows.shape[1]
        normalize = kw.pop("normalize", False)
        eps = kw.pop("eps", 1e-12)
        ix = cls(None, n_rows=n, dim=dim, **kw)
        lib = _lib.load()
        for s in range(0, n, chunk_rows):
            part = np.ascontiguousarray(rows[s:s + chunk_rows], dtype=np.float32)
            # fp32 rows in; normalised in fp32 on the device if asked; a bf16 index rounds once, afterwards
            _lib.check(lib.rr_index_upload_rows_f32(ix._h, s, part.shape[0], _lib.ptr(part),
                                                    eps if normalize else 0.0), "rr_index_upload_rows_f32")
        return ix

    @property
    def handle(self):
        return self._h

    def l2_normalize(self, eps: float = 1e-12) -> None:
        """l2_normalize (utils.py:40-44) of every row, on the device, in place."""
        _lib.check(_lib.load().rr_index_l2_normalize(self._h, eps), "rr_index_l2_normalize")

    def set_meta(self, n_reviews: np.ndarray, avg_stars: np.ndarray) -> None:
        """Row-aligned ``n_reviews`` / ``avg_stars`` as run_search derives them
        (app/app_product_search.py:264-265: to_numeric, n NaN -> 0, stars NaN kept)."""
        n = np.ascontiguousarray(n_reviews, dtype=np.float64)
        r = np.ascontiguousarray(avg_stars, dtype=np.float64)
        if n.shape != (self.n_rows,) or r.shape != (self.n_rows,):
            raise ValueError(f"metadata must have {self.n_rows} rows "
                             f"(got {n.shape} and {r.shape})")
        l1p = np.ascontiguousarray(np.log1p(n))
        _lib.check(_lib.load().rr_index_set_meta(self._h, _lib.ptr(n), _lib.ptr(r), _lib.ptr(l1p)),
                   "rr_index_set_meta")
        self.has_meta = True

    def dense_topk(self, queries: np.ndarray, pool: int) -> Tuple[np.ndarray, np.ndarray]:
        """Batched cosine_similarity_search: (n_queries, pool') rows and scores,
        pool' = min(pool, n_rows), each row ordered (score desc, row asc)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"queries must be (n, {self.dim}); got {q.shape}")
        if pool < 0:
            raise ValueError("pool must be >= 0")
        eff = min(int(pool), self.n_rows)
        rows = np.empty((q.shape[0], eff), dtype=np.int64)
        scores = np.empty((q.shape[0], eff), dtype=np.float32)
        got = C.c_int32()
        lib = _lib.load()
        for s in range(0, q.shape[0], MAX_BATCH):
            part = q[s:s + MAX_BATCH]
            r = np.empty((part.shape[0], eff), dtype=np.int64)
            v = np.empty((part.shape[0], eff), dtype=np.float32)
            _lib.check(lib.rr_dense_topk(self._h, _lib.ptr(part), part.shape[0], int(pool),
                                         _lib.ptr(r) if eff else None,
                                         _lib.ptr(v) if eff else None, C.byref(got)),
                       "rr_dense_topk")
            rows[s:s + MAX_BATCH] = r
            scores[s:s + MAX_BATCH] = v
        return rows, scores

    def last_scan_ms(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.load().rr_index_last_scan_ms(self._h, C.byref(ms)), "rr_index_last_scan_ms")
        return ms.value

    def matrix_changed(self) -> None:
        """After writing to a matrix the index ADOPTED (``device_ptr=``) behind its back: drops what the index derived
        from the old content (row-norm bounds of the filter scan, the bf16 filter plane); rebuilt at the next batched
        search.  Writes made through the index itself (uploads, l2_normalize) do this on their own."""
        _lib.check(_lib.load().rr_index_matrix_changed(self._h), "rr_index_matrix_changed")

    def last_scan_info(self):
        """(kernel family, variant, queries per launch, MFMA terms, stream element bytes) of the last batched scan launch
        (rr_index_last_scan_info; family 5 = filter scan, variant 9 = the 256-query query-stationary rr_scan_fltq)."""
        out = (C.c_int32 * 8)()
        _lib.check(_lib.load().rr_index_last_scan_info(self._h, out), "rr_index_last_scan_info")
        return tuple(out[:5])

    def select_trace(self):
        """(fast_path_taken, groups_opened, tiles_opened, candidate_rows, then cycle counts of
        the selection's phases) for the first query of the last selection."""
        out = (C.c_int32 * 16)()
        _lib.check(_lib.load().rr_index_select_trace(self._h, out), "rr_index_select_trace")
        return tuple(out)

    def set_scan_mode(self, stored: bool) -> None:
        """Diagnostic: True forces the batched scan's single pass that stores every score
        (RR_SCAN_MODE_STORED); False restores the two-pass default."""
        _lib.check(_lib.load().rr_index_set_scan_mode(self._h, 1 if stored else 0), "rr_index_set_scan_mode")

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().rr_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
