"""Best review per candidate product: the device-resident replacement of
``_best_snippets`` (app/app_product_search.py:320-370) / ``best_review_snippets``
(app/test.py:181-215).

The reference re-reads reviews_with_embeddings.parquet on every query, keeps the reviews whose sku
is among the candidates, cuts them to ``max_rows`` in file order, l2-normalises their embeddings,
scores them against the query and keeps the best review per sku.  Here the review embeddings are
uploaded and normalised once, grouped by product row, and a query only touches the reviews of its
candidates (csrc/rr_reviews.hip).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Sequence

import numpy as np
import pandas as pd

from . import _lib


class ReviewIndex:
    def __init__(self, reviews: pd.DataFrame, embeddings: np.ndarray, product_skus: Sequence[str],
                 device: int = 0, eps: float = 1e-12):
        """reviews: frame with ``sku``, ``text`` and optionally ``stars`` (file order = row order);
        embeddings: (n_reviews, dim) float32, row-aligned; product_skus: the metadata's sku column."""
        if "sku" not in reviews.columns:
            raise ValueError("review table lacks a 'sku' column")      # app/app_product_search.py:328-330
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        if emb.ndim != 2 or emb.shape[0] != len(reviews):
            raise ValueError("embeddings must be (n_reviews, dim), row-aligned with the review table")
        self.texts = reviews["text"].astype(str).tolist() if "text" in reviews.columns else [""] * len(reviews)
        self.stars = (pd.to_numeric(reviews["stars"], errors="coerce").to_numpy(dtype=np.float64)
                      if "stars" in reviews.columns else np.full(len(reviews), np.nan))
        row_of = {str(s): i for i, s in enumerate(product_skus)}
        prod = np.array([row_of.get(str(s), -1) for s in reviews["sku"].tolist()], dtype=np.int64)
        known = np.nonzero(prod >= 0)[0]
        order = known[np.argsort(prod[known], kind="stable")]          # by product, file order inside
        n_products = len(product_skus)
        self.indptr = np.zeros(n_products + 1, dtype=np.int64)
        np.cumsum(np.bincount(prod[known], minlength=n_products), out=self.indptr[1:])
        self.ids = np.ascontiguousarray(order, dtype=np.int32)
        self.n_reviews, self.dim, self.n_products = emb.shape[0], emb.shape[1], n_products
        h = C.c_void_p()
        _lib.check(_lib.load().rr_reviews_create(_lib.ptr(emb), self.n_reviews, self.dim, n_products,
                                                 _lib.ptr(self.indptr), _lib.ptr(self.ids), device, eps,
                                                 C.byref(h)), "rr_reviews_create")
        self._h = h

    @property
    def handle(self):
        return self._h

    def cut_for(self, rows: np.ndarray, max_rows: int) -> int:
        """Largest review id still inside the reference's ``iloc[:max_rows]`` cut
        (app/app_product_search.py:342-345) for the candidate product rows ``rows``."""
        if max_rows <= 0:
            return -1
        rows = np.unique(rows[(rows >= 0) & (rows < self.n_products)])
        counts = self.indptr[rows + 1] - self.indptr[rows]
        if int(counts.sum()) <= max_rows:
            return int(self.n_reviews)                     # nothing is cut
        sel = np.concatenate([self.ids[self.indptr[r]:self.indptr[r + 1]] for r in rows])
        return int(np.partition(sel, max_rows - 1)[max_rows - 1])

    def snippets(self, skus: Sequence[str], best_id: np.ndarray, best_score: np.ndarray,
                 text_cut: int = 600) -> Dict[str, Dict]:
        """The reference's ``{sku: {"score", "text", "stars"}}`` for candidates that have a review."""
        out: Dict[str, Dict] = {}
        for sku, rid, sc in zip(skus, best_id.tolist(), best_score.tolist()):
            if rid >= 0:
                out[str(sku)] = {"score": float(sc), "text": self.texts[rid][:text_cut],
                                 "stars": float(self.stars[rid])}
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().rr_reviews_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
