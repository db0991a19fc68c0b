"""The reference's on-disk artefacts in and out.

  product_emb.npy            float32 (N, 384), rows L2-normalised      nlp/11_build_product_embeddings.py:82-85
  product_emb_meta.parquet   sku, n_reviews, avg_stars, last_ts, agg_text, row-aligned     nlp/11...:86-90
  product_bm25.pkl           {"skus": [...], "corpus": [[tokens]...], "tokenizer": "simple_en_v1"},
                             pickle protocol 4                          nlp/12_product_prep.py:85-89

Loading asserts the invariants the reference's artefact auditor checks (test.py:175-178):
required columns, len(meta) == rows of the matrix, unique non-null sku.
"""
from __future__ import annotations

import pathlib
import pickle
from typing import Dict, Optional, Tuple

import numpy as np
import pandas as pd

try:                     # bind the parquet engine before any GPU runtime is loaded (import-order
    import pyarrow       # sensitivity seen on the GPU box)  # noqa: F401
except ImportError:      # pandas will report the missing engine when a parquet file is touched
    pass

from . import text

EMB_FILE = "product_emb.npy"
META_FILE = "product_emb_meta.parquet"
BM25_FILE = "product_bm25.pkl"
REVIEWS_FILE = "reviews_with_embeddings.parquet"   # sku, text, stars, embedding (app/test.py:26,181-199)
REQUIRED_COLUMNS = ("sku", "agg_text")          # app/test.py:138-139
AUDIT_COLUMNS = ("sku", "n_reviews", "avg_stars", "agg_text")   # test.py:175


class ArtifactError(ValueError):
    pass


def build_bm25_blob(meta: pd.DataFrame) -> Dict:
    """nlp/12_product_prep.py:80-89: index-time tokenisation of agg_text, skus as str."""
    corpus = [text.tokenize_document(t) for t in meta["agg_text"].fillna("").astype(str).tolist()]
    return {"skus": meta["sku"].astype(str).tolist(), "corpus": corpus, "tokenizer": "simple_en_v1"}


def save_artifacts(data_dir, meta: pd.DataFrame, embeddings: np.ndarray,
                   bm25_blob: Optional[Dict] = None) -> pathlib.Path:
    """Writes the three artefacts exactly as the reference's builders do."""
    d = pathlib.Path(data_dir)
    d.mkdir(parents=True, exist_ok=True)
    np.save(d / EMB_FILE, np.asarray(embeddings, dtype=np.float32))
    meta.to_parquet(d / META_FILE, index=False)
    if bm25_blob is not None:
        with open(d / BM25_FILE, "wb") as f:
            pickle.dump(bm25_blob, f, protocol=4)
    return d


def audit_meta(meta: pd.DataFrame, n_rows: int, strict: bool = False) -> None:
    need = AUDIT_COLUMNS if strict else REQUIRED_COLUMNS
    missing = [c for c in need if c not in meta.columns]
    if missing:
        raise ArtifactError(f"{META_FILE} lacks column(s) {missing}")
    if len(meta) != n_rows:
        # app/app_product_search.py:104-107, app/test.py:141-142
        raise ArtifactError(f"length mismatch: meta={len(meta)} vs emb_rows={n_rows}")
    if strict:
        sku = meta["sku"]
        if sku.isna().any() or sku.astype(str).duplicated().any():
            raise ArtifactError("sku must be unique and non-null")   # test.py:178


def load_artifacts(data_dir, strict: bool = False, mmap: bool = True
                   ) -> Tuple[pd.DataFrame, np.ndarray, Optional[Dict]]:
    """(meta, embeddings, bm25_blob or None).  The matrix is memory-mapped (app/test.py:140)."""
    d = pathlib.Path(data_dir)
    if not (d / EMB_FILE).exists() or not (d / META_FILE).exists():
        raise ArtifactError(f"{EMB_FILE} and/or {META_FILE} missing in {d}")   # app/test.py:135-136
    emb = np.load(d / EMB_FILE, mmap_mode="r" if mmap else None)
    if emb.ndim != 2:
        raise ArtifactError(f"{EMB_FILE} must be 2-D, got shape {emb.shape}")
    meta = pd.read_parquet(d / META_FILE)
    audit_meta(meta, emb.shape[0], strict)
    blob = None
    if (d / BM25_FILE).exists():
        with open(d / BM25_FILE, "rb") as f:
            blob = pickle.load(f)
        if not isinstance(blob, dict) or "skus" not in blob or "corpus" not in blob:
            raise ArtifactError(f"{BM25_FILE} must hold a dict with 'skus' and 'corpus'")
        if len(blob["skus"]) != len(blob["corpus"]):
            raise ArtifactError(f"{BM25_FILE}: {len(blob['skus'])} skus vs {len(blob['corpus'])} documents")
    return meta.reset_index(drop=True), emb, blob


def load_reviews(data_dir) -> Optional[Tuple[pd.DataFrame, np.ndarray]]:
    """(review table with sku / text / stars, (n_reviews, dim) float32 embeddings) from
    reviews_with_embeddings.parquet, or None when the file is absent (the reference then skips
    snippets: app/test.py:186, app/app_product_search.py:285)."""
    f = pathlib.Path(data_dir) / REVIEWS_FILE
    if not f.exists():
        return None
    df = pd.read_parquet(f)
    if "sku" not in df.columns or "embedding" not in df.columns:
        return None                                   # app/test.py:190-192: warn and skip
    emb = np.stack(df["embedding"].values).astype(np.float32) if len(df) else np.zeros((0, 1), np.float32)
    if len(df) == 0:
        return None
    return df.drop(columns=["embedding"]).reset_index(drop=True), emb


def save_reviews(data_dir, reviews: pd.DataFrame, embeddings: np.ndarray) -> pathlib.Path:
    """Writes reviews_with_embeddings.parquet in the reference's layout
    (nlp/11_build_product_embeddings.py:95-169: one list-valued `embedding` column)."""
    d = pathlib.Path(data_dir)
    d.mkdir(parents=True, exist_ok=True)
    out = reviews.copy()
    out["embedding"] = [np.asarray(e, dtype=np.float32) for e in embeddings]
    out.to_parquet(d / REVIEWS_FILE, index=False)
    return d / REVIEWS_FILE
