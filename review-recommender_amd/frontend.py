"""Module-level front-end shim: the reference's `run_search` as an importable function over one cached engine.

`app/app_product_search.py` defines `run_search(query, k, rerank_k, w_dense, w_bm25, w_rerank, w_prior, w_best,
prior_C, use_snips, max_scan, min_reviews, gate_penalty)` at module level (:245-248), its UI calls it positionally
(:402) and `evals/performance_metrics.py:266` calls it as `search_function(query, **config)` with the keys of
`evals/test_queries.py:255-312`.  A maintainer who wants the GPU path under the unchanged UI / eval code replaces the
function body by a call into this module:

    from review_recommender_amd import frontend
    frontend.configure(data_dir="data/processed", emb_model_dir=..., rerank_model_dir=...)   # once, e.g. under @st.cache_resource
    run_search = frontend.run_search

The engine (index, BM25, optional review index, optional GPU encoders) is built once and shared by all callers /
threads (the library serialises calls per handle, like Streamlit's shared cached resources).
"""
from __future__ import annotations

import threading
from typing import Dict, Optional, Tuple

import pandas as pd

from .engine import SearchEngine

_engine: Optional[SearchEngine] = None
_lock = threading.Lock()


def configure(data_dir=None, *, engine: Optional[SearchEngine] = None, emb_model_dir: str = "",
              rerank_model_dir: str = "", encoder=None, cross_encoder=None, device: int = 0,
              flavour: str = "app") -> SearchEngine:
    """Builds (or adopts) the engine behind `run_search`.  Models: local Hugging Face directories run on the GPU
    (cross_encoder.QueryEncoder / CrossEncoder); alternatively any object with the sentence-transformers call shape
    (`encode([q], normalize_embeddings=True)`, `predict(pairs, batch_size=64, show_progress_bar=False)`)."""
    global _engine
    with _lock:
        if engine is None:
            if data_dir is None:
                raise ValueError("configure() needs data_dir (the reference's data/processed layout) or an engine")
            if emb_model_dir and encoder is None:
                from .cross_encoder import QueryEncoder
                encoder = QueryEncoder.from_pretrained_dir(emb_model_dir, device=device)
            if rerank_model_dir and cross_encoder is None:
                from .cross_encoder import CrossEncoder
                cross_encoder = CrossEncoder.from_pretrained_dir(rerank_model_dir, device=device)
            engine = SearchEngine.from_artifacts(data_dir, encoder=encoder, cross_encoder=cross_encoder,
                                                 device=device, flavour=flavour)
        _engine = engine
        return engine


def engine() -> SearchEngine:
    if _engine is None:
        raise RuntimeError("review_recommender_amd.frontend.configure(...) has not been called")
    return _engine


def run_search(query: str, k: int, rerank_k: int, w_dense: float, w_bm25: float, w_rerank: float, w_prior: float,
               w_best: float, prior_C: float, use_snips: bool, max_scan: int, min_reviews: int,
               gate_penalty: float) -> Tuple[pd.DataFrame, Dict, Dict]:
    """app/app_product_search.py:245-248: same name, same thirteen parameters (positional or keyword), same
    `(DataFrame, snips, dbg)` triple."""
    return engine().run_search(query, k, rerank_k, w_dense, w_bm25, w_rerank, w_prior, w_best, prior_C, use_snips,
                               max_scan, min_reviews, gate_penalty)
