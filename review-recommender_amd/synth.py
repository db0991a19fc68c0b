"""Seeded synthetic inputs of the shapes SURVEY section 8d prescribes (there is no
network for the real dataset).  Used by tests, __graft_entry__.smoke() and bench.py."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def unit_rows(n: int, dim: int, seed: int) -> np.ndarray:
    """standard_normal float32 rows scaled to unit L2 norm (numpy formula of utils.py:40-44)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, dim), dtype=np.float32)
    nrm = np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    return (x / nrm).astype(np.float32)


def metadata(n: int, seed: int, nan_fraction: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """n_reviews ~ floor(lognormal) clipped to [1, 5000]; avg_stars ~ clip(N(4.1, 0.6), 1, 5)
    rounded to 3 dp (nlp/10_product_prep.py:82); optionally a fraction of NaN ratings."""
    rng = np.random.default_rng(seed)
    n_reviews = np.clip(np.floor(rng.lognormal(2.5, 1.2, n)), 1, 5000).astype(np.int64)
    stars = np.round(np.clip(rng.normal(4.1, 0.6, n), 1.0, 5.0), 3)
    if nan_fraction > 0:
        stars[rng.random(n) < nan_fraction] = np.nan
    return n_reviews, stars


def skus(n: int, first: int = 0) -> List[str]:
    return [f"B{first + i:09d}" for i in range(n)]


def zipf_cdf(vocab: int, s: float = 1.07) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, vocab + 1, dtype=np.float64), s)
    return np.cumsum(w / w.sum())


def bm25_forward_csr(n_docs: int, vocab: int, mean_len: int, seed: int, s: float = 1.07):
    """Integer-id corpus: document length ~ max(1, Poisson(mean_len)) tokens drawn from
    Zipf(s) over ``vocab`` terms.  Returns (doc_indptr, doc_terms, doc_tf, doc_len) with
    ascending term ids per document."""
    rng = np.random.default_rng(seed)
    doc_len = np.maximum(1, rng.poisson(mean_len, n_docs)).astype(np.int64)
    total = int(doc_len.sum())
    cdf = zipf_cdf(vocab, s)
    tok = np.searchsorted(cdf, rng.random(total), side="left").astype(np.int64)
    np.minimum(tok, vocab - 1, out=tok)
    doc = np.repeat(np.arange(n_docs, dtype=np.int64), doc_len)
    key = doc * vocab + tok
    key.sort()
    uniq, counts = np.unique(key, return_counts=True)
    d = uniq // vocab
    indptr = np.zeros(n_docs + 1, dtype=np.int64)
    np.cumsum(np.bincount(d, minlength=n_docs), out=indptr[1:])
    return indptr, (uniq % vocab).astype(np.int32), counts.astype(np.int32), doc_len.astype(np.int32)


def query_terms(n_queries: int, vocab: int, seed: int, df: np.ndarray = None,
                lo: int = 3, hi: int = 6, s: float = 1.07) -> List[np.ndarray]:
    """3-6 term ids per query from the same Zipf law, restricted to terms with df >= 1."""
    rng = np.random.default_rng(seed)
    cdf = zipf_cdf(vocab, s)
    out = []
    for _ in range(n_queries):
        want = int(rng.integers(lo, hi + 1))
        ids: List[int] = []
        while len(ids) < want:
            t = int(min(np.searchsorted(cdf, rng.random()), vocab - 1))
            if df is None or df[t] >= 1:
                ids.append(t)
        out.append(np.array(ids, dtype=np.int32))
    return out


WORDS = ("wireless headphones bluetooth noise cancelling yellow cat socks soft comfortable cute "
         "gaming keyboard mechanical rgb backlight kitchen knife steel sharp blue red green black "
         "white dog toy puppy leash cotton shirt design pattern print graphic travel mug coffee "
         "insulated bottle water running shoes light durable battery charger cable fast usb").split()


def text_corpus(n_docs: int, seed: int, mean_len: int = 12) -> List[str]:
    """Small English-looking product texts for gate / tokenizer / BM25-string tests."""
    rng = np.random.default_rng(seed)
    lens = np.maximum(3, rng.poisson(mean_len, n_docs))
    return [" ".join(rng.choice(WORDS, size=int(m))) for m in lens]
