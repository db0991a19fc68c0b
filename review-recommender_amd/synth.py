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


# ---------------------------------------------------------------- K5: seeded BERT weights
# Real ms-marco-MiniLM / bge-small weights are not available offline: parity of the
# encoder kernels is checked on seeded random weights of the same architecture.  The recipe below is plain
# numpy (identical on every machine); tests/golden/make_k5_golden.py loads the same arrays into
# transformers' BertForSequenceClassification / BertModel to produce the committed logits.
def bert_state_dict(seed: int, n_layers: int = 6, hidden: int = 384, ffn: int = 1536, vocab: int = 30522,
                    max_pos: int = 512, type_vocab: int = 2, n_labels: int = 1, prefix: str = "bert.") -> dict:
    """Hugging Face BERT state-dict names -> float32 arrays.  Scales are chosen so that activations, attention
    scores and logits are O(1) (weights N(0, 0.05), LayerNorm gains 1 + N(0, 0.1)): a demanding numerics test,
    unlike the tiny initializer_range=0.02 default.  n_labels = 0 gives an encoder without pooler / classifier."""
    rng = np.random.default_rng(seed)
    nrm = lambda shape, s: (rng.standard_normal(shape, dtype=np.float32) * np.float32(s)).astype(np.float32)
    sd = {}
    e = prefix + "embeddings."
    sd[e + "word_embeddings.weight"] = nrm((vocab, hidden), 0.05)
    sd[e + "position_embeddings.weight"] = nrm((max_pos, hidden), 0.05)
    sd[e + "token_type_embeddings.weight"] = nrm((type_vocab, hidden), 0.05)
    sd[e + "LayerNorm.weight"] = (1.0 + nrm((hidden,), 0.1)).astype(np.float32)
    sd[e + "LayerNorm.bias"] = nrm((hidden,), 0.05)
    for l in range(n_layers):
        p = f"{prefix}encoder.layer.{l}."
        for name in ("query", "key", "value"):
            sd[p + f"attention.self.{name}.weight"] = nrm((hidden, hidden), 0.05)
            sd[p + f"attention.self.{name}.bias"] = nrm((hidden,), 0.02)
        sd[p + "attention.output.dense.weight"] = nrm((hidden, hidden), 0.05)
        sd[p + "attention.output.dense.bias"] = nrm((hidden,), 0.02)
        sd[p + "attention.output.LayerNorm.weight"] = (1.0 + nrm((hidden,), 0.1)).astype(np.float32)
        sd[p + "attention.output.LayerNorm.bias"] = nrm((hidden,), 0.05)
        sd[p + "intermediate.dense.weight"] = nrm((ffn, hidden), 0.05)
        sd[p + "intermediate.dense.bias"] = nrm((ffn,), 0.02)
        sd[p + "output.dense.weight"] = nrm((hidden, ffn), 0.05)
        sd[p + "output.dense.bias"] = nrm((hidden,), 0.02)
        sd[p + "output.LayerNorm.weight"] = (1.0 + nrm((hidden,), 0.1)).astype(np.float32)
        sd[p + "output.LayerNorm.bias"] = nrm((hidden,), 0.05)
    if n_labels > 0:
        sd[prefix + "pooler.dense.weight"] = nrm((hidden, hidden), 0.05)
        sd[prefix + "pooler.dense.bias"] = nrm((hidden,), 0.02)
        sd["classifier.weight"] = nrm((n_labels, hidden), 0.05)
        sd["classifier.bias"] = nrm((n_labels,), 0.02)
    return sd


def token_pairs(n_pairs: int, seed: int, vocab: int = 30522, min_len: int = 4, max_len: int = 512,
                query_len=(3, 12)):
    """Synthetic [CLS] query [SEP] text [SEP] id sequences: list of (token_ids, type_ids) int32 arrays with
    total lengths spread over [min_len, max_len] (ids 999.. avoid the special-token range)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_pairs):
        total = int(rng.integers(min_len, max_len + 1))
        ql = int(min(rng.integers(query_len[0], query_len[1] + 1), max(total - 3, 1)))
        tl = max(total - 3 - ql, 0)
        q = rng.integers(999, vocab, ql)
        t = rng.integers(999, vocab, tl)
        ids = np.concatenate([[101], q, [102], t, [102]]).astype(np.int32)
        typ = np.concatenate([np.zeros(ql + 2), np.ones(tl + 1)]).astype(np.int32)
        out.append((ids, typ))
    return out
