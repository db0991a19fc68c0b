"""BM25 index of the hot path: host-side CSR construction + the device scorer.

Stands in for ``rank_bm25.BM25Okapi`` as the reference uses it
(``BM25Okapi(blob["corpus"])`` app/app_product_search.py:142, app/test.py:156;
``.get_scores(tokens)`` app/app_product_search.py:206, app/test.py:170), with the
package's default parameters k1=1.5, b=0.75, epsilon=0.25.

Host side (this file): vocabulary in first-appearance order, per-document term
frequencies, document lengths, avgdl, idf with the epsilon floor -- computed once
at load, in float64 with ``math.log`` per term and a running sum in vocabulary
order, like the package.  Device side (csrc/rr_bm25.hip): every per-query operation.
The corpus is stored twice on the device, 8 bytes per posting each way:
postings (term -> ascending docs) for ``get_scores`` and a forward list
(doc -> ascending terms) for scoring a candidate pool.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib


def idf_with_floor(df: np.ndarray, n_docs: int, epsilon: float = 0.25) -> np.ndarray:
    """idf[t] = ln(N - df + 0.5) - ln(df + 0.5); negatives -> epsilon * mean(idf).

    ``df`` is in vocabulary order; the mean is a left-to-right float64 sum in that
    order divided by the vocabulary size (rank_bm25 ``_calc_idf``)."""
    out = np.empty(len(df), dtype=np.float64)
    total = 0.0
    log = math.log
    for i, f in enumerate(df.tolist()):
        v = log(n_docs - f + 0.5) - log(f + 0.5)
        out[i] = v
        total += v
    if len(df):
        floor = epsilon * (total / len(df))
        out[out < 0] = floor
    return out


class BM25Corpus:
    """Host-side integer-id corpus: forward CSR + corpus-wide statistics.

    ``doc_terms`` ascend within each document.  ``idf`` / ``avgdl``
    always describe the corpus the statistics were taken over, also after
    ``select`` / ``slice`` (a shard scores with corpus-wide statistics).
    """

    def __init__(self, doc_indptr, doc_terms, doc_tf, doc_len, n_terms: int, *,
                 idf: Optional[np.ndarray] = None, avgdl: Optional[float] = None,
                 k1: float = 1.5, b: float = 0.75, epsilon: float = 0.25,
                 vocab: Optional[Dict[str, int]] = None):
        self.doc_indptr = np.ascontiguousarray(doc_indptr, dtype=np.int64)
        self.doc_terms = np.ascontiguousarray(doc_terms, dtype=np.int32)
        self.doc_tf = np.ascontiguousarray(doc_tf, dtype=np.int32)
        self.doc_len = np.ascontiguousarray(doc_len, dtype=np.int32)
        self.n_docs, self.n_terms = len(self.doc_len), int(n_terms)
        if len(self.doc_indptr) != self.n_docs + 1:
            raise ValueError("doc_indptr must have n_docs + 1 entries")
        self.nnz = int(self.doc_indptr[-1])
        if len(self.doc_terms) != self.nnz or len(self.doc_tf) != self.nnz:
            raise ValueError("doc_terms / doc_tf length does not match doc_indptr[-1]")
        self.k1, self.b, self.epsilon, self.vocab = float(k1), float(b), float(epsilon), vocab
        if avgdl is None:
            avgdl = int(self.doc_len.astype(np.int64).sum()) / self.n_docs
        self.avgdl = float(avgdl)
        if idf is None:
            df = np.bincount(self.doc_terms, minlength=self.n_terms).astype(np.int64)
            idf = idf_with_floor(df, self.n_docs, epsilon)
        self.idf = np.ascontiguousarray(idf, dtype=np.float64)
        if len(self.idf) != self.n_terms:
            raise ValueError("idf must have one entry per vocabulary term")

    @classmethod
    def from_corpus(cls, corpus: Sequence[Sequence[str]], **kw) -> "BM25Corpus":
        """``corpus`` = list of token lists, the ``blob["corpus"]`` of product_bm25.pkl
        (nlp/12_product_prep.py:85-89).  Vocabulary ids follow first appearance, the
        order rank_bm25's dicts iterate in."""
        vocab: Dict[str, int] = {}
        indptr = np.zeros(len(corpus) + 1, dtype=np.int64)
        terms: List[np.ndarray] = []
        tfs: List[np.ndarray] = []
        doc_len = np.zeros(len(corpus), dtype=np.int32)
        for d, doc in enumerate(corpus):
            doc_len[d] = len(doc)
            counts: Dict[int, int] = {}
            for w in doc:
                t = vocab.get(w)
                if t is None:
                    t = vocab[w] = len(vocab)
                counts[t] = counts.get(t, 0) + 1
            ids = np.fromiter(counts.keys(), dtype=np.int32, count=len(counts))
            cnt = np.fromiter(counts.values(), dtype=np.int32, count=len(counts))
            o = np.argsort(ids, kind="stable")
            terms.append(ids[o])
            tfs.append(cnt[o])
            indptr[d + 1] = indptr[d] + len(counts)
        cat = (lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int32))
        return cls(indptr, cat(terms), cat(tfs), doc_len, len(vocab), vocab=vocab, **kw)

    def _derive(self, indptr, terms, tf, doc_len) -> "BM25Corpus":
        return BM25Corpus(indptr, terms, tf, doc_len, self.n_terms, idf=self.idf,
                          avgdl=self.avgdl, k1=self.k1, b=self.b, epsilon=self.epsilon,
                          vocab=self.vocab)

    def select(self, order: np.ndarray) -> "BM25Corpus":
        """Document j of the result is document order[j] of this corpus; -1 gives an
        empty document (scores 0.0 -- the ``by_sku.get(sku, 0.0)`` miss of
        app/app_product_search.py:208)."""
        order = np.asarray(order, dtype=np.int64)
        safe = np.where(order >= 0, order, 0)
        counts = np.where(order >= 0, np.diff(self.doc_indptr)[safe], 0)
        indptr = np.zeros(len(order) + 1, dtype=np.int64)
        np.cumsum(counts, out=indptr[1:])
        starts = self.doc_indptr[safe]
        # gather index: for every output entry, its position in the source arrays
        src = np.repeat(starts - indptr[:-1], counts) + np.arange(indptr[-1], dtype=np.int64)
        doc_len = np.where(order >= 0, self.doc_len[safe], 0)
        return self._derive(indptr, self.doc_terms[src], self.doc_tf[src], doc_len)

    def slice(self, lo: int, hi: int) -> "BM25Corpus":
        s, e = int(self.doc_indptr[lo]), int(self.doc_indptr[hi])
        return self._derive(self.doc_indptr[lo:hi + 1] - s, self.doc_terms[s:e],
                            self.doc_tf[s:e], self.doc_len[lo:hi])

    def term_ids(self, tokens: Sequence[str]) -> np.ndarray:
        """Vocabulary ids of query tokens; -1 for a token the corpus never saw
        (``idf.get(q) or 0`` in the package)."""
        if self.vocab is None:
            raise ValueError("this corpus was built from integer ids; pass ids directly")
        return np.array([self.vocab.get(t, -1) for t in tokens], dtype=np.int32)

    def to_device(self, device: int = 0, row_offset: int = 0) -> "BM25Index":
        return BM25Index(self, device=device, row_offset=row_offset)


class BM25Index:
    """A BM25Corpus resident on one GPU (postings + forward lists)."""

    def __init__(self, corpus: BM25Corpus, device: int = 0, row_offset: int = 0):
        c = corpus
        self.corpus = c
        self.n_docs, self.n_terms, self.nnz = c.n_docs, c.n_terms, c.nnz
        self.row_offset, self.device = int(row_offset), int(device)
        # postings = forward entries stably sorted by term (docs stay ascending per term)
        order = np.argsort(c.doc_terms, kind="stable")
        doc_of_entry = np.repeat(np.arange(c.n_docs, dtype=np.int32), np.diff(c.doc_indptr))
        post_docs = np.ascontiguousarray(doc_of_entry[order])
        post_tf = np.ascontiguousarray(c.doc_tf[order])
        post_indptr = np.zeros(c.n_terms + 1, dtype=np.int64)
        np.cumsum(np.bincount(c.doc_terms, minlength=c.n_terms), out=post_indptr[1:])
        lib = _lib.load()
        handle = C.c_void_p()
        _lib.check(lib.rr_bm25_create(
            device, c.n_docs, c.n_terms, c.nnz, _lib.ptr(post_indptr), _lib.ptr(post_docs),
            _lib.ptr(post_tf), _lib.ptr(c.doc_indptr), _lib.ptr(c.doc_terms), _lib.ptr(c.doc_tf),
            _lib.ptr(c.doc_len), _lib.ptr(c.idf), c.avgdl, c.k1, c.b, self.row_offset,
            C.byref(handle)), "rr_bm25_create")
        self._h = handle

    def term_ids(self, tokens: Sequence[str]) -> np.ndarray:
        return self.corpus.term_ids(tokens)

    def get_scores_ids(self, term_ids: Sequence[int]) -> np.ndarray:
        ids = np.ascontiguousarray(term_ids, dtype=np.int32)
        out = np.empty(self.n_docs, dtype=np.float64)
        _lib.check(_lib.load().rr_bm25_get_scores(self._h, _lib.ptr(ids), len(ids), _lib.ptr(out)),
                   "rr_bm25_get_scores")
        return out

    def scores_at_ids(self, term_id_lists: Sequence[Sequence[int]], rows: np.ndarray,
                      mode: str = "forward") -> np.ndarray:
        """BM25 of query q at rows[q, :] (global rows).  float32, shape of ``rows``."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        if rows.ndim != 2 or rows.shape[0] != len(term_id_lists):
            raise ValueError("rows must be (n_queries, pool)")
        off = np.zeros(len(term_id_lists) + 1, dtype=np.int32)
        for i, t in enumerate(term_id_lists):
            off[i + 1] = off[i] + len(t)
        flat = (np.concatenate([np.asarray(t, dtype=np.int32) for t in term_id_lists])
                if off[-1] else np.zeros(1, dtype=np.int32))
        flat = np.ascontiguousarray(flat, dtype=np.int32)
        out = np.zeros(rows.shape, dtype=np.float32)
        if rows.shape[1] == 0:
            return out
        _lib.check(_lib.load().rr_bm25_scores_at(
            self._h, _lib.ptr(flat), _lib.ptr(off), rows.shape[0], _lib.ptr(rows), rows.shape[1],
            {"forward": 0, "postings": 1}[mode], _lib.ptr(out)), "rr_bm25_scores_at")
        return out

    @property
    def handle(self):
        return self._h

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().rr_bm25_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BM25Okapi:
    """Drop-in for ``rank_bm25.BM25Okapi`` on the calls the reference makes:
    ``BM25Okapi(corpus)`` and ``get_scores(tokens) -> float64[N]``."""

    def __init__(self, corpus: Sequence[Sequence[str]], tokenizer=None, k1: float = 1.5,
                 b: float = 0.75, epsilon: float = 0.25, device: int = 0):
        if tokenizer is not None:
            corpus = [tokenizer(doc) for doc in corpus]
        self.index = BM25Corpus.from_corpus(corpus, k1=k1, b=b, epsilon=epsilon).to_device(device)
        self.corpus_size = self.index.n_docs
        self.avgdl = self.index.corpus.avgdl
        self.k1, self.b, self.epsilon = k1, b, epsilon

    @property
    def idf(self) -> Dict[str, float]:
        c = self.index.corpus
        return {w: float(c.idf[i]) for w, i in c.vocab.items()}

    def get_scores(self, query: Sequence[str]) -> np.ndarray:
        return self.index.get_scores_ids(self.index.term_ids(query))

    def get_batch_scores(self, query: Sequence[str], doc_ids: Sequence[int]) -> List[float]:
        rows = np.asarray(doc_ids, dtype=np.int64)[None, :]
        return self.index.scores_at_ids([self.index.term_ids(query)], rows)[0].astype(
            np.float64).tolist()
