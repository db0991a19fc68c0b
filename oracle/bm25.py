"""Oracle: Okapi BM25 as the third-party package rank_bm25 computes it.

TEST INFRASTRUCTURE -- never imported by the product path.

**Parity unpinned.**  The reference calls ``rank_bm25.BM25Okapi(corpus)`` and
``.get_scores(tokens)`` (app/app_product_search.py:142,206; app/test.py:156,170)
with default parameters.  rank_bm25 is not vendored under /root/reference, is
not pinned in requirements.txt / pyproject.toml, and is not installed in this
image; the reference's tests hold no BM25 value.  This file restates the
published algorithm of rank_bm25 0.2.x ``BM25Okapi``:

  ctor      per document: term-frequency dict, length; nd[t] = #docs containing
            t; avgdl = sum(len) / N                                (BM25.__init__)
  idf       idf[t] = ln(N - df + 0.5) - ln(df + 0.5), summed in vocabulary
            insertion order; average_idf = sum / |V|; every NEGATIVE idf is
            replaced by epsilon * average_idf                     (_calc_idf)
  scores    score = zeros(N, float64); for each query token IN ORDER
            (duplicates counted again):
              score += idf.get(t, 0) * tf*(k1+1) / (tf + k1*(1-b+b*dl/avgdl))
                                                                   (get_scores)

Two scorers are given: ``get_scores`` is the per-token dense numpy form the
package uses; ``get_scores_loop`` is the literal per-document Python loop (the
shape of the work the reference pays for today) used for the CPU baseline.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np


class BM25OkapiOracle:
    def __init__(self, corpus: Sequence[Sequence[str]], k1: float = 1.5,
                 b: float = 0.75, epsilon: float = 0.25):
        self.k1, self.b, self.epsilon = k1, b, epsilon
        self.corpus_size = 0
        self.doc_freqs: List[Dict[str, int]] = []
        self.doc_len: List[int] = []
        nd: Dict[str, int] = {}
        total = 0
        for doc in corpus:
            self.doc_len.append(len(doc))
            total += len(doc)
            freq: Dict[str, int] = {}
            for w in doc:
                freq[w] = freq.get(w, 0) + 1
            self.doc_freqs.append(freq)
            for w in freq:
                nd[w] = nd.get(w, 0) + 1
            self.corpus_size += 1
        self.avgdl = total / self.corpus_size
        self.idf: Dict[str, float] = {}
        idf_sum = 0.0
        negative = []
        for w, df in nd.items():
            v = math.log(self.corpus_size - df + 0.5) - math.log(df + 0.5)
            self.idf[w] = v
            idf_sum += v
            if v < 0:
                negative.append(w)
        self.average_idf = idf_sum / len(self.idf)
        floor = self.epsilon * self.average_idf
        for w in negative:
            self.idf[w] = floor

    def get_scores(self, query: Sequence[str]) -> np.ndarray:
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for t in query:
            tf = np.array([(d.get(t) or 0) for d in self.doc_freqs])
            score += (self.idf.get(t) or 0) * (
                tf * (self.k1 + 1)
                / (tf + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl)))
        return score

    get_scores_loop = get_scores


class BM25CsrOracle:
    """The same arithmetic over integer term ids and a CSR postings layout.

    Vectorised so the oracle finishes in seconds at 1M documents.  For every
    posting the float64 operations are the ones ``get_scores`` performs for a
    document with tf > 0; documents with tf == 0 receive +0.0 there, so the two
    forms agree bit for bit (checked in tests/test_oracle_bm25.py).
    """

    def __init__(self, indptr: np.ndarray, doc_ids: np.ndarray, tf: np.ndarray,
                 doc_len: np.ndarray, idf: np.ndarray, avgdl: float,
                 k1: float = 1.5, b: float = 0.75):
        self.indptr, self.doc_ids, self.tf = indptr, doc_ids, tf
        self.doc_len = doc_len.astype(np.int64)
        self.idf, self.avgdl, self.k1, self.b = idf, float(avgdl), k1, b
        self.n_docs = len(doc_len)

    def get_scores(self, term_ids: Sequence[int]) -> np.ndarray:
        score = np.zeros(self.n_docs)
        for t in term_ids:
            if t < 0:
                continue  # token absent from the vocabulary: idf.get -> 0
            lo, hi = int(self.indptr[t]), int(self.indptr[t + 1])
            docs = self.doc_ids[lo:hi]
            tf = self.tf[lo:hi].astype(np.int64)
            dl = self.doc_len[docs]
            score[docs] += self.idf[t] * (
                tf * (self.k1 + 1)
                / (tf + self.k1 * (1 - self.b + self.b * dl / self.avgdl)))
        return score
