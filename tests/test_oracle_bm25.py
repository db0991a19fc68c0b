"""BM25 oracle: hand-computed known answers on the reference's own 3-document fixture
corpus (tests/conftest.py:90-100), the epsilon floor, and dict-form == CSR-form bit for bit.
rank_bm25 itself is absent from the image: BM25 parity is UNPINNED (oracle/bm25.py header)."""
import math

import numpy as np

from oracle.bm25 import BM25CsrOracle, BM25OkapiOracle
from review_recommender_amd import synth
from review_recommender_amd.bm25 import BM25Corpus, idf_with_floor

CORPUS = [["wireless", "headphones", "bluetooth"],
          ["yellow", "cat", "socks", "soft"],
          ["gaming", "keyboard", "mechanical"]]


def test_hand_computed_three_doc_corpus():
    bm = BM25OkapiOracle(CORPUS)
    assert bm.corpus_size == 3 and bm.avgdl == 10 / 3
    idf1 = math.log(3 - 1 + 0.5) - math.log(1 + 0.5)          # every term has df = 1
    assert all(v == idf1 for v in bm.idf.values())
    s = bm.get_scores(["wireless", "headphones"])
    per_term = idf1 * (1 * 2.5 / (1 + 1.5 * (1 - 0.75 + 0.75 * 3 / (10 / 3))))
    assert s[0] == per_term + per_term and s[1] == 0.0 and s[2] == 0.0
    # a duplicated query token counts twice; an unknown token adds nothing
    s2 = bm.get_scores(["cat", "cat", "zebra"])
    one = idf1 * (1 * 2.5 / (1 + 1.5 * (1 - 0.75 + 0.75 * 4 / (10 / 3))))
    assert s2[1] == one + one and s2[0] == 0.0


def test_negative_idf_gets_the_epsilon_floor():
    corpus = [["a1", "b1"], ["a1", "c1"], ["a1", "d1"], ["a1", "b1"]]
    bm = BM25OkapiOracle(corpus)
    raw = {w: math.log(4 - df + 0.5) - math.log(df + 0.5) for w, df in
           {"a1": 4, "b1": 2, "c1": 1, "d1": 1}.items()}
    assert raw["a1"] < 0
    avg = (raw["a1"] + raw["b1"] + raw["c1"] + raw["d1"]) / 4
    assert bm.idf["a1"] == 0.25 * avg and bm.idf["b1"] == raw["b1"]


def _string_corpus(n_docs, seed):
    return [t.split() for t in synth.text_corpus(n_docs, seed)]


def test_csr_form_equals_dict_form_bit_for_bit():
    corpus = _string_corpus(400, 7)
    bm = BM25OkapiOracle(corpus)
    host = BM25Corpus.from_corpus(corpus)
    # the product's host-side statistics equal the oracle's, exactly
    assert host.avgdl == bm.avgdl and host.n_docs == bm.corpus_size
    assert list(host.vocab) == list(bm.idf)
    assert np.array_equal(host.idf, np.array([bm.idf[w] for w in host.vocab]))
    order = np.argsort(host.doc_terms, kind="stable")
    docs = np.repeat(np.arange(host.n_docs), np.diff(host.doc_indptr))[order]
    indptr = np.concatenate([[0], np.cumsum(np.bincount(host.doc_terms, minlength=host.n_terms))])
    csr = BM25CsrOracle(indptr, docs, host.doc_tf[order], host.doc_len, host.idf, host.avgdl)
    for q in (["wireless", "cat"], ["dog", "dog", "zzz", "usb"], [], ["blue"]):
        a = bm.get_scores(q)
        b = csr.get_scores(host.term_ids(q))
        assert np.array_equal(a, b)


def test_idf_helper_matches_oracle_on_integer_corpus():
    ip, terms, tf, dl = synth.bm25_forward_csr(300, 50, 12, 3)
    df = np.bincount(terms, minlength=50)
    idf = idf_with_floor(df, 300)
    raw = np.array([math.log(300 - f + 0.5) - math.log(f + 0.5) for f in df.tolist()])
    total = 0.0
    for v in raw.tolist():
        total += v                      # left-to-right, vocabulary order
    floor = 0.25 * (total / 50)
    assert np.array_equal(idf[raw >= 0], raw[raw >= 0])
    assert (raw < 0).any() and np.all(idf[raw < 0] == floor)


def test_select_and_slice_keep_corpus_statistics():
    corpus = _string_corpus(50, 11)
    host = BM25Corpus.from_corpus(corpus)
    order = np.array([3, -1, 0, 3, 49])
    sel = host.select(order)
    assert sel.n_docs == 5 and sel.avgdl == host.avgdl and sel.idf is host.idf
    assert sel.doc_len.tolist() == [len(corpus[3]), 0, len(corpus[0]), len(corpus[3]), len(corpus[49])]
    s3, e3 = host.doc_indptr[3], host.doc_indptr[4]
    assert np.array_equal(sel.doc_terms[sel.doc_indptr[0]:sel.doc_indptr[1]], host.doc_terms[s3:e3])
    assert sel.doc_indptr[2] == sel.doc_indptr[1]          # the missing sku is an empty document
    sl = host.slice(10, 20)
    assert sl.n_docs == 10 and np.array_equal(sl.doc_len, host.doc_len[10:20])
