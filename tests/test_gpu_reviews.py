"""best review per candidate (use_snips) against the oracle's restatement of
_best_snippets (app/app_product_search.py:320-370)."""
import numpy as np
import pandas as pd
import pytest

from oracle.bm25 import BM25OkapiOracle
from oracle.pipeline import run_search_oracle
from review_recommender_amd import synth
from review_recommender_amd.engine import SearchEngine

pytestmark = pytest.mark.gpu


def make(n=3000, n_rev=20000, seed=3):
    V = synth.unit_rows(n, 384, seed)
    n_r, stars = synth.metadata(n, seed + 1)
    texts = synth.text_corpus(n, seed + 2, 20)
    meta = pd.DataFrame({"sku": synth.skus(n), "n_reviews": n_r, "avg_stars": stars, "agg_text": texts})
    rng = np.random.default_rng(seed + 3)
    owner = rng.integers(0, n + 200, n_rev)                         # some reviews belong to unknown skus
    E = (rng.standard_normal((n_rev, 384)) * rng.uniform(0.5, 3.0, (n_rev, 1))).astype(np.float32)
    # reviews resemble their product so best scores are well separated
    known = owner < n
    E[known] = (V[owner[known]] * 4 + E[known] * 0.05).astype(np.float32)
    reviews = pd.DataFrame({"sku": [f"B{o:09d}" for o in owner],
                            "text": [f"review {i} " + "x" * (i % 900) for i in range(n_rev)],
                            "stars": rng.integers(1, 6, n_rev).astype(np.float64)})
    return meta, V, reviews, E


@pytest.mark.parametrize("flavour,max_scan", [("app", 300000), ("app", 150), ("cli", 1000000), ("app", 0)])
def test_snippets_and_best_column(flavour, max_scan):
    meta, V, reviews, E = make()
    corpus = [t.split() for t in meta["agg_text"]]
    blob = {"skus": meta["sku"].tolist(), "corpus": corpus}
    engine = SearchEngine(meta, V, blob, normalize=False, flavour=flavour, reviews=(reviews, E))
    ora = BM25OkapiOracle(corpus)
    cfg = dict(k=10, rerank_k=0, w_dense=0.5, w_bm25=0.2, w_rerank=0.0, w_prior=0.1, w_best=0.2, prior_C=20.0,
               min_reviews=8, gate_penalty=1.0)
    for seed, query in ((41, "wireless mug"), (42, "cat socks")):
        qv = synth.unit_rows(1, 384, seed)[0]
        want, want_snips, _, cand = run_search_oracle(query=query, qvec=qv, meta=meta, V=V, bm25=ora,
                                                      bm25_skus=blob["skus"], flavour=flavour, use_snips=True,
                                                      max_scan=max_scan, reviews=(reviews, E), **cfg)
        got, snips, _ = engine.run_search(query, cfg["k"], 0, 0.5, 0.2, 0.0, 0.1, 0.2, 20.0, True, max_scan, 8, 1.0,
                                          qvec=qv)
        assert set(snips) == set(want_snips)
        if max_scan > 0:
            assert len(snips) > 20
        for sku, w in want_snips.items():
            g = snips[sku]
            assert g["text"] == w["text"] and g["stars"] == w["stars"]
            assert abs(g["score"] - w["score"]) < 1e-5
            assert len(g["text"]) <= (600 if flavour == "app" else 400)
        assert got["sku"].tolist() == want["sku"].tolist()
        np.testing.assert_allclose(got["_final"].values, want["_final"].values, atol=1e-5, rtol=0)
        np.testing.assert_allclose(got["_best"].values, want["_best"].values, atol=1e-5, rtol=0)


def test_no_review_index_means_empty_snips():
    meta, V, _, _ = make(500, 10)
    engine = SearchEngine(meta, V, None, normalize=False)
    qv = synth.unit_rows(1, 384, 5)[0]
    got, snips, _ = engine.run_search("mug", 5, 0, 1.0, 0, 0, 0, 0.5, 20.0, True, 1000, 8, 1.0, qvec=qv)
    assert snips == {} and np.all(got["_best"] == 0)


@pytest.mark.parametrize("max_scan", [300000, 150, 37, 1, 0])
def test_batched_snippets_with_the_per_query_cut_on_the_device(max_scan):
    """One rr_reviews_best_cut_dev call for a whole batch (no host round trip for the iloc[:max_rows] cut):
    per query the snippets must be the ones the oracle's _best_snippets finds for that query alone."""
    from oracle.dense import cosine_similarity_search
    from oracle.pipeline import best_snippets_oracle
    from review_recommender_amd.engine import FusionWeights
    meta, V, reviews, E = make(2500, 15000, seed=9)
    engine = SearchEngine(meta, V, None, normalize=False, reviews=(reviews, E))
    Q = synth.unit_rows(9, 384, 77)
    w = FusionWeights(0.6, 0.0, 0.0, 0.0, 0.4, gate_penalty=1.0)
    res = engine.searcher.search_batch(Q, None, 10, 0, w, reviews=engine.reviews, max_scan=max_scan)
    skus = meta["sku"].astype(str)
    for b in range(9):
        rows_o, _ = cosine_similarity_search(Q[b], V, 150)
        assert np.array_equal(rows_o, res.pool_rows[b])
        want = best_snippets_oracle(reviews, E, Q[b], skus.iloc[rows_o].tolist(), max_rows=max_scan)
        got = engine.reviews.snippets(skus.iloc[res.pool_rows[b]].tolist(), res.best_ids[b], res.best_raw[b])
        assert set(got) == set(want)
        for s_, w_ in want.items():
            assert got[s_]["text"] == w_["text"] and abs(got[s_]["score"] - w_["score"]) < 1e-5
