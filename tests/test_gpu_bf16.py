"""bf16 storage (BASELINE configs 4-5): the matrix is rounded to bf16 once (after the fp32
normalisation), queries / products / accumulation stay fp32.  Oracle: the
reference's fp32 matvec + argpartition over the rounded matrix widened back to fp32."""
import numpy as np
import pandas as pd
import pytest

from oracle import dense as OD
from oracle.bm25 import BM25OkapiOracle
from oracle.pipeline import run_search_oracle
from oracle.primitives import l2_normalize
from parity import assert_topk_matches, min_gap
from review_recommender_amd import synth
from review_recommender_amd.engine import SearchEngine
from review_recommender_amd.index import ProductIndex

pytestmark = pytest.mark.gpu


def test_round_to_bf16_is_nearest_even():
    x = np.array([1.0, 1.00390625, 1.005859375, 1.001953125, -2.5, 3.0e-5, 0.0], dtype=np.float32)
    r = OD.round_to_bf16(x)
    assert np.all((r.view(np.uint32) & 0xFFFF) == 0)
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.0078125      # tie -> even, above tie -> up
    assert np.all(np.abs(r - x) <= np.abs(x) * 2.0 ** -8)


@pytest.mark.parametrize("batch", [1, 3, 8, 16, 40, 64, 70])
@pytest.mark.parametrize("n", [1000, 4097, 50_000])
def test_bf16_dense_parity(n, batch):
    V = synth.unit_rows(n, 384, 300 + n)
    Vb = OD.round_to_bf16(V)
    Q = synth.unit_rows(batch, 384, 5)
    ix = ProductIndex.from_rows(V, dtype="bf16")
    rows, scores = ix.dense_topk(Q, 150)
    for i in range(batch):
        ref = OD.sims_float64(Vb, Q[i])
        assert_topk_matches(rows[i], scores[i], ref, 150)
        o_rows, o_sims = OD.cosine_similarity_search(Q[i], Vb, 150)
        if min_gap(ref, 150) > 4e-7:
            assert np.array_equal(rows[i], o_rows)
        np.testing.assert_allclose(scores[i], o_sims, atol=1e-5, rtol=0)
    ix.close()


def test_bf16_index_normalises_in_fp32_then_rounds_once():
    rng = np.random.default_rng(4)
    X = (rng.standard_normal((3000, 384)) * 5).astype(np.float32)
    q = synth.unit_rows(1, 384, 6)
    ix = ProductIndex.from_rows(X, dtype="bf16", normalize=True)
    rows, scores = ix.dense_topk(q, 100)
    ref = OD.sims_float64(OD.round_to_bf16(l2_normalize(X)), q[0])
    # the device norm sums squares in another order than numpy: an element can land on the other
    # side of a bf16 rounding boundary (2^-9 relative) once in a while, so compare loosely here
    assert_topk_matches(rows[0], scores[0], ref, 100, tie_eps=2e-4, score_tol=2e-4)
    ix.close()


def test_bf16_hybrid_pipeline_matches_oracle():
    n = 8000
    V = synth.unit_rows(n, 384, 21)
    Vb = OD.round_to_bf16(V)
    n_rev, stars = synth.metadata(n, 22)
    texts = synth.text_corpus(n, 23, mean_len=20)
    meta = pd.DataFrame({"sku": synth.skus(n), "n_reviews": n_rev, "avg_stars": stars, "agg_text": texts})
    corpus = [t.split() for t in texts]
    blob = {"skus": meta["sku"].tolist(), "corpus": corpus}
    engine = SearchEngine(meta, V, blob, normalize=False, dtype="bf16")
    ora = BM25OkapiOracle(corpus)
    for seed, query in ((31, "wireless yellow mug"), (32, "cat socks design")):
        qv = synth.unit_rows(1, 384, seed)[0]
        got, _, _ = engine.search(query, k=100, alpha=0.5, qvec=qv)
        want, _, _, _ = run_search_oracle(query=query, qvec=qv, meta=meta, V=Vb, bm25=ora, bm25_skus=blob["skus"],
                                          k=100, rerank_k=0, w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0,
                                          w_best=0.0, prior_C=20.0, min_reviews=8, gate_penalty=1.0)
        assert got["sku"].tolist() == want["sku"].tolist()
        np.testing.assert_allclose(got["_final"].values, want["_final"].values, atol=1e-5, rtol=0)


def test_bf16_row_norm_bound_follows_rows_uploaded_after_a_batched_search(hip):
    """rr_index_upload_rows_f32 must reset the cached row-norm bound for bf16 storage too.
    Small-norm rows first, one batched (filter-scan) search caches their bound; larger rows are then
    uploaded over them and the batched answer must still equal the single-query scan's."""
    import ctypes as C
    from review_recommender_amd import _lib
    n = 60_000
    small = (synth.unit_rows(n, 384, 41) * 0.05).astype(np.float32)
    big = (synth.unit_rows(n, 384, 42) * 3.0).astype(np.float32)
    Q = synth.unit_rows(40, 384, 43)
    ix = ProductIndex.from_rows(small, dtype="bf16")
    ix.dense_topk(Q, 150)                                     # caches the bound of the small rows
    _lib.check(hip.rr_index_upload_rows_f32(ix.handle, 0, n, _lib.ptr(big), 0.0), "upload")
    rows_b, scores_b = ix.dense_topk(Q, 150)                  # batched: filter scan with the bound
    for i in range(0, 40, 7):
        r1, s1 = ix.dense_topk(Q[i:i + 1], 150)               # single-query scan: no bound involved
        assert np.array_equal(rows_b[i], r1[0]) and np.array_equal(scores_b[i], s1[0])
        assert_topk_matches(rows_b[i], scores_b[i], OD.sims_float64(OD.round_to_bf16(big), Q[i]), 150,
                            score_tol=3e-5)
    ix.close()
