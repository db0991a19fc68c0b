"""K3 and the whole path on the GPU against the oracle pipeline (restatement of
run_search app/app_product_search.py:245-317 and the CLI search app/test.py:228-309).

Bars: returned product rows bit-exact; fused scores within 1e-5 (north star).  Where the
fusion kernel is fed the oracle's own candidate pool it must be bit-exact on every column.
"""
import ctypes as C

import numpy as np
import pandas as pd
import pytest

from oracle.bm25 import BM25OkapiOracle
from oracle.pipeline import run_search_oracle
from review_recommender_amd import _lib, synth
from review_recommender_amd.engine import COLUMN_NAMES, FusionWeights, HybridSearcher, SearchEngine

pytestmark = pytest.mark.gpu
TOL = 1e-5

# evals/test_queries.py:255-312 (BENCHMARK_CONFIGS) + the UI defaults (config.py:64-72)
CONFIGS = {
    "dense_only": dict(k=20, rerank_k=0, w_dense=1.0, w_bm25=0.0, w_rerank=0.0, w_prior=0.0, w_best=0.0,
                       prior_C=20.0, min_reviews=1, gate_penalty=0.0),
    "bm25_only": dict(k=20, rerank_k=0, w_dense=0.0, w_bm25=1.0, w_rerank=0.0, w_prior=0.0, w_best=0.0,
                      prior_C=20.0, min_reviews=1, gate_penalty=0.0),
    "hybrid": dict(k=20, rerank_k=0, w_dense=0.5, w_bm25=0.3, w_rerank=0.0, w_prior=0.2, w_best=0.0,
                   prior_C=20.0, min_reviews=5, gate_penalty=0.3),
    "hybrid_rerank": dict(k=50, rerank_k=20, w_dense=0.4, w_bm25=0.2, w_rerank=0.3, w_prior=0.1, w_best=0.0,
                          prior_C=20.0, min_reviews=5, gate_penalty=0.5),
    "ui_defaults": dict(k=10, rerank_k=50, w_dense=0.55, w_bm25=0.20, w_rerank=0.20, w_prior=0.20,
                        w_best=0.10, prior_C=20.0, min_reviews=8, gate_penalty=0.5),
    "north_star_alpha": dict(k=100, rerank_k=0, w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0,
                             w_best=0.0, prior_C=20.0, min_reviews=8, gate_penalty=1.0),
}
QUERIES = ["wireless headphones for running", "yellow cat socks", "blue insulated coffee mug",
           "the of and", "gaming keyboard rgb design"]


class FakeCrossEncoder:
    """Deterministic stand-in with the CrossEncoder.predict signature
    (the reference's tests mock it the same way, tests/test_integration.py:46-48)."""
    def predict(self, pairs, batch_size=64, show_progress_bar=False):
        return np.array([((len(t) * 7 + sum(map(ord, t[:20]))) % 97) / 9.7 - 4.0 for _, t in pairs],
                        dtype=np.float32)


@pytest.fixture(scope="module")
def world():
    n = 10_000                                  # BASELINE config 1 size
    V = synth.unit_rows(n, 384, 1234)
    n_rev, stars = synth.metadata(n, 2, nan_fraction=0.0)
    texts = synth.text_corpus(n, 3, mean_len=25)
    meta = pd.DataFrame({"sku": synth.skus(n), "n_reviews": n_rev, "avg_stars": stars,
                         "last_ts": np.arange(n), "agg_text": texts})
    corpus = [t.split() for t in texts]
    blob = {"skus": meta["sku"].tolist(), "corpus": corpus, "tokenizer": "simple_en_v1"}
    ora_bm25 = BM25OkapiOracle(corpus)
    return dict(V=V, meta=meta, blob=blob, ora_bm25=ora_bm25, n=n)


def run_both(world, engine, query, qvec, cfg, flavour):
    ce = FakeCrossEncoder()
    want, _, dbg_o, cand = run_search_oracle(
        query=query, qvec=qvec, meta=world["meta"], V=world["V"], bm25=world["ora_bm25"],
        bm25_skus=world["blob"]["skus"], flavour=flavour,
        rerank_fn=(lambda pairs: ce.predict(pairs)), **cfg)
    got, snips, dbg = engine.run_search(query, cfg["k"], cfg["rerank_k"], cfg["w_dense"], cfg["w_bm25"],
                                        cfg["w_rerank"], cfg["w_prior"], cfg["w_best"], cfg["prior_C"],
                                        False, 0, cfg["min_reviews"], cfg["gate_penalty"], qvec=qvec)
    assert snips == {} and {k: v for k, v in dbg.items() if k != "groups"} == \
        {k: v for k, v in dbg_o.items() if k != "groups"}
    assert [sorted(g) for g in dbg["groups"]] == [sorted(g) for g in dbg_o["groups"]]
    return want, got, cand


@pytest.mark.parametrize("flavour", ["app", "cli"])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_run_search_matches_oracle(world, name, flavour):
    cfg = CONFIGS[name]
    engine = SearchEngine(world["meta"], world["V"], world["blob"], cross_encoder=FakeCrossEncoder(),
                          normalize=False, flavour=flavour)
    Q = synth.unit_rows(len(QUERIES), 384, 99)
    for query, qvec in zip(QUERIES, Q):
        want, got, cand = run_both(world, engine, query, qvec, cfg, flavour)
        assert len(got) == len(want) == cfg["k"]
        np.testing.assert_allclose(got["_final"].values, want["_final"].values, atol=TOL, rtol=0)
        # IDs: exact, except that rows whose oracle finals tie (gate_penalty 0 gives runs of
        # exact zeros) may come in any order -- the reference's own sort is unstable there
        wf = want["_final"].values
        for f in np.unique(wf):
            sel = wf == f
            if f == wf[-1]:
                # the k-th value may continue below the cut: any pool row with that score qualifies
                allowed = set(cand.loc[cand["_final"].values == f, "sku"])
                assert set(got.loc[sel, "sku"]) <= allowed
            else:
                assert set(got.loc[sel, "sku"]) == set(want.loc[sel, "sku"])
        cols = [c for c in COLUMN_NAMES if c in want.columns]
        common = [s_ for s_ in got["sku"] if s_ in set(want["sku"])]
        assert len(common) >= cfg["k"] - int((wf == wf[-1]).sum())
        g_al, w_al = got.set_index("sku").loc[common], want.set_index("sku").loc[common]
        for c in cols:
            np.testing.assert_allclose(g_al[c].values.astype(np.float64),
                                       w_al[c].values.astype(np.float64), atol=TOL, rtol=0)
        assert list(got.columns[:len(world["meta"].columns)]) == list(world["meta"].columns)


def fuse_host(hip, index, params, rows, dense, bm25, rerank=None, gate=None, meta=None):
    B, pool, k = rows.shape[0], params.pool, params.k
    out_rows = np.empty((B, pool), dtype=np.int64)
    cols = np.empty((B, 8, pool), dtype=np.float64)
    order = np.empty((B, k), dtype=np.int32)
    n, avg, l1p = meta if meta is not None else (None, None, None)
    _lib.check(hip.rr_fuse_topk(index.handle, C.byref(params), B, _lib.ptr(rows), _lib.ptr(dense),
                                _lib.ptr(bm25), _lib.ptr(n), _lib.ptr(avg), _lib.ptr(l1p),
                                _lib.ptr(rerank), None, _lib.ptr(gate), _lib.ptr(out_rows),
                                _lib.ptr(cols), _lib.ptr(order)), "rr_fuse_topk")
    return out_rows, cols, order


@pytest.mark.parametrize("nan_fraction", [0.0, 0.02, 1.0])
@pytest.mark.parametrize("name", ["hybrid", "hybrid_rerank", "ui_defaults", "dense_only"])
def test_fusion_kernel_is_bit_exact_on_the_oracle_pool(hip, world, name, nan_fraction):
    """rr_fuse_topk fed the oracle's candidate pool (rows, raw dense, raw BM25, gate, raw
    reranker scores): every column of every pool row must equal numpy's result exactly."""
    cfg = CONFIGS[name]
    meta = world["meta"].copy()
    n_rev, stars = synth.metadata(world["n"], 2, nan_fraction=nan_fraction)
    meta["avg_stars"] = stars
    if nan_fraction == 1.0:
        meta["avg_stars"] = np.nan
    engine = SearchEngine(meta, world["V"], None, normalize=False)
    ce = FakeCrossEncoder()
    qvec = synth.unit_rows(1, 384, 5)[0]
    query = "yellow cat socks"
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want, _, _, cand = run_search_oracle(query=query, qvec=qvec, meta=meta, V=world["V"],
                                             bm25=world["ora_bm25"], bm25_skus=world["blob"]["skus"],
                                             rerank_fn=lambda p: ce.predict(p), **cfg)
    pool = len(cand)
    rows = cand["_row"].values.astype(np.int64)[None, :]
    dense = (world["V"] @ qvec)[rows[0]].astype(np.float32)[None, :]
    from oracle.pipeline import bm25_for_candidates_app
    bm_raw = bm25_for_candidates_app(world["ora_bm25"], world["blob"]["skus"], query,
                                     cand["sku"].tolist())[None, :]
    rr = None
    if cfg["rerank_k"] > 0:
        rr = np.zeros((1, pool), dtype=np.float32)
        texts = cand["agg_text"].astype(str).str.slice(0, 2000).tolist()[:cfg["rerank_k"]]
        rr[0, :cfg["rerank_k"]] = ce.predict([(query, t) for t in texts])
    gate = cand["_gate"].values.astype(np.float32)[None, :]
    w = FusionWeights(cfg["w_dense"], cfg["w_bm25"], cfg["w_rerank"], cfg["w_prior"], cfg["w_best"],
                      cfg["prior_C"], cfg["min_reviews"], cfg["gate_penalty"])
    params = HybridSearcher.make_params(w, cfg["k"], pool, pool, cfg["rerank_k"])
    out_rows, cols, order = fuse_host(hip, engine.index, params, rows, np.ascontiguousarray(dense),
                                      np.ascontiguousarray(bm_raw), rr, np.ascontiguousarray(gate))
    assert np.array_equal(out_rows, rows)
    for j, c in enumerate(COLUMN_NAMES):
        ref = cand[c].values.astype(np.float64)
        assert np.array_equal(cols[0, j], ref, equal_nan=True), f"column {c} differs"
    # order: final desc, stable
    fin = cand["_final"].values
    exp = np.lexsort((np.arange(pool), -fin.astype(np.float64)))[:cfg["k"]]
    assert np.array_equal(order[0], exp)


def test_search_alpha_sugar_and_batch_api(world):
    engine = SearchEngine(world["meta"], world["V"], world["blob"], normalize=False)
    qv = synth.unit_rows(1, 384, 55)[0]
    frame, snips, dbg = engine.search("wireless mug", k=100, alpha=0.5, qvec=qv)
    want, _, _, _ = run_search_oracle(query="wireless mug", qvec=qv, meta=world["meta"], V=world["V"],
                                      bm25=world["ora_bm25"], bm25_skus=world["blob"]["skus"], k=100,
                                      rerank_k=0, w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0,
                                      w_best=0.0, prior_C=20.0, min_reviews=8, gate_penalty=1.0)
    assert frame["sku"].tolist() == want["sku"].tolist()
    np.testing.assert_allclose(frame["_final"].values, want["_final"].values, atol=TOL, rtol=0)
    assert dbg["pool"] == 150 and len(frame) == 100
    # a batch of <= 4 queries == the single calls, bitwise (same scan kernel, batch-invariant);
    # a larger batch runs the matrix-core scan: same rows, scores within f32 rounding
    toks_of = lambda n: [engine.searcher.bm25.term_ids(["wireless", "mug"]) for _ in range(n)]
    w = FusionWeights(0.5, 0.5, 0.0, 0.0, 0.0)
    Q = synth.unit_rows(6, 384, 56)
    small = engine.searcher.search_batch(Q[:3], toks_of(3), 100, 0, w)
    big = engine.searcher.search_batch(Q, toks_of(6), 100, 0, w)
    for b in range(6):
        one = engine.searcher.search_batch(Q[b:b + 1], toks_of(1), 100, 0, w)
        if b < 3:
            assert np.array_equal(one.topk_rows()[0], small.topk_rows()[b])
            assert np.array_equal(one.columns[0], small.columns[b])
        assert np.array_equal(one.topk_rows()[0], big.topk_rows()[b])
        np.testing.assert_allclose(one.columns[0], big.columns[b], atol=TOL, rtol=0)


def test_degrade_to_zeros_conventions(world):
    # no BM25 blob -> zeros + bm25_active False (app/app_product_search.py:202,313);
    # no cross-encoder -> zeros (:275); query of stop words only -> zeros (:204)
    engine = SearchEngine(world["meta"], world["V"], None, normalize=False)
    qv = synth.unit_rows(1, 384, 57)[0]
    frame, _, dbg = engine.run_search("cat socks", 10, 20, 0.5, 0.3, 0.2, 0.0, 0.0, 20.0, False, 0, 8, 1.0,
                                      qvec=qv)
    assert dbg["bm25_active"] is False and np.all(frame["_bm25"] == 0) and np.all(frame["_rerank"] == 0)
    engine2 = SearchEngine(world["meta"], world["V"], world["blob"], normalize=False)
    frame, _, dbg = engine2.run_search("the of and", 10, 0, 0.5, 0.5, 0.0, 0.0, 0.0, 20.0, False, 0, 8, 1.0,
                                       qvec=qv)
    assert dbg["tokens"] == [] and np.all(frame["_bm25"] == 0)


def test_row_count_mismatch_is_a_hard_error(world):
    with pytest.raises(ValueError):
        SearchEngine(world["meta"].iloc[:-1], world["V"], None)


def test_missing_and_duplicate_skus_in_the_bm25_blob(world):
    # app flavour: sku dict, last duplicate wins, missing sku scores 0.0 (app/...:207-208)
    meta = world["meta"].iloc[:2000].reset_index(drop=True)
    V = world["V"][:2000]
    corpus = world["blob"]["corpus"][:2000]
    skus = meta["sku"].tolist()
    blob_skus = skus[:1500] + ["ZZZ%d" % i for i in range(400)] + [skus[3]] * 100
    blob = {"skus": blob_skus, "corpus": corpus}
    ora = BM25OkapiOracle(corpus)
    engine = SearchEngine(meta, V, blob, normalize=False)
    qv = synth.unit_rows(1, 384, 58)[0]
    cfg = CONFIGS["north_star_alpha"]
    want, _, _, _ = run_search_oracle(query="wireless cat", qvec=qv, meta=meta, V=V, bm25=ora,
                                      bm25_skus=blob_skus, **cfg)
    got, _, _ = engine.run_search("wireless cat", cfg["k"], 0, 0.5, 0.5, 0, 0, 0, 20.0, False, 0, 8, 1.0,
                                  qvec=qv)
    assert got["sku"].tolist() == want["sku"].tolist()
    np.testing.assert_allclose(got["_final"].values, want["_final"].values, atol=TOL, rtol=0)


def test_one_engine_shared_by_threads(world):
    """Streamlit runs each session in its own thread and shares cached objects
    (app/app_product_search.py:53,71,119): concurrent run_search calls on one engine must give the
    answers the sequential calls give."""
    import threading
    engine = SearchEngine(world["meta"], world["V"], world["blob"], normalize=False)
    Q = synth.unit_rows(12, 384, 77)
    queries = [QUERIES[i % len(QUERIES)] for i in range(12)]
    cfg = CONFIGS["hybrid"]

    def call(i):
        f, _, _ = engine.run_search(queries[i], cfg["k"], 0, cfg["w_dense"], cfg["w_bm25"], 0.0, cfg["w_prior"], 0.0,
                                    20.0, False, 0, cfg["min_reviews"], cfg["gate_penalty"], qvec=Q[i])
        return f["sku"].tolist(), f["_final"].values.copy()

    want = [call(i) for i in range(12)]
    got = [None] * 12
    def worker(i):
        got[i] = call(i)
    for rep in range(3):
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(12)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        for i in range(12):
            assert got[i][0] == want[i][0] and np.array_equal(got[i][1], want[i][1])


def test_cli_without_bm25_artefact_blends_in_float64_from_the_second_term(hip, world):
    """in the CLI, `cand["_bm25"] = 0.0` (no artefact, app/test.py:252) is a float64 column, so
    the blend is float64 from the second term on even when the float32 rerank column is active.  The fusion
    kernel fed the oracle's pool must reproduce every column bit for bit (rr_fuse_params.bm25_f64); end to end
    (the dense scores then come from K1, whose summation order is not BLAS's) the bar is 1e-5 and the frame's
    dtypes must be the reference's."""
    ce = FakeCrossEncoder()
    engine = SearchEngine(world["meta"], world["V"], None, cross_encoder=ce, normalize=False, flavour="cli")
    cfg = dict(k=100, rerank_k=60, w_dense=0.37, w_bm25=0.21, w_rerank=0.33, w_prior=0.17, w_best=0.0,
               prior_C=20.0, min_reviews=8, gate_penalty=1.0)
    differs = 0
    for seed, query in enumerate(QUERIES):
        qvec = synth.unit_rows(1, 384, 400 + seed)[0]
        want, _, _, cand = run_search_oracle(query=query, qvec=qvec, meta=world["meta"], V=world["V"], bm25=None,
                                             flavour="cli", rerank_fn=(lambda pairs: ce.predict(pairs)), **cfg)
        got, _, dbg = engine.run_search(query, cfg["k"], cfg["rerank_k"], cfg["w_dense"], cfg["w_bm25"],
                                        cfg["w_rerank"], cfg["w_prior"], cfg["w_best"], cfg["prior_C"], False, 0,
                                        cfg["min_reviews"], cfg["gate_penalty"], qvec=qvec)
        assert dbg["bm25_active"] is False
        assert got["_bm25"].dtype == want["_bm25"].dtype == np.float64
        assert got["_rerank"].dtype == want["_rerank"].dtype == np.float32
        g, w = got.set_index("sku")["_final"], want.set_index("sku")["_final"]
        common = [s_ for s_ in g.index if s_ in w.index]
        assert len(common) >= cfg["k"] - 2
        np.testing.assert_allclose(g.loc[common].values, w.loc[common].values, atol=TOL, rtol=0)
        # K3 alone on the oracle's pool: bit-exact, and NOT what the float32-first order gives
        pool = len(cand)
        rows = cand["_row"].values.astype(np.int64)[None, :]
        dense = np.ascontiguousarray((world["V"] @ qvec)[rows[0]].astype(np.float32)[None, :])
        rr = np.zeros((1, pool), dtype=np.float32)
        texts = cand["agg_text"].astype(str).str.slice(0, 2000).tolist()[:cfg["rerank_k"]]
        rr[0, :cfg["rerank_k"]] = ce.predict([(query, t) for t in texts])
        gate = np.ascontiguousarray(cand["_gate"].values.astype(np.float32)[None, :])
        wts = FusionWeights(cfg["w_dense"], cfg["w_bm25"], cfg["w_rerank"], cfg["w_prior"], cfg["w_best"],
                            cfg["prior_C"], cfg["min_reviews"], cfg["gate_penalty"], apply_trust=False)
        params = HybridSearcher.make_params(wts, cfg["k"], pool, pool, cfg["rerank_k"], bm25_f64=True)
        _, cols, _ = fuse_host(hip, engine.index, params, rows, dense, None, rr, gate)
        for j, c in enumerate(COLUMN_NAMES):
            if c in cand.columns:
                assert np.array_equal(cols[0, j], cand[c].values.astype(np.float64), equal_nan=True), f"column {c} differs"
        s32 = np.float32(cfg["w_dense"]) * cand["_dense"].values + np.float32(cfg["w_rerank"]) * cand["_rerank"].values
        alt = (s32.astype(np.float64) + cfg["w_prior"] * cand["_prior"].values).astype(np.float32)
        differs += int(np.any(alt != cand["_final"].values))
    assert differs > 0        # the float32-first order would not have produced these finals


def test_query_longer_than_64_tokens(world):
    """get_scores has no token limit (a pasted paragraph is a legal query); K2 takes such a query in
    several 64-token passes whose running float64 sum keeps the token order."""
    engine = SearchEngine(world["meta"], world["V"], world["blob"], normalize=False)
    words = [w for doc in world["blob"]["corpus"][:40] for w in doc][:150]
    query = " ".join(words)
    qv = synth.unit_rows(1, 384, 91)[0]
    cfg = CONFIGS["north_star_alpha"]
    got, _, dbg = engine.run_search(query, cfg["k"], 0, 0.5, 0.5, 0, 0, 0, 20.0, False, 0, 8, 1.0, qvec=qv)
    assert len(dbg["tokens"]) > 64
    want, _, _, cand = run_search_oracle(query=query, qvec=qv, meta=world["meta"], V=world["V"],
                                         bm25=world["ora_bm25"], bm25_skus=world["blob"]["skus"], **cfg)
    assert got["sku"].tolist() == want["sku"].tolist()
    assert np.array_equal(got["_bm25"].values, want["_bm25"].values)
    np.testing.assert_allclose(got["_final"].values, want["_final"].values, atol=TOL, rtol=0)


def test_token_lists_are_staged_on_every_call(world):
    """a caller that reuses and mutates one list object between calls must get the new ids."""
    engine = SearchEngine(world["meta"], world["V"], world["blob"], normalize=False)
    bm = engine.searcher.bm25
    Q = synth.unit_rows(2, 384, 92)
    w = FusionWeights(0.5, 0.5, 0.0, 0.0, 0.0)
    lists = [bm.term_ids(["wireless", "mug"]), bm.term_ids(["cat"])]
    a = engine.searcher.search_batch(Q, lists, 20, 0, w)
    lists[0] = bm.term_ids(["socks", "blue"])            # same outer list object, new content
    b = engine.searcher.search_batch(Q, lists, 20, 0, w)
    c = engine.searcher.search_batch(Q, [bm.term_ids(["socks", "blue"]), bm.term_ids(["cat"])], 20, 0, w)
    assert np.array_equal(b.bm25_raw, c.bm25_raw) and not np.array_equal(a.bm25_raw[0], b.bm25_raw[0])
    flat = (np.concatenate(lists).astype(np.int32), np.array([0, len(lists[0]), len(lists[0]) + len(lists[1])], np.int32))
    d = engine.searcher.search_batch(Q, flat, 20, 0, w)
    assert np.array_equal(d.bm25_raw, c.bm25_raw)


def test_module_level_run_search_shim_takes_the_eval_harness_call_shape(world, tmp_path):
    """evals/performance_metrics.py:266 calls `search_function(query, **config)` with the keys of
    evals/test_queries.py:255-312 (incl. use_snips / max_scan); the UI calls it positionally
    (app/app_product_search.py:402).  frontend.run_search is that function over one cached engine."""
    from review_recommender_amd import artifacts, frontend

    class Enc:                                   # the reference's tests mock the encoder the same way
        def encode(self, texts, normalize_embeddings=True):
            return synth.unit_rows(len(texts), 384, 321)

    artifacts.save_artifacts(tmp_path, world["meta"], world["V"], world["blob"])
    eng = frontend.configure(data_dir=tmp_path, encoder=Enc(), cross_encoder=FakeCrossEncoder())
    assert frontend.engine() is eng
    benchmark_config = {"k": 50, "rerank_k": 20, "w_dense": 0.4, "w_bm25": 0.2, "w_rerank": 0.3, "w_prior": 0.1,
                        "w_best": 0.0, "prior_C": 20.0, "use_snips": False, "max_scan": 50000, "min_reviews": 5,
                        "gate_penalty": 0.5}                       # "Hybrid + Rerank", evals/test_queries.py:297-311
    a, snips, dbg = frontend.run_search("yellow cat socks", **benchmark_config)
    b, _, _ = frontend.run_search("yellow cat socks", 50, 20, 0.4, 0.2, 0.3, 0.1, 0.0, 20.0, False, 50000, 5, 0.5)
    assert a.equals(b) and len(a) == 50 and snips == {} and dbg["pool"] == 150
    ce = FakeCrossEncoder()
    qv = synth.unit_rows(1, 384, 321)[0]
    from oracle.primitives import l2_normalize
    want, _, _, _ = run_search_oracle(query="yellow cat socks", qvec=qv, meta=world["meta"], V=l2_normalize(world["V"]),
                                      bm25=world["ora_bm25"], bm25_skus=world["blob"]["skus"],
                                      rerank_fn=(lambda pairs: ce.predict(pairs)),
                                      **{k_: v for k_, v in benchmark_config.items() if k_ not in ("use_snips", "max_scan")})
    assert a["sku"].tolist()[:10] == want["sku"].tolist()[:10]
    np.testing.assert_allclose(a["_final"].values, want["_final"].values, atol=TOL, rtol=0)
    assert list(a.columns) == [c for c in want.columns if c != "_row"]


def test_staged_batches_on_the_input_stream_give_the_same_answers(world):
    """HybridSearcher.stage_batch / release: a batch's queries and token ids go up on the searcher's input stream
    (ring buffers, events), the compute stream waits for `.ready` -- several batches in flight, more batches than ring
    slots, answers equal to the plain single-stream calls."""
    import torch
    from review_recommender_amd.sharded import ShardedSearcher
    engine = SearchEngine(world["meta"], world["V"], world["blob"], normalize=False)
    searcher = engine.searcher
    sh = ShardedSearcher(searcher, world["n"], 0, 1)
    w = FusionWeights(w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0, w_best=0.0, gate_penalty=1.0)
    rng = np.random.default_rng(5)
    words = synth.WORDS
    batches = []
    for b in range(11):                                  # more than the 4 query slots and the 8 token slots
        B = 1 + (b * 37) % 70
        q = torch.from_numpy(synth.unit_rows(B, 384, 100 + b)).pin_memory()
        terms = [searcher.bm25.term_ids(list(rng.choice(words, size=int(rng.integers(0, 6))))) for _ in range(B)]
        batches.append((q, terms))
    plain = []
    for q, terms in batches:
        r, c, o = sh.search_batch_dev(q.to(searcher.device), terms, 10, w)
        plain.append((r.cpu().numpy(), c.cpu().numpy(), o.cpu().numpy()))
    cur = torch.cuda.current_stream(searcher.device)
    outs = []
    for q, terms in batches:                             # nothing synchronises between the batches
        st = searcher.stage_batch(q, terms)
        cur.wait_event(st.ready)
        outs.append(sh.search_batch_dev(st.q, st.terms, 10, w))
        searcher.release(st)
    torch.cuda.synchronize()
    for (r0, c0, o0), (r, c, o) in zip(plain, outs):
        assert np.array_equal(r0, r.cpu().numpy()) and np.array_equal(o0, o.cpu().numpy())
        assert np.array_equal(c0, c.cpu().numpy(), equal_nan=True)
