"""Committed end-to-end fixtures (SURVEY section 8c (2),(3); tests/golden/make_pipeline_golden.py):
** oracle-generated, reference-unpinned ** -- they freeze the oracle's output so it cannot drift together with
the product.  CPU: the live oracle must still reproduce the files.  GPU: the HIP path against the FILES (no live
oracle run): pool membership / order, all eight pool columns and the top-k, for every BENCHMARK_CONFIGS case,
both flavours, the four gate penalties, NaN ratings, an all-NaN pool, exact ties and an odd BM25 blob."""
import json
import warnings

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN
from review_recommender_amd import synth

N = 10_000
COLS = ("_dense", "_bm25", "_prior", "_rerank", "_best", "_gate", "_trust", "_final")


def fake_rerank(pairs):
    return np.array([((len(t) * 7 + sum(map(ord, t[:20]))) % 97) / 9.7 - 4.0 for _, t in pairs], dtype=np.float32)


class FakeCE:
    def predict(self, pairs, batch_size=64, show_progress_bar=False):
        return fake_rerank(pairs)


def load():
    fx = np.load(GOLDEN / "pipeline_10k.npz")
    return fx, json.loads(bytes(fx["cases_json"]).decode())


def make_world(name):
    V = synth.unit_rows(N, 384, 1234)
    if name == "ties":
        V[100:140] = V[7]
    n_rev, stars = synth.metadata(N, 2, nan_fraction=0.01 if name == "nan1" else 0.0)
    if name == "allnan":
        stars = np.full(N, np.nan)
    texts = synth.text_corpus(N, 3, mean_len=25)
    meta = pd.DataFrame({"sku": synth.skus(N), "n_reviews": n_rev, "avg_stars": stars, "last_ts": np.arange(N),
                         "agg_text": texts})
    corpus = [t.split() for t in texts]
    skus = meta["sku"].tolist()
    if name == "oddblob":
        skus = skus[:8000] + ["ZZZ%d" % i for i in range(1900)] + [skus[3]] * 100
    return V, meta, corpus, skus


def qvec_of(case, V):
    return V[7].copy() if case["qvec_is_row7"] else synth.unit_rows(1, 384, case["qvec_seed"])[0]


def test_oracle_still_reproduces_the_committed_pipeline_fixtures():
    from oracle.bm25 import BM25OkapiOracle
    from oracle.pipeline import run_search_oracle
    fx, cases = load()
    assert len(cases) >= 45
    V, meta, corpus, skus = make_world("plain")
    bm = BM25OkapiOracle(corpus)
    picked = [i for i, c in enumerate(cases) if c["world"] == "plain"][::4]
    assert len(picked) >= 9
    for i in picked:
        c = cases[i]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            top, _, dbg, cand = run_search_oracle(query=c["query"], qvec=qvec_of(c, V), meta=meta, V=V, bm25=bm,
                                                  bm25_skus=skus, flavour=c["flavour"], rerank_fn=fake_rerank, **c["params"])
        assert dbg["pool"] == c["pool"] and dbg["tokens"] == c["tokens"]
        assert np.array_equal(cand["_row"].values, fx[f"pool_rows_{i}"])
        for j, col in enumerate(COLS):
            if col in cand.columns:
                assert np.array_equal(cand[col].values.astype(np.float64), fx[f"pool_cols_{i}"][j], equal_nan=True), (i, col)
        assert np.array_equal(top["_row"].values, fx[f"top_rows_{i}"])
        assert np.array_equal(top["_final"].values, fx[f"top_final_{i}"], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("world_name", ["plain", "nan1", "allnan", "ties", "oddblob"])
def test_hip_path_matches_the_committed_pipeline_fixtures(world_name):
    from review_recommender_amd.engine import SearchEngine
    fx, cases = load()
    V, meta, corpus, skus = make_world(world_name)
    blob = {"skus": skus, "corpus": corpus}
    engines = {}
    mine = [i for i, c in enumerate(cases) if c["world"] == world_name]
    assert mine
    row_of = {s: r for r, s in enumerate(meta["sku"])}
    for i in mine:
        c = cases[i]
        fl = c["flavour"]
        if fl not in engines:
            engines[fl] = SearchEngine(meta, V, blob, cross_encoder=FakeCE(), normalize=False, flavour=fl)
        p = c["params"]
        got, _, dbg = engines[fl].run_search(c["query"], p["k"], p["rerank_k"], p["w_dense"], p["w_bm25"], p["w_rerank"],
                                             p["w_prior"], p["w_best"], p["prior_C"], False, 0, p["min_reviews"],
                                             p["gate_penalty"], qvec=qvec_of(c, V))
        assert dbg["pool"] == c["pool"] and dbg["tokens"] == c["tokens"]
        want_rows, want_final = fx[f"top_rows_{i}"], fx[f"top_final_{i}"]
        got_rows = np.array([row_of[s] for s in got["sku"]])
        gf = got["_final"].values
        assert len(gf) == len(want_final)
        np.testing.assert_allclose(gf, want_final, atol=1e-5, rtol=0, equal_nan=True)
        # rows: exact, except inside groups of equal (or NaN) finals, whose order the reference leaves unspecified
        pool_rows, pool_final = fx[f"pool_rows_{i}"], fx[f"pool_cols_{i}"][7]
        wf = np.where(np.isnan(want_final), -np.inf, want_final)
        for f in np.unique(wf):
            sel = wf == f
            allowed = set(pool_rows[np.where(np.isnan(pool_final), -np.inf, pool_final).astype(np.float32) == np.float32(f)].tolist())
            if f == wf[-1]:
                assert set(got_rows[sel].tolist()) <= allowed, (i, c["config"])
            else:
                assert set(got_rows[sel].tolist()) == set(want_rows[sel].tolist()), (i, c["config"])
        # the columns of the returned rows against the fixture's pool columns
        pos = {int(r): j for j, r in enumerate(pool_rows)}
        idx = [pos[int(r)] for r in got_rows]
        for j, col in enumerate(COLS):
            if col in got.columns:
                np.testing.assert_allclose(got[col].values.astype(np.float64), fx[f"pool_cols_{i}"][j][idx],
                                           atol=1e-5, rtol=0, equal_nan=True, err_msg=f"case {i} column {col}")


@pytest.mark.gpu
def test_one_million_row_dense_fixture():
    """top-150 of three queries over the 1M x 384 seed recipe (matrix regenerated from the seed)."""
    from review_recommender_amd.index import ProductIndex
    fx = np.load(GOLDEN / "dense_1M_top150.npz")
    V = synth.unit_rows(int(fx["n"]), 384, int(fx["seed_rows"]))
    Q = synth.unit_rows(3, 384, int(fx["seed_queries"]))
    ix = ProductIndex.from_rows(V)
    rows, scores = ix.dense_topk(Q, 150)
    for i in range(3):
        np.testing.assert_allclose(scores[i], fx["scores_f32"][i], atol=1e-5, rtol=0)
        assert set(rows[i][:149].tolist()) <= set(fx["rows"][i].tolist()) or fx["boundary_gap_f64"][i] < 4e-7
        d64 = fx["dots_f64"][i]
        gaps = -np.diff(d64)
        same = rows[i] == fx["rows"][i]
        for j in np.nonzero(~same)[0]:                     # swaps only where neighbours are within fp32 rounding
            lo, hi = max(j - 1, 0), min(j, len(gaps) - 1)
            assert min(gaps[lo], gaps[hi]) < 4e-7, (i, j)
        if fx["boundary_gap_f64"][i] > 4e-7:
            assert set(rows[i].tolist()) == set(fx["rows"][i].tolist())
    # the batched path gives the same bits
    rb, sb = ix.dense_topk(np.repeat(Q, 3, axis=0), 150)
    for i in range(3):
        assert np.array_equal(rb[3 * i], rows[i]) and np.array_equal(sb[3 * i], scores[i])
    ix.close()
