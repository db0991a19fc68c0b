#!/usr/bin/env python3
"""Generates the end-to-end fixtures of SURVEY section 8(c)(2),(3) from the ORACLE (oracle/pipeline.py, oracle/dense.py).

    python tests/golden/make_pipeline_golden.py

** oracle-generated **: these files freeze what the build's statement-by-statement restatement produces, so that the
oracle cannot drift silently together with the product (both are builder code).  The app flavour's run_search cannot
be imported here (`import streamlit` at module level, app/app_product_search.py:9) and the reference's tests pin no
fused value; the CLI flavour of the same oracle IS pinned by a run of the reference's own app/test.py -- see
make_cli_golden.py / cli_search.json, which are the reference-run fixtures -- and the dense / primitive pieces
underneath by the reference's utils.py (make_golden.py).

  pipeline_10k.npz   N = 10 000 x 384 (BASELINE config 1 shape, seeds below): for every case the parameter tuple,
                     the query, the pool rows in pool order, the eight pool columns, the top-k rows and finals.
                     Cases: the four BENCHMARK_CONFIGS of evals/test_queries.py:255-312 + the UI defaults + the
                     north-star alpha sugar, both flavours (app / cli), x 3 queries; gate_penalty in {0, .3, .5, 1};
                     1 % NaN ratings; an all-NaN pool; exact dense ties; BM25 blob with missing / duplicate skus.
  dense_1M_top150.npz  top-150 rows + fp32 scores (BLAS matvec, utils.py:111-124 restated) + float64 dots at
                     those rows for 3 queries over the 1M x 384 seed recipe (matrix regenerated from the seed where
                     the fixture is used: 1.5 GB, not shipped).
"""
import json
import pathlib
import sys
import warnings

import numpy as np
import pandas as pd

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import dense as OD  # noqa: E402
from oracle.bm25 import BM25OkapiOracle  # noqa: E402
from oracle.pipeline import run_search_oracle  # noqa: E402
from review_recommender_amd import synth  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent
N = 10_000
COLS = ("_dense", "_bm25", "_prior", "_rerank", "_best", "_gate", "_trust", "_final")
CONFIGS = {   # evals/test_queries.py:255-312 + config.py:64-72 + SURVEY 8b sugar
    "dense_only": dict(k=20, rerank_k=0, w_dense=1.0, w_bm25=0.0, w_rerank=0.0, w_prior=0.0, w_best=0.0, prior_C=20.0, min_reviews=1, gate_penalty=0.0),
    "bm25_only": dict(k=20, rerank_k=0, w_dense=0.0, w_bm25=1.0, w_rerank=0.0, w_prior=0.0, w_best=0.0, prior_C=20.0, min_reviews=1, gate_penalty=0.0),
    "hybrid": dict(k=20, rerank_k=0, w_dense=0.5, w_bm25=0.3, w_rerank=0.0, w_prior=0.2, w_best=0.0, prior_C=20.0, min_reviews=5, gate_penalty=0.3),
    "hybrid_rerank": dict(k=50, rerank_k=20, w_dense=0.4, w_bm25=0.2, w_rerank=0.3, w_prior=0.1, w_best=0.0, prior_C=20.0, min_reviews=5, gate_penalty=0.5),
    "ui_defaults": dict(k=10, rerank_k=50, w_dense=0.55, w_bm25=0.20, w_rerank=0.20, w_prior=0.20, w_best=0.10, prior_C=20.0, min_reviews=8, gate_penalty=0.5),
    "north_star_alpha": dict(k=100, rerank_k=0, w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0, w_best=0.0, prior_C=20.0, min_reviews=8, gate_penalty=1.0),
}
QUERIES = ["wireless headphones for running", "yellow cat socks", "blue insulated coffee mug"]


def fake_rerank(pairs):
    """Deterministic stand-in for CrossEncoder.predict (the tests use the same function)."""
    return np.array([((len(t) * 7 + sum(map(ord, t[:20]))) % 97) / 9.7 - 4.0 for _, t in pairs], dtype=np.float32)


def world(nan_fraction=0.0, all_nan=False, ties=False, odd_blob=False):
    V = synth.unit_rows(N, 384, 1234)
    if ties:
        V[100:140] = V[7]                                  # 41 rows with exactly equal scores for every query
    n_rev, stars = synth.metadata(N, 2, nan_fraction=nan_fraction)
    if all_nan:
        stars = np.full(N, np.nan)
    texts = synth.text_corpus(N, 3, mean_len=25)
    meta = pd.DataFrame({"sku": synth.skus(N), "n_reviews": n_rev, "avg_stars": stars, "last_ts": np.arange(N),
                         "agg_text": texts})
    corpus = [t.split() for t in texts]
    skus = meta["sku"].tolist()
    if odd_blob:                                           # missing skus + a duplicated one (last wins, app/...:207)
        skus = skus[:8000] + ["ZZZ%d" % i for i in range(1900)] + [skus[3]] * 100
    return V, meta, corpus, skus


def main():
    cases, arrays = [], {}
    worlds = {"plain": world(), "nan1": world(nan_fraction=0.01), "allnan": world(all_nan=True),
              "ties": world(ties=True), "oddblob": world(odd_blob=True)}
    bm = {name: BM25OkapiOracle(w[2]) for name, w in worlds.items() if name in ("plain", "oddblob")}
    plan = []
    for cname in CONFIGS:
        for flavour in ("app", "cli"):
            for qi in range(3):
                plan.append(("plain", cname, flavour, qi, None))
    for gp in (0.0, 0.3, 0.5, 1.0):
        plan.append(("plain", "ui_defaults", "app", 1, gp))
    for wname in ("nan1", "allnan", "ties", "oddblob"):
        for cname in ("hybrid", "ui_defaults"):
            plan.append((wname, cname, "app", 1, None))
    plan.append(("ties", "north_star_alpha", "cli", 0, None))
    for wname, cname, flavour, qi, gp in plan:
        V, meta, corpus, skus = worlds[wname]
        cfg = dict(CONFIGS[cname])
        if gp is not None:
            cfg["gate_penalty"] = gp
        query = QUERIES[qi]
        qseed = 500 + qi
        qvec = V[7].copy() if wname == "ties" else synth.unit_rows(1, 384, qseed)[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                # nanmean of an all-NaN pool warns, like the reference
            top, _, dbg, cand = run_search_oracle(query=query, qvec=qvec, meta=meta, V=V,
                                                  bm25=bm["oddblob" if wname == "oddblob" else "plain"], bm25_skus=skus,
                                                  flavour=flavour, rerank_fn=fake_rerank, **cfg)
        i = len(cases)
        cases.append({"world": wname, "config": cname, "flavour": flavour, "query": query, "qvec_seed": qseed,
                      "qvec_is_row7": wname == "ties", "params": cfg, "pool": int(dbg["pool"]), "tokens": dbg["tokens"]})
        arrays[f"pool_rows_{i}"] = cand["_row"].values.astype(np.int64)
        arrays[f"pool_cols_{i}"] = np.stack([cand[c].values.astype(np.float64) if c in cand.columns
                                             else np.ones(len(cand)) for c in COLS])
        arrays[f"top_rows_{i}"] = top["_row"].values.astype(np.int64)
        arrays[f"top_final_{i}"] = top["_final"].values.astype(np.float32)
    arrays["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(OUT / "pipeline_10k.npz", **arrays)
    print("pipeline_10k.npz:", len(cases), "cases")

    # ---- (3) 1M x 384 dense fixture
    n = 1_000_000
    V = synth.unit_rows(n, 384, 1234)
    Q = synth.unit_rows(3, 384, 4321)
    rows, sims32, sims64 = [], [], []
    for q in Q:
        r, s = OD.cosine_similarity_search(q, V, 150)
        rows.append(r)
        sims32.append(s)
        sims64.append(V[r].astype(np.float64) @ q.astype(np.float64))
    gaps = []
    for q in Q:                                            # float64 gap between the 150th and 151st best row
        s64 = OD.sims_float64(V, q)
        top = np.sort(s64)[::-1][:151]
        gaps.append(float(top[149] - top[150]))
    np.savez_compressed(OUT / "dense_1M_top150.npz", n=n, seed_rows=1234, seed_queries=4321, rows=np.stack(rows),
                        scores_f32=np.stack(sims32), dots_f64=np.stack(sims64), boundary_gap_f64=np.array(gaps))
    print("dense_1M_top150.npz: boundary gaps", gaps)


if __name__ == "__main__":
    main()
