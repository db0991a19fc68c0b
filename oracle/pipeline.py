"""Oracle: the search pipeline (restatement of run_search and the CLI search).

TEST INFRASTRUCTURE -- never imported by the product path.

Follows, statement by statement:
  * ``run_search``  app/app_product_search.py:245-317   (flavour="app")
  * ``search(args)`` app/test.py:228-309                (flavour="cli")
minus Streamlit / hub I/O: the artefacts the reference loads (meta frame,
normalised matrix, BM25 object + sku list, encoders) are injected, the way the
reference's own integration test injects mocks (tests/test_integration.py:41-48).

The same numpy / pandas expressions are used as in the reference so that every
dtype promotion (float32 weak-scalar products, the float64 ``_prior`` column,
the float64 ``_rerank`` column when rerank_k == 0) is decided by numpy itself.

Pinning: the **cli flavour is pinned by a run of the reference's own app/test.py**
(tests/golden/make_cli_golden.py -> cli_search.json / cli_helpers.json; the module
imports here with numpy + pandas, its three loader hooks stubbed): this file
reproduces every committed result exactly (tests/test_cli_golden.py).  The BM25
arithmetic inside those runs is oracle/bm25.py (rank_bm25 is absent: unpinned).
The **app flavour stays unpinned** beyond the pinned primitives and the statements it
shares with the cli flavour: app/app_product_search.py does ``import streamlit`` at
module level (not installed), and the reference's tests pin no fused value.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from . import primitives as P
from .dense import cosine_similarity_search

APP_POOL_FLOOR = 150   # app/app_product_search.py:253
CLI_POOL_FLOOR = 100   # app/test.py:238
APP_TRUST_SAT = 80     # app/app_product_search.py:303


def bm25_for_candidates_app(bm25, bm25_skus: Optional[Sequence[str]], query: str,
                            cand_skus: Sequence[str]) -> np.ndarray:
    """app/app_product_search.py:201-208: full-corpus scores -> sku dict -> gather."""
    if bm25 is None:
        return np.zeros(len(cand_skus), dtype=np.float32)
    toks = P.tokenize_query(query)
    if not toks:
        return np.zeros(len(cand_skus), dtype=np.float32)
    all_scores = np.array(bm25.get_scores(toks), dtype=np.float32)
    by_sku = {bm25_skus[i]: all_scores[i] for i in range(len(bm25_skus))}
    return np.array([by_sku.get(str(s), 0.0) for s in cand_skus], dtype=np.float32)


def bm25_for_candidates_cli(bm25, bm25_skus: Sequence[str], meta_skus: Sequence[str],
                            query: str, top_idx: np.ndarray) -> np.ndarray:
    """app/test.py:159-173: permutation by sku (None if any missing), then index."""
    pos = {s: i for i, s in enumerate(bm25_skus)}
    try:
        order = [pos[s] for s in meta_skus]
    except KeyError:
        order = None
    all_scores = np.array(bm25.get_scores(P.tokenize_query(query)), dtype=np.float32)
    if order is not None:
        all_scores = all_scores[np.array(order)]
    return all_scores[top_idx]


def best_snippets_oracle(reviews: pd.DataFrame, embeddings: np.ndarray, qvec: np.ndarray,
                         cand_skus: Sequence[str], max_rows: int, text_cut: int = 600) -> Dict[str, Dict]:
    """app/app_product_search.py:320-370 (text_cut 600) / app/test.py:181-215 (400) over an
    in-memory review table instead of the parquet the reference re-reads per query."""
    try:
        meta = reviews[["sku", "text", "stars"]]
        sel = meta["sku"].astype(str).isin(set(cand_skus))
        sub_meta = meta[sel]
        if sub_meta.empty:
            return {}
        emb = embeddings[sub_meta.index.values]
        if len(sub_meta) > max_rows:
            sub_meta = sub_meta.iloc[:max_rows]
            emb = emb[:max_rows]
        E = np.stack(list(emb)).astype(np.float32)
        En = P.l2_normalize(E, axis=1)
        sims = En @ qvec
        sub_meta = sub_meta.reset_index(drop=True)
        sub_meta["__sim"] = sims
        best = {}
        for sku, grp in sub_meta.groupby("sku"):
            j = int(grp["__sim"].values.argmax())
            row = grp.iloc[j]
            best[str(sku)] = {"score": float(row["__sim"]), "text": str(row["text"])[:text_cut],
                              "stars": float(row.get("stars", np.nan))}
        return best
    except Exception:          # the reference swallows every failure and returns {} (:366-370)
        return {}


def run_search_oracle(
    *, query: str, qvec: np.ndarray, meta: pd.DataFrame, V: np.ndarray,
    bm25=None, bm25_skus: Optional[Sequence[str]] = None,
    k: int = 10, rerank_k: int = 0,
    w_dense: float = 0.55, w_bm25: float = 0.20, w_rerank: float = 0.20,
    w_prior: float = 0.20, w_best: float = 0.10, prior_C: float = 20.0,
    min_reviews: int = 8, gate_penalty: float = 0.5,
    rerank_fn: Optional[Callable[[List[Tuple[str, str]]], np.ndarray]] = None,
    flavour: str = "app", use_snips: bool = False, max_scan: int = 0,
    reviews: Optional[Tuple[pd.DataFrame, np.ndarray]] = None,
) -> Tuple[pd.DataFrame, Dict, Dict]:
    """Returns (top-k frame, snips, dbg, full pool frame).  ``reviews`` = (review table,
    embeddings) stands in for reviews_with_embeddings.parquet; without it (or with
    use_snips False) ``_best`` is the all-zero column of app/app_product_search.py:288-294."""
    assert flavour in ("app", "cli")
    app = flavour == "app"
    pool = max(k, rerank_k, APP_POOL_FLOOR if app else CLI_POOL_FLOOR)

    cand_idx, dense_scores = cosine_similarity_search(qvec, V, pool)
    cand = meta.iloc[cand_idx].reset_index(drop=True).copy()
    cand["_dense"] = P.minmax_normalize(dense_scores.astype(np.float32),
                                        empty_passthrough=not app)

    if app:
        raw = bm25_for_candidates_app(bm25, bm25_skus, query,
                                      cand["sku"].astype(str).tolist())
        cand["_bm25"] = P.minmax_normalize(raw)
    elif bm25 is not None:
        raw = bm25_for_candidates_cli(bm25, bm25_skus,
                                      meta["sku"].astype(str).tolist(), query, cand_idx)
        cand["_bm25"] = P.minmax_normalize(raw, empty_passthrough=True)
    else:
        cand["_bm25"] = 0.0

    n = pd.to_numeric(cand.get("n_reviews", pd.Series([np.nan] * len(cand))),
                      errors="coerce").fillna(0).values
    r = pd.to_numeric(cand.get("avg_stars", pd.Series([np.nan] * len(cand))),
                      errors="coerce").fillna(np.nan).values
    prior_rating = P.bayesian_prior(r, n, prior_strength=prior_C)
    prior_volume = np.log1p(n) / (np.log1p(n).max() + 1e-9)
    cand["_prior"] = P.minmax_normalize(prior_rating, empty_passthrough=not app) * 0.7 \
        + 0.3 * prior_volume

    if rerank_k > 0:
        rr_k = min(rerank_k, len(cand))
        texts = cand["agg_text"].astype(str).str.slice(0, 2000).tolist()[:rr_k]
        if rerank_fn is None:
            rr = np.zeros(rr_k, dtype=np.float32)
        else:
            rr = np.array(rerank_fn([(query, t) for t in texts]), dtype=np.float32)
        z = np.zeros(len(cand), dtype=np.float32)
        z[:rr_k] = P.minmax_normalize(rr, empty_passthrough=not app)
        cand["_rerank"] = z
    else:
        cand["_rerank"] = 0.0

    # app/app_product_search.py:285-294 (CLI: app/test.py:273-289)
    snips = {}
    if use_snips and reviews is not None:
        snips = best_snippets_oracle(reviews[0], reviews[1], qvec, cand["sku"].astype(str).tolist(),
                                     max_rows=max_scan, text_cut=600 if app else 400)
    best_contrib = np.zeros(len(cand), dtype=np.float32)
    if snips:
        for i, sk in enumerate(cand["sku"].astype(str).tolist()):
            v = snips.get(sk, {}).get("score")
            if v is not None:
                best_contrib[i] = v
        best_contrib = P.minmax_normalize(best_contrib, empty_passthrough=not app)
    cand["_best"] = best_contrib

    groups = P.build_gate_groups(query)
    gate = [P.calculate_gate_factor(t, groups, penalty=gate_penalty)[0]
            for t in cand["agg_text"].astype(str).str.slice(0, 6000).tolist()]
    cand["_gate"] = np.array(gate, dtype=np.float32)
    if app:
        cand["_trust"] = P.trust_score_from_reviews(n, min_reviews=min_reviews,
                                                    saturation=APP_TRUST_SAT)

    final = (w_dense * cand["_dense"].values + w_bm25 * cand["_bm25"].values
             + w_rerank * cand["_rerank"].values + w_prior * cand["_prior"].values
             + w_best * cand["_best"].values).astype(np.float32)
    if app:
        final = final * cand["_trust"].values * cand["_gate"].values
    else:
        final = final * cand["_gate"].values
    cand["_final"] = final
    cand["_row"] = cand_idx  # oracle-only helper column: global row of each hit

    out = cand.sort_values("_final", ascending=False).head(k).reset_index(drop=True)
    dbg = {"bm25_active": bm25 is not None, "tokens": P.tokenize_query(query),
           "groups": [list(g) for g in groups], "pool": pool}
    return out, snips, dbg, cand


def cli_rows(frame: pd.DataFrame, snippets: Optional[Dict] = None) -> List[Dict]:
    """app/test.py:312-328: the CLI's JSON row schema (4-dp rounding)."""
    rows = []
    for _, row in frame.iterrows():
        snip = snippets.get(str(row["sku"])) if snippets else None
        rows.append({
            "sku": str(row["sku"]),
            "score": round(float(row["_final"]), 4),
            "dense": round(float(row["_dense"]), 4),
            "bm25": round(float(row["_bm25"]), 4),
            "rerank": round(float(row["_rerank"]), 4),
            "prior": round(float(row["_prior"]), 4),
            "bestrev": round(float(row["_best"]), 4),
            "n_reviews": int(row.get("n_reviews", 0)
                             if pd.notna(row.get("n_reviews", np.nan)) else 0),
            "avg_stars": round(float(row.get("avg_stars", np.nan)), 2)
            if pd.notna(row.get("avg_stars", np.nan)) else None,
            "snippet_stars": float(snip["stars"]) if snip and snip.get("stars") is not None else None,
            "snippet": snip["text"] if snip else None,
        })
    return rows
