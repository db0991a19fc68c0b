"""Seeded inputs of the CLI fixtures (tests/golden/cli_search.json): shared by the generator, which runs the
REFERENCE's app/test.py on them in the build container, and by the tests, which rebuild the same inputs from the
seeds on any machine.  Test infrastructure."""
import numpy as np
import pandas as pd

from review_recommender_amd import synth

N = 10_000
DIM = 384
QUERIES = ["wireless headphones for running", "yellow cat socks", "blue insulated coffee mug",
           "the of and", "zzzqx unobtainium"]                # [3]: only stop words; [4]: nothing in the vocabulary
# app/test.py:346-361 flags; the CLI has no min_reviews / trust
CONFIGS = {   # evals/test_queries.py:255-312 (weights), app/test.py:349-360 (defaults), SURVEY 8b sugar
    "dense_only": dict(k=20, rerank_k=0, w_dense=1.0, w_bm25=0.0, w_rerank=0.0, w_prior=0.0, w_best=0.0, prior_C=20.0, gate_penalty=0.0),
    "bm25_only": dict(k=20, rerank_k=0, w_dense=0.0, w_bm25=1.0, w_rerank=0.0, w_prior=0.0, w_best=0.0, prior_C=20.0, gate_penalty=0.0),
    "hybrid": dict(k=20, rerank_k=0, w_dense=0.5, w_bm25=0.3, w_rerank=0.0, w_prior=0.2, w_best=0.0, prior_C=20.0, gate_penalty=0.3),
    "hybrid_rerank": dict(k=50, rerank_k=20, w_dense=0.4, w_bm25=0.2, w_rerank=0.3, w_prior=0.1, w_best=0.0, prior_C=20.0, gate_penalty=0.5),
    "cli_defaults": dict(k=10, rerank_k=50, w_dense=0.55, w_bm25=0.15, w_rerank=0.15, w_prior=0.10, w_best=0.05, prior_C=20.0, gate_penalty=0.5),
    "north_star_alpha": dict(k=100, rerank_k=0, w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0, w_best=0.0, prior_C=20.0, gate_penalty=1.0),
    "wide_pool": dict(k=120, rerank_k=200, w_dense=0.5, w_bm25=0.2, w_rerank=0.2, w_prior=0.1, w_best=0.0, prior_C=5.0, gate_penalty=0.7),
}
WORLDS = ("plain", "nan1", "allnan", "ties", "oddblob", "permuted", "noblob", "scaled", "reviews")
N_REVIEWS = 6000


def fake_rerank(pairs):
    """Deterministic stand-in for CrossEncoder.predict (the reference's own tests mock it too,
    tests/test_integration.py:41-48)."""
    return np.array([((len(t) * 7 + sum(map(ord, t[:20]))) % 97) / 9.7 - 4.0 for _, t in pairs], dtype=np.float32)


class FakeCrossEncoder:
    def predict(self, pairs, batch_size=64, show_progress_bar=False):
        return fake_rerank(pairs)


def make_world(name):
    """-> dict(emb (N, 384) float32 as written to product_emb.npy, meta frame, blob or None, reviews or None)."""
    assert name in WORLDS, name
    V = synth.unit_rows(N, DIM, 1234)
    if name == "ties":
        V[100:140] = V[7]                                  # 41 rows with exactly equal scores for every query
    if name == "scaled":                                   # rows NOT unit length: load_product_index normalises (app/test.py:144)
        V = (V * np.random.default_rng(77).uniform(0.5, 2.0, (N, 1)).astype(np.float32)).astype(np.float32)
    n_rev, stars = synth.metadata(N, 2, nan_fraction=0.01 if name == "nan1" else 0.0)
    if name == "allnan":
        stars = np.full(N, np.nan)
    texts = synth.text_corpus(N, 3, mean_len=25)
    meta = pd.DataFrame({"sku": synth.skus(N), "n_reviews": n_rev, "avg_stars": stars, "last_ts": np.arange(N),
                         "agg_text": texts})
    corpus = [t.split() for t in texts]
    skus = meta["sku"].tolist()
    blob = {"skus": skus, "corpus": corpus, "tokenizer": "simple_en_v1"}
    if name == "oddblob":     # a meta sku missing from the blob: ensure_same_order -> None, scores used UNPERMUTED (app/test.py:159-173)
        blob["skus"] = skus[:8000] + ["ZZZ%d" % i for i in range(1900)] + [skus[3]] * 100
    if name == "permuted":    # the blob in another row order than the metadata: scores are permuted back by sku
        perm = np.random.default_rng(91).permutation(N)
        blob = {"skus": [skus[i] for i in perm], "corpus": [corpus[i] for i in perm], "tokenizer": "simple_en_v1"}
    if name == "noblob":
        blob = None
    reviews = None
    if name == "reviews":
        rng = np.random.default_rng(55)
        owner = rng.integers(0, N + 300, N_REVIEWS)                     # some reviews belong to unknown skus
        E = (rng.standard_normal((N_REVIEWS, DIM)) * rng.uniform(0.5, 3.0, (N_REVIEWS, 1))).astype(np.float32)
        known = owner < N
        E[known] = (V[owner[known]] * 4 + E[known] * 0.05).astype(np.float32)
        frame = pd.DataFrame({"sku": [f"B{o:09d}" for o in owner],
                              "text": [f"review {i} " + "x" * (i % 700) for i in range(N_REVIEWS)],
                              "stars": rng.integers(1, 6, N_REVIEWS).astype(np.float64)})
        reviews = (frame, E)
    return {"emb": V, "meta": meta, "blob": blob, "reviews": reviews}


def qvec_of(case, emb):
    """The query vector the stubbed encoder returned for this case."""
    if case["qvec_is_row7"]:
        v = emb[7].astype(np.float32)
        return (v / np.maximum(np.linalg.norm(v), 1e-12)).astype(np.float32)
    return synth.unit_rows(1, DIM, case["qvec_seed"])[0]


def plan():
    """[(world, config name, query index, overrides)] -- the cases of cli_search.json, in file order."""
    out = []
    for cname in CONFIGS:
        for qi in range(3):
            out.append(("plain", cname, qi, {}))
    for gp in (0.0, 0.3, 0.5, 1.0):
        out.append(("plain", "cli_defaults", 1, {"gate_penalty": gp}))
    out.append(("plain", "hybrid", 3, {}))                  # no query token survives the stop list
    out.append(("plain", "hybrid", 4, {}))                  # no query token is in the vocabulary
    for wname in ("nan1", "allnan", "ties", "oddblob", "permuted", "noblob", "scaled"):
        for cname in ("hybrid", "cli_defaults"):
            out.append((wname, cname, 1, {}))
    out.append(("ties", "north_star_alpha", 0, {}))
    out.append(("permuted", "bm25_only", 0, {}))
    out.append(("oddblob", "bm25_only", 2, {}))
    # snippets (reviews_with_embeddings.parquet present, --no-snippets not given): `_bestrev` carries weight
    for ms in (1_000_000, 150):
        out.append(("reviews", "cli_defaults", 1, {"max_reviews_scan": ms, "w_best": 0.25}))
    out.append(("reviews", "cli_defaults", 0, {"no_snippets": True}))
    return out
