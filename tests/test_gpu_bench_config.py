"""The exact configuration bench.py times, checked against the oracle.

bench.py builds its corpus on the GPU (review-recommender_amd/device_corpus.py: torch sort / bincount
CSR, adopted by rr_bm25_create_dev) and runs ShardedSearcher.search_batch_dev over it.  Here the same
builder runs at BASELINE config 3's size (1M products, ~40M postings), and
  * BM25 at candidate rows through the adopted handle (both modes) is compared bit for bit with
    BM25CsrOracle over the same arrays and with a host-built rr_bm25_create handle;
  * one full bench step (256 queries, hybrid alpha = 0.5, k = 100, pool = 150) is checked by
    size-independent properties for every query and against the oracle pipeline for sampled queries.
"""
import ctypes as C

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import dense as OD
from oracle.bm25 import BM25CsrOracle
from oracle.pipeline import run_search_oracle
from oracle.primitives import minmax_normalize
from parity import assert_topk_matches
from review_recommender_amd import _lib, synth
from review_recommender_amd.bm25 import BM25Corpus
from review_recommender_amd.device_corpus import build_device_shard
from review_recommender_amd.engine import FusionWeights

pytestmark = pytest.mark.gpu
DOCS, VOCAB, BATCH, K, POOL = 1_000_000, 200_000, 256, 100, 150


@pytest.fixture(scope="module")
def shard():
    dev = torch.device("cuda", 0)
    sh = build_device_shard(torch, None, docs=DOCS, rank=0, world=1, dev=dev, vocab=VOCAB, doc_len=40)
    host = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in sh.bm25_arrays.items()}
    return sh, host


def test_device_built_csr_invariants(shard):
    sh, a = shard
    nnz = sh.stats["nnz"]
    assert 30 * DOCS < nnz < 41 * DOCS                       # ~40 tokens/doc; Zipf repeats inside a doc merge into tf
    assert a["post_indptr"][-1] == nnz == a["doc_indptr"][-1]
    # ascending docs per term / ascending terms per doc: what the binary searches of rr_bm25_at rely on
    d = np.diff(a["post_docs"].astype(np.int64))
    starts = a["post_indptr"][1:-1]
    starts = starts[(starts > 0) & (starts < nnz)]
    d[starts - 1] = 1
    assert np.all(d > 0)
    t = np.diff(a["doc_terms"].astype(np.int64))
    starts = a["doc_indptr"][1:-1]
    starts = starts[(starts > 0) & (starts < nnz)]
    t[starts - 1] = 1
    assert np.all(t > 0)
    # doc_len counts duplicates: sum of tf per doc
    tf_sum = np.add.reduceat(a["doc_tf"].astype(np.int64), a["doc_indptr"][:-1])
    assert np.array_equal(tf_sum, a["doc_len"].astype(np.int64))
    assert abs(a["avgdl"] - a["doc_len"].astype(np.int64).sum() / DOCS) < 1e-12


def test_adopted_bm25_handle_is_bit_exact_vs_oracle_and_host_built_handle(shard, hip):
    sh, a = shard
    ora = BM25CsrOracle(a["post_indptr"], a["post_docs"], a["post_tf"], a["doc_len"], a["idf"], a["avgdl"])
    nq = 48
    terms = synth.query_terms(nq, VOCAB, 7, sh.stats["df"])
    terms[3] = np.concatenate([terms[3], terms[3][:1], [-1]]).astype(np.int32)   # duplicate + unknown token
    rng = np.random.default_rng(8)
    rows = rng.integers(0, DOCS, (nq, POOL)).astype(np.int64)
    for q in range(nq):                                       # half the candidates really contain a query term
        t = int(terms[q][0])
        lo, hi = int(a["post_indptr"][t]), int(a["post_indptr"][t + 1])
        if hi > lo:
            take = a["post_docs"][lo + rng.integers(0, hi - lo, POOL // 2)]
            rows[q, :POOL // 2] = take
    want = np.stack([ora.get_scores(terms[q].tolist())[rows[q]] for q in range(nq)]).astype(np.float32)
    assert np.count_nonzero(want) > nq * POOL // 3
    searcher = sh.sharded.s
    rows_dev = torch.from_numpy(rows).cuda()
    for mode in ("forward", "postings"):
        got = searcher.bm25_at(terms, rows_dev, mode).cpu().numpy()
        assert np.array_equal(got, want), mode
    host_built = BM25Corpus(a["doc_indptr"], a["doc_terms"], a["doc_tf"], a["doc_len"], VOCAB, idf=a["idf"],
                            avgdl=a["avgdl"]).to_device(0)
    for mode in ("forward", "postings"):
        assert np.array_equal(host_built.scores_at_ids(terms, rows, mode), want), mode
    # get_scores (the rank_bm25 API) over the host-built handle: float64, all documents
    full = host_built.get_scores_ids(terms[0])
    assert np.array_equal(full, ora.get_scores(terms[0].tolist()))
    host_built.close()


class _IdTokenBM25:
    """rank_bm25-shaped adapter over the CSR oracle: tokens are "t<id>" strings."""
    def __init__(self, ora):
        self.ora = ora

    def get_scores(self, toks):
        return self.ora.get_scores([int(t[1:]) for t in toks])


def test_one_bench_step_at_1M_products_matches_the_oracle(shard):
    sh, a = shard
    check_one_bench_step(sh, a, DOCS, (0, 17, 101, 255))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_one_bench_step_at_10M_products_matches_the_oracle(dtype):
    """The headline configuration itself (BASELINE metric: 10M products, hybrid alpha = 0.5, k = 100, batches of 256):
    bench.py's own builder at its default size, one whole step checked by properties for every query and against the
    oracle pipeline for one sampled query of each query set of the scan launch.
    dtype "bf16" = BASELINE config 4's storage (10M x 384 bf16, batch 256) on the one GPU there is: the oracle is the
    fp32 pipeline over the bf16-rounded matrix upcast to fp32; config 4's row sharding itself is covered by
    tests/test_gpu_sharded.py (bitwise = unsharded) and tests/test_sharded_gloo.py."""
    dev = torch.device("cuda", 0)
    docs = 10_000_000
    sh = build_device_shard(torch, None, docs=docs, rank=0, world=1, dev=dev, vocab=VOCAB, doc_len=40, dtype=dtype)
    a = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in sh.bm25_arrays.items()}
    check_one_bench_step(sh, a, docs, (17, 201) if dtype == "f32" else (201,))
    info = sh.index.last_scan_info()
    assert info[0] == 5 and info[1] == 9 and info[2] == 256, info      # the 256-query launch bench.py's roofline names
    assert info[4] == 2                                                 # a bf16 stream: the filter plane / the bf16 rows
    sh.index.close()
    del sh, a
    torch.cuda.empty_cache()


def synthetic_pair_tokens(b, rows):
    """bench.SyntheticPairScorer's token recipe in numpy: the packed [CLS] q [SEP] text [SEP] ids of (query b, product row)."""
    out = []
    for r in rows:
        r = int(r)
        ln = 54 + (r * 2654435761 % 448) + 11
        pos = np.arange(ln, dtype=np.int64)
        hq = (b * 40503 + pos * 9973) % 29000 + 1000
        ht = (r * 7919 + pos * 104729) % 29000 + 1000
        tok = np.where(pos == 0, 101, np.where(pos <= 8, hq, np.where((pos == 9) | (pos == ln - 1), 102, ht)))
        out.append((tok.astype(np.int32), (pos > 9).astype(np.int32)))
    return out


def test_config5_step_matches_the_oracle(shard):
    """BASELINE config 5's step on one GPU: bench.py's builder (1M products here), batch 64, hybrid + cross-encoder rerank
    of the first 200 pool rows -> top 20, the K5 kernels in the reference's precision (fp32) over bench.py's own
    SyntheticPairScorer.  Properties for all 64 queries; for ONE sampled query the numpy oracle (oracle/cross_encoder.py)
    scores its 200 pairs and the oracle pipeline fuses them: skus identical (per tie band), `_final` within 1e-5."""
    import bench
    from oracle.cross_encoder import predict_oracle
    from parity import assert_ranking_matches
    sh, a = shard
    B, K5, RR, PL = 64, 20, 200, 200
    dev = torch.device("cuda", 0)
    scorer = bench.SyntheticPairScorer(torch, dev, seed=7, precision="fp32")
    w = FusionWeights(w_dense=0.4, w_bm25=0.2, w_rerank=0.3, w_prior=0.1, w_best=0.0, gate_penalty=1.0)    # bench.py --rerank-k
    Q = synth.unit_rows(B, 384, 4321)
    terms = synth.query_terms(B, VOCAB, 99, sh.stats["df"])
    q_dev = torch.from_numpy(Q).cuda()
    rows, cols, order = [t.cpu().numpy() for t in
                         sh.sharded.search_batch_dev(q_dev, terms, K5, w, rerank_k=RR, rerank_fn=scorer)]
    assert rows.shape == (B, PL) and cols.shape == (B, 8, PL) and order.shape == (B, K5)
    assert scorer.pairs == B * RR and 65 * scorer.pairs <= scorer.tokens <= 512 * scorer.pairs
    rr_n, final = cols[:, 3], cols[:, 7]
    assert np.all((rows >= 0) & (rows < DOCS)) and all(len(set(r.tolist())) == PL for r in rows)
    assert np.all(rr_n.max(axis=1) == 1.0) and np.all(rr_n.min(axis=1) == 0.0)          # min-max of 200 distinct logits
    top_final = np.take_along_axis(final, order.astype(np.int64), axis=1)
    assert np.all(np.diff(top_final, axis=1) <= 0) and all(len(set(o.tolist())) == K5 for o in order)
    rest = np.ones((B, PL), bool)
    np.put_along_axis(rest, order.astype(np.int64), False, axis=1)
    assert np.all(np.where(rest, final, -1).max(axis=1) <= top_final[:, -1])

    # one sampled query against the oracle: numpy BERT on its 200 pairs, the reference's pipeline around it
    b = 41
    V = sh.matrix.cpu().numpy()
    ora = BM25CsrOracle(a["post_indptr"], a["post_docs"], a["post_tf"], a["doc_len"], a["idf"], a["avgdl"])
    s = sh.sharded.s
    n_out, s_out, l_out = (torch.empty(DOCS, dtype=torch.float64, device="cuda") for _ in range(3))
    all_rows = torch.arange(DOCS, dtype=torch.int64, device="cuda")
    _lib.check(s.lib.rr_index_gather_meta_dev(s.index.handle, C.c_void_p(all_rows.data_ptr()), DOCS,
                                              C.c_void_p(n_out.data_ptr()), C.c_void_p(s_out.data_ptr()),
                                              C.c_void_p(l_out.data_ptr()), s._stream()), "gather_meta")
    torch.cuda.synchronize()
    skus = synth.skus(DOCS)
    # (the synthetic products have no text: the product's ROW stands in for it, which is what the token recipe hashes)
    meta = pd.DataFrame({"sku": skus, "n_reviews": n_out.cpu().numpy(), "avg_stars": s_out.cpu().numpy(),
                         "agg_text": [str(i) for i in range(DOCS)]})
    sd = synth.bert_state_dict(7, n_layers=6, n_labels=1)
    rerank = lambda pairs: predict_oracle(sd, synthetic_pair_tokens(b, [int(t) for _, t in pairs]), n_layers=6)
    query = " ".join(f"t{int(t)}" for t in terms[b])
    want, _, dbg, cand = run_search_oracle(query=query, qvec=Q[b], meta=meta, V=V, bm25=_IdTokenBM25(ora), bm25_skus=skus,
                                           k=K5, rerank_k=RR, w_dense=0.4, w_bm25=0.2, w_rerank=0.3, w_prior=0.1,
                                           w_best=0.0, prior_C=20.0, min_reviews=8, gate_penalty=1.0, rerank_fn=rerank)
    assert dbg["pool"] == PL
    got_rows = rows[b][order[b]]
    np.testing.assert_allclose(top_final[b], want["_final"].values, atol=1e-5, rtol=0)
    assert_ranking_matches(got_rows.tolist(), want["_row"].tolist(), want["_final"].values, 2e-5,
                           cand["_row"].tolist(), cand["_final"].values)
    # the rerank column of the pool, against the oracle's (min-max stretches the logits' error by 1 / span)
    pos = {int(r): j for j, r in enumerate(cand["_row"].values)}
    idx = [pos[int(r)] for r in rows[b]]
    assert np.abs(cols[b, 3] - cand["_rerank"].values[idx]).max() < 2e-5


def check_one_bench_step(sh, a, DOCS, samples):
    w = FusionWeights(w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0, w_best=0.0, gate_penalty=1.0)
    Q = synth.unit_rows(BATCH, 384, 4321)
    terms = synth.query_terms(BATCH, VOCAB, 99, sh.stats["df"])
    q_dev = torch.from_numpy(Q).cuda()
    rows, cols, order = [t.cpu().numpy() for t in sh.sharded.search_batch_dev(q_dev, terms, K, w)]
    assert rows.shape == (BATCH, POOL) and cols.shape == (BATCH, 8, POOL) and order.shape == (BATCH, K)
    dense_n, bm_n, final = cols[:, 0], cols[:, 1], cols[:, 7]
    # properties that hold at any size, every query
    assert np.all((rows >= 0) & (rows < DOCS))
    assert all(len(set(r.tolist())) == POOL for r in rows)
    assert np.all(np.diff(dense_n, axis=1) <= 0) and np.all(dense_n[:, 0] > 0.999) and np.all(dense_n[:, -1] == 0)
    assert np.all((bm_n >= 0) & (bm_n <= 1)) and np.all(final >= 0) and np.all(final <= 1.0 + 1e-6)
    top_final = np.take_along_axis(final, order.astype(np.int64), axis=1)
    assert np.all(np.diff(top_final, axis=1) <= 0)
    assert all(len(set(o.tolist())) == K for o in order)
    rest = np.ones((BATCH, POOL), bool)
    np.put_along_axis(rest, order.astype(np.int64), False, axis=1)
    assert np.all(np.where(rest, final, -1).max(axis=1) <= top_final[:, -1])     # nothing better was left out
    # the same batch again gives the same bits (no state carried between steps)
    again = [t.cpu().numpy() for t in sh.sharded.search_batch_dev(q_dev, terms, K, w)]
    assert np.array_equal(again[0], rows) and np.array_equal(again[1], cols) and np.array_equal(again[2], order)

    # sampled queries against the oracle pipeline (reference semantics: sku dict over all N, app flavour)
    V = sh.matrix.float().cpu().numpy()                      # (bf16 storage: the rounded matrix upcast to fp32)
    ora = BM25CsrOracle(a["post_indptr"], a["post_docs"], a["post_tf"], a["doc_len"], a["idf"], a["avgdl"])
    # metadata as the builder generated it: re-derive from the device index through K3's own gather
    n_out = torch.empty(DOCS, dtype=torch.float64, device="cuda")
    s_out = torch.empty(DOCS, dtype=torch.float64, device="cuda")
    l_out = torch.empty(DOCS, dtype=torch.float64, device="cuda")
    all_rows = torch.arange(DOCS, dtype=torch.int64, device="cuda")
    s = sh.sharded.s
    _lib.check(s.lib.rr_index_gather_meta_dev(s.index.handle, C.c_void_p(all_rows.data_ptr()), DOCS,
                                              C.c_void_p(n_out.data_ptr()), C.c_void_p(s_out.data_ptr()),
                                              C.c_void_p(l_out.data_ptr()), s._stream()), "gather_meta")
    torch.cuda.synchronize()
    skus = synth.skus(DOCS)
    meta = pd.DataFrame({"sku": skus, "n_reviews": n_out.cpu().numpy(), "avg_stars": s_out.cpu().numpy(),
                         "agg_text": ""})
    bm = _IdTokenBM25(ora)
    for b in samples:
        sims64 = OD.sims_float64(V, Q[b])
        raw_dense = (V[rows[b]] @ Q[b]).astype(np.float32)
        assert_topk_matches(rows[b], raw_dense, sims64, POOL)
        query = " ".join(f"t{int(t)}" for t in terms[b])
        want, _, dbg, cand = run_search_oracle(query=query, qvec=Q[b], meta=meta, V=V, bm25=bm, bm25_skus=skus,
                                               k=K, rerank_k=0, w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0,
                                               w_best=0.0, prior_C=20.0, min_reviews=8, gate_penalty=1.0)
        assert dbg["tokens"] == [f"t{int(t)}" for t in terms[b]]
        got_rows = rows[b][order[b]]
        wf = want["_final"].values
        np.testing.assert_allclose(top_final[b], wf, atol=1e-5, rtol=0)
        wr = want["_row"].values
        for f in np.unique(wf):                                   # equal finals may come in any order
            sel = wf == f
            if f == wf[-1]:
                assert set(got_rows[sel]) <= set(cand.loc[cand["_final"].values == f, "_row"])
            else:
                assert set(got_rows[sel]) == set(wr[sel])
        # raw BM25 of the pool: bit-exact float32 of the float64 oracle scores
        bm_raw = ora.get_scores(terms[b].tolist())[rows[b]].astype(np.float32)
        assert np.array_equal(cols[b, 1].astype(np.float32), minmax_normalize(bm_raw))
