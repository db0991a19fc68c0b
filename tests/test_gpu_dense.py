"""K1 parity on the GPU, through the C ABI (rr_dense_topk / rr_index_*): the HIP scan +
selection against the oracle (numpy restatement of utils.py:111-124) and the golden
vectors the reference's own utils.py produced."""
import numpy as np
import pytest

from oracle import dense as OD
from parity import assert_topk_matches, min_gap
from review_recommender_amd import synth
from review_recommender_amd.engine import cosine_similarity_search
from review_recommender_amd.index import ProductIndex

pytestmark = pytest.mark.gpu


def check_against_oracle(V, Q, pool, index=None):
    ix = index or ProductIndex(V)
    rows, scores = ix.dense_topk(Q, pool)
    eff = min(pool, V.shape[0])
    assert rows.shape == (len(Q), eff) and rows.dtype == np.int64 and scores.dtype == np.float32
    for i, q in enumerate(Q):
        ref64 = OD.sims_float64(V, q)
        assert_topk_matches(rows[i], scores[i], ref64, eff)
        # and against the reference-order oracle wherever its top-k has clear gaps
        o_rows, o_sims = OD.cosine_similarity_search(q, V, pool)
        if min_gap(ref64, eff) > 4e-7:
            assert np.array_equal(rows[i], o_rows), "IDs must be bit-exact"
        np.testing.assert_allclose(scores[i], o_sims[:eff] if len(o_sims) else o_sims, atol=1e-5, rtol=0)
    if index is None:
        ix.close()
    return rows, scores


def test_reference_toy_cases():
    # tests/test_utils.py:181-208 of the reference
    emb = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], dtype=np.float32)
    q = np.array([0.0, 1.0], dtype=np.float32)
    rows, sims = cosine_similarity_search(q, emb, 2)
    assert len(rows) == 2 and rows[0] == 1 and sims[0] == 1.0 and rows.dtype == np.int64
    rows, sims = cosine_similarity_search(q, emb[:2], 10)        # top_k > N returns N
    assert len(rows) == 2 and len(sims) == 2
    rows, sims = cosine_similarity_search(q, emb, 0)             # top_k == 0 -> empty
    assert len(rows) == 0 and len(sims) == 0


def test_golden_10k_from_reference_utils(golden_dense):
    n, dim, seed_v, seed_q, k = golden_dense["recipe"].tolist()
    V = synth.unit_rows(n, dim, seed_v)
    Q = synth.unit_rows(golden_dense["rows"].shape[0], dim, seed_q)
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(Q, k)
    for i in range(len(Q)):
        ref64 = OD.sims_float64(V, Q[i])
        assert min_gap(ref64, k) > 4e-7            # the recipe has no near-ties: IDs must be exact
        assert np.array_equal(rows[i], golden_dense["rows"][i])
        np.testing.assert_allclose(scores[i], golden_dense["sims"][i], atol=1e-5, rtol=0)
    ix.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 150, 151, 1000, 4097, 20000])
@pytest.mark.parametrize("pool", [1, 10, 150])
def test_ragged_row_counts(n, pool):
    V = synth.unit_rows(n, 384, 100 + n)
    Q = synth.unit_rows(2, 384, 7)
    check_against_oracle(V, Q, pool)


@pytest.mark.parametrize("dim", [2, 3, 64, 100, 128, 384, 768, 1000])
def test_other_dimensions(dim):
    V = synth.unit_rows(3000, dim, dim)
    Q = synth.unit_rows(3, dim, dim + 1)
    check_against_oracle(V, Q, 100)


@pytest.mark.parametrize("batch", [1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65, 130])
def test_batches_and_batch_invariance(batch):
    V = synth.unit_rows(100_000, 384, 11)        # (>= 8 x pool tiles: below that every batch runs the VALU scans)
    Q = synth.unit_rows(batch, 384, 12)
    ix = ProductIndex(V)
    rows, scores = check_against_oracle(V, Q, 150, ix)
    # Batches of <= 4 run the per-row-chain VALU scan; larger ones the bf16 filter scan whose candidates
    # are rescored with that same per-row chain: a query's answer is bitwise the same alone and in
    # any batch (queries are taken 128 at a time).
    r1, s1 = ix.dense_topk(Q[-1:], 150)
    assert np.array_equal(r1[0], rows[-1]) and np.array_equal(s1[0].view(np.uint32), scores[-1].view(np.uint32))
    if batch > 4:
        assert ix.select_trace()[0] in (1, 2)
        Q2 = np.concatenate([synth.unit_rows(4 + (batch % 7), 384, 13), Q[-1:]])
        r2, s2 = ix.dense_topk(Q2, 150)
        assert ix.select_trace()[0] == 2, "the filter + rescoring path must be the one that ran"
        assert np.array_equal(r2[-1], rows[-1]) and np.array_equal(s2[-1].view(np.uint32), scores[-1].view(np.uint32))
    ix.close()


def test_matrix_core_scan_ragged_tail_and_ties():
    # 64-query batches over row counts that do not fill the last 64-row tile, with duplicates
    for n in (64, 65, 1000, 4097, 70001):
        V = synth.unit_rows(n, 384, 200 + n)
        if n > 500:
            V[300:400] = V[5]
        Q = synth.unit_rows(40, 384, 3)
        Q[7] = V[5]
        check_against_oracle(V, Q, min(150, n))


def _same_up_to_rounding(rows_a, sc_a, rows_b, sc_b, atol=5e-7):
    """Two exact-arithmetic variants of one search: scores agree to fp32 rounding; rows may swap only
    where neighbouring scores are that close."""
    np.testing.assert_allclose(sc_a, sc_b, atol=atol, rtol=0)     # (a 24-element fmaf chain x 16 partials vs split bf16 terms)
    diff = rows_a != rows_b
    if diff.any():
        gaps = np.abs(np.diff(sc_a.astype(np.float64), axis=-1))
        near = np.zeros_like(diff)
        near[..., :-1] |= gaps < 2 * atol
        near[..., 1:] |= gaps < 2 * atol
        assert not (diff & ~near).any(), "rows differ where the scores are well separated"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("batch", [5, 16, 17, 40, 64, 100, 128, 130])
def test_filter_scan_plus_rescoring_is_bitwise_the_single_query_scan(batch, dtype):
    # default batched path: bf16 filter scan -> candidate M-tiles -> per-row-chain rescoring (trace[0] == 2).
    # Its answer must be the single-query scan's, bit for bit; the stored-score pass of the split-operand
    # scan (diagnostic mode, also the per-query fallback) agrees to fp32 rounding.
    V = synth.unit_rows(300_000, 384, 77)
    V[1000:1040] = V[11]                         # ties across M-tiles
    Q = synth.unit_rows(batch, 384, 78)
    Q[3] = V[11]
    ix = ProductIndex(V) if dtype == "f32" else ProductIndex.from_rows(V, dtype="bf16")
    rows2, sc2 = ix.dense_topk(Q, 150)
    tail = batch % 128 or 128                    # queries are taken 128 at a time; a tail of <= 4 runs the VALU scan
    assert ix.select_trace()[0] == (2 if tail > 4 else 1), "the filter + rescoring path must be the one that ran"
    for i in (0, 3, batch - 1):
        r1, s1 = ix.dense_topk(Q[i:i + 1], 150)
        assert np.array_equal(r1[0], rows2[i]) and np.array_equal(s1[0].view(np.uint32), sc2[i].view(np.uint32))
    ix.set_scan_mode(stored=True)
    rows1, sc1 = ix.dense_topk(Q[:64], 150)
    assert ix.select_trace()[0] == 1
    ix.set_scan_mode(stored=False)
    _same_up_to_rounding(rows1, sc1, rows2[:64], sc2[:64])
    if dtype == "f32":
        check_against_oracle(V, Q[:6], 150, index=ix)
    ix.close()


def test_filter_path_falls_back_per_query_on_massive_ties():
    # query 0 ties on > 16384 M-tiles (its candidate list overflows -> flagged -> stored-score pass);
    # the other queries of the batch stay on the filter path.  Both kinds must be exact.
    V = synth.unit_rows(200_000, 384, 91)
    V[::10] = V[3]                               # 20000 copies of row 3, spread over 20000 M-tiles
    Q = synth.unit_rows(20, 384, 92)
    Q[0] = V[3]
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(Q, 150)
    assert rows[0].tolist() == sorted({3} | set(range(0, 10 * 150, 10)))[:150]   # equal scores -> ascending row
    assert np.all(scores[0] == scores[0][0])
    r1, s1 = ix.dense_topk(Q[5:6], 150)          # a query that stayed on the filter path
    assert np.array_equal(r1[0], rows[5]) and np.array_equal(s1[0], scores[5])
    r0, s0 = ix.dense_topk(Q[0:1], 150)          # the flagged one: served by the single-query chain inside the batch too
    assert np.array_equal(r0[0], rows[0]) and np.array_equal(s0[0].view(np.uint32), scores[0].view(np.uint32))
    ix.set_scan_mode(stored=True)
    rows1, sc1 = ix.dense_topk(Q, 150)
    ix.set_scan_mode(stored=False)
    _same_up_to_rounding(rows1, sc1, rows, scores)
    check_against_oracle(V, Q[:8], 150, index=ix)
    ix.close()


def test_filter_path_rows_and_queries_of_any_norm():
    # the filter's error bound scales with the largest row norm and the query norm: rows from 0.01 to 30
    # long (one bound for all of them: loose for the short rows, more candidates, same answer)
    V = synth.unit_rows(120_000, 384, 301)
    V *= np.geomspace(0.01, 30.0, len(V), dtype=np.float32)[np.random.default_rng(5).permutation(len(V))][:, None]
    Q = synth.unit_rows(24, 384, 302) * np.float32(7.5)
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(Q, 150)
    for i in range(len(Q)):
        ref64 = OD.sims_float64(V, Q[i])
        assert_topk_matches(rows[i], scores[i], ref64, 150, tie_eps=4e-7 * 225, score_tol=1e-5 * 225)
        r1, s1 = ix.dense_topk(Q[i:i + 1], 150)
        assert np.array_equal(r1[0], rows[i]) and np.array_equal(s1[0].view(np.uint32), scores[i].view(np.uint32))
    ix.close()


def test_filter_path_crowded_cut_still_exact():
    # 6000 rows within ~1e-5 of each other around the cut: bf16 scores cannot tell them apart, the
    # candidate lists overflow for that query and the exact pass serves it; the rest stay filtered
    rng = np.random.default_rng(11)
    V = synth.unit_rows(150_000, 384, 401)
    u = synth.unit_rows(1, 384, 402)[0]
    crowd = u[None, :] + 3e-3 * rng.standard_normal((6000, 384)).astype(np.float32)
    V[rng.choice(len(V), 6000, replace=False)] = crowd / np.linalg.norm(crowd, axis=1, keepdims=True)
    Q = synth.unit_rows(12, 384, 403)
    Q[0] = u
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(Q, 150)
    for i in range(len(Q)):
        # (fp32 itself reorders this crowd against float64: a wider tie band than elsewhere)
        assert_topk_matches(rows[i], scores[i], OD.sims_float64(V, Q[i]), 150, tie_eps=2e-6)
        r1, s1 = ix.dense_topk(Q[i:i + 1], 150)          # bit for bit: the flagged query goes through the single-query chain
        assert np.array_equal(r1[0], rows[i]) and np.array_equal(s1[0].view(np.uint32), scores[i].view(np.uint32))
    ix.close()


def test_eight_flagged_queries_go_through_the_chain_more_go_to_the_split_operand_pass():
    # queries that tie on 20000 rows overflow their candidate lists.  Eight of them in a call: served by the single-query
    # chain, bit for bit a batch of one.  Twelve: all by the stored-score pass of the split-operand scan (equal to fp32
    # rounding) -- a pass of the chain for eight of them would only add to the passes the others need.  Exact either way.
    V = synth.unit_rows(200_000, 384, 91)
    V[::10] = V[3]
    ix = ProductIndex(V)
    want = sorted({3} | set(range(0, 10 * 150, 10)))[:150]
    for flagged in ([1, 4, 5, 9, 12, 13, 20, 21], [1, 4, 5, 9, 12, 13, 20, 21, 30, 31, 38, 39]):
        Q = synth.unit_rows(40, 384, 93)
        Q[flagged] = V[3]
        rows, scores = ix.dense_topk(Q, 150)
        r1, s1 = ix.dense_topk(Q[1:2], 150)
        for q in flagged:
            assert rows[q].tolist() == want
            if len(flagged) <= 8:
                assert np.array_equal(scores[q].view(np.uint32), s1[0].view(np.uint32))
            else:
                assert np.allclose(scores[q], s1[0], rtol=0, atol=1e-6)   # (scores near 1: a few fp32 ulps between the two arithmetics)
        for q in (0, 2, 37):                         # neighbours that stayed on the filter path
            rq, sq = ix.dense_topk(Q[q:q + 1], 150)
            assert np.array_equal(rq[0], rows[q]) and np.array_equal(sq[0].view(np.uint32), scores[q].view(np.uint32))
    ix.close()


def test_batched_search_with_a_nan_row_takes_the_exact_scans():
    V = synth.unit_rows(20_000, 384, 41)
    V[5, 3] = np.nan                             # no finite row-norm bound: no filtering
    Q = synth.unit_rows(20, 384, 42)
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(Q, 150)
    assert 5 not in rows.ravel().tolist() and np.isfinite(scores).all()
    Vc = V.copy(); Vc[5] = 0
    for i in (0, 19):
        ref64 = OD.sims_float64(Vc, Q[i]); ref64[5] = -np.inf
        assert_topk_matches(rows[i], scores[i], ref64, 150)
    ix.close()


@pytest.mark.parametrize("pool", [255, 256, 257, 1000, 2048])
def test_large_pools_cross_the_fast_path_limits(pool):
    # the 3-level selection opens at most 256 groups per level in LDS; larger pools fall back
    V = synth.unit_rows(200_000, 384, 51)
    Q = synth.unit_rows(2, 384, 52)
    check_against_oracle(V, Q, pool)


def test_exact_ties_resolve_to_the_smaller_row():
    V = synth.unit_rows(5000, 384, 21)
    V[100:4000:100] = V[7]                      # 39 exact copies of row 7
    q = V[7].copy()
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(q[None, :], 50)
    dup = [7] + list(range(100, 4000, 100))
    assert rows[0][:40].tolist() == sorted(dup)           # equal scores -> ascending row
    assert np.all(scores[0][:40] == scores[0][0])
    ix.close()


def test_massive_ties_all_rows_equal():
    V = np.tile(synth.unit_rows(1, 384, 5), (300000, 1))   # every tile maximum ties (> 4096 tiles)
    q = synth.unit_rows(1, 384, 6)
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(q, 150)
    assert rows[0].tolist() == list(range(150)) and np.all(scores[0] == scores[0][0])
    z = np.zeros((9000, 384), dtype=np.float32)
    ixz = ProductIndex(z)
    rows, scores = ixz.dense_topk(q, 2048)
    assert rows[0].tolist() == list(range(2048)) and np.all(scores[0] == 0)
    ix.close(); ixz.close()


def test_clustered_rows_take_the_radix_fallback():
    # rows sorted by score: the best pool tiles hold > 8192 qualifying rows
    V = synth.unit_rows(60000, 384, 31)
    q = synth.unit_rows(1, 384, 32)
    order = np.argsort(-(V.astype(np.float64) @ q[0].astype(np.float64)))
    Vs = np.ascontiguousarray(V[order])
    rows, _ = check_against_oracle(Vs, q, 150)
    assert rows[0].tolist() == list(range(150))


def test_nan_rows_rank_last_and_results_stay_deterministic():
    V = synth.unit_rows(2000, 384, 41)
    V[5, 3] = np.nan
    q = synth.unit_rows(1, 384, 42)
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(q, 1999)
    assert 5 not in rows[0].tolist()
    ix.close()


def test_argument_errors_are_raised_not_swallowed():
    V = synth.unit_rows(100, 384, 1)
    ix = ProductIndex(V)
    with pytest.raises(ValueError):
        ix.dense_topk(np.zeros((1, 383), dtype=np.float32), 10)
    with pytest.raises(ValueError):
        ix.dense_topk(np.zeros((1, 384), dtype=np.float32), -1)
    ix.close()


def test_l2_normalize_on_device_matches_numpy_to_rounding():
    from oracle.primitives import l2_normalize
    rng = np.random.default_rng(3)
    X = rng.standard_normal((1000, 384)).astype(np.float32) * 3
    X[10] = 0
    ix = ProductIndex(X, normalize=True)
    q = synth.unit_rows(1, 384, 9)
    rows, scores = ix.dense_topk(q, 100)
    ref = OD.sims_float64(l2_normalize(X), q[0])
    assert_topk_matches(rows[0], scores[0], ref, 100, tie_eps=1e-6)
    ix.close()


def test_one_million_rows_full_size():
    """BASELINE config 2 size (1M x 384 fp32): checked against a chunked float64 oracle."""
    n = 1_000_000
    V = synth.unit_rows(n, 384, 1234)
    Q = synth.unit_rows(2, 384, 77)
    ix = ProductIndex(V)
    rows, scores = ix.dense_topk(Q, 150)
    for i in range(2):
        ref = np.concatenate([V[s:s + 100000].astype(np.float64) @ Q[i].astype(np.float64)
                              for s in range(0, n, 100000)])
        assert_topk_matches(rows[i], scores[i], ref, 150)
        o_rows, o_sims = OD.cosine_similarity_search(Q[i], V, 150)
        if min_gap(ref, 150) > 4e-7:
            assert np.array_equal(rows[i], o_rows)
    assert ix.last_scan_ms() > 0
    ix.close()


def test_ten_million_rows_batched_at_full_size():
    """BASELINE's headline size (10M x 384 fp32, 15.4 GB), generated on the device in seeded blocks, searched with the
    headline batch: 256 queries = ONE launch of the query-stationary filter scan rr_scan_fltq (what bench.py times),
    checked against (a) the single-query scan, bitwise, for queries of both sets, (b) a chunked float64 oracle on the
    host for queries of both sets, (c) the size-independent properties of every answer; then the 128-query launch
    (rr_scan_flt16) the same way."""
    import torch
    n, pool = 10_000_000, 150
    mat = torch.empty((n, 384), device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda")
    for b, s in enumerate(range(0, n, 1_250_000)):
        g.manual_seed(9000 + b)
        blk = torch.randn((1_250_000, 384), generator=g, device="cuda")
        mat[s:s + 1_250_000] = blk / blk.norm(dim=1, keepdim=True)
    del blk
    ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
    Q = synth.unit_rows(256, 384, 4242)
    Q[5] = mat[7_654_321].cpu().numpy()                      # queries that are rows: score 1 at that row
    Q[200] = mat[9_999_999].cpu().numpy()                    # (the matrix's last row, second query set)

    def f64_scores(q):                                       # float64 oracle, 1.25M rows at a time
        q64 = torch.from_numpy(q.astype(np.float64)).cuda()
        return np.concatenate([(mat[s:s + 1_250_000].double() @ q64).cpu().numpy() for s in range(0, n, 1_250_000)])

    rows, scores = ix.dense_topk(Q, pool)
    info = ix.last_scan_info()
    assert info[0] == 5 and info[1] == 9 and info[2] == 256 and info[4] == 2, info      # rr_scan_fltq over the bf16 plane
    assert ix.select_trace()[0] == 2
    assert rows[5][0] == 7_654_321 and abs(scores[5][0] - 1.0) < 1e-6
    assert rows[200][0] == 9_999_999 and abs(scores[200][0] - 1.0) < 1e-6
    for i in range(len(Q)):                                  # properties: sorted, distinct, in range
        assert np.all(np.diff(scores[i]) <= 0) and len(set(rows[i].tolist())) == pool
        assert rows[i].min() >= 0 and rows[i].max() < n
    for i in (0, 5, 127, 128, 200, 255):                     # bitwise the single-query scan
        r1, s1 = ix.dense_topk(Q[i:i + 1], pool)
        assert np.array_equal(r1[0], rows[i]) and np.array_equal(s1[0].view(np.uint32), scores[i].view(np.uint32)), i
    for i in (0, 127, 128, 255):
        assert_topk_matches(rows[i], scores[i], f64_scores(Q[i]), pool)
    # the 128-query launch (rr_scan_flt16, store prefilter on at this size): the same answers, bit for bit
    rows128, scores128 = ix.dense_topk(Q[:128], pool)
    info = ix.last_scan_info()
    assert info[0] == 5 and info[1] == 4 and info[4] == 2, info
    assert np.array_equal(rows128, rows[:128]) and np.array_equal(scores128.view(np.uint32), scores[:128].view(np.uint32))
    ix.close()


def test_adopted_matrix_changed_behind_the_index():
    """An adopted device matrix (zero-copy) that the caller rewrites in place: rr_index_matrix_changed drops the cached
    row-norm bounds and the bf16 filter plane, so the next batched search filters on the NEW rows."""
    import torch
    n, pool = 300_000, 150
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    mat = torch.randn((n, 384), device="cuda", generator=g); mat /= mat.norm(dim=1, keepdim=True)
    ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
    Q = synth.unit_rows(32, 384, 5)
    ix.dense_topk(Q, pool)                                   # builds the plane and the bounds of the old content
    g.manual_seed(4)
    new = torch.randn((n, 384), device="cuda", generator=g)
    mat.copy_(3.0 * new / new.norm(dim=1, keepdim=True))     # in place, behind the index; rows of norm 3 (the old bound: 1)
    torch.cuda.synchronize()
    ix.matrix_changed()
    rows, scores = ix.dense_topk(Q, pool)
    fresh = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
    want_rows, want_scores = fresh.dense_topk(Q, pool)
    assert np.array_equal(rows, want_rows) and np.array_equal(scores.view(np.uint32), want_scores.view(np.uint32))
    r1, s1 = ix.dense_topk(Q[:1], pool)                      # and both equal the single-query scan of the new rows
    assert np.array_equal(r1[0], rows[0])
    ix.close(); fresh.close()


def test_f32_chain_matrix_core_path_in_a_subprocess():
    """RR_SCAN_F32_CHAIN=1 selects the f32-input MFMA kernels (scores = pure fmaf chains) for
    batches; the switch is read once per process, so the check runs in a child process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from oracle import dense as OD
from parity import assert_topk_matches
from review_recommender_amd import synth
from review_recommender_amd.index import ProductIndex
for dtype in ("f32", "bf16"):
    V = synth.unit_rows(20001, 384, 71)
    Vo = OD.round_to_bf16(V) if dtype == "bf16" else V
    Q = synth.unit_rows(40, 384, 72)
    ix = ProductIndex.from_rows(V, dtype=dtype)
    rows, scores = ix.dense_topk(Q, 150)
    for i in range(len(Q)):
        assert_topk_matches(rows[i], scores[i], OD.sims_float64(Vo, Q[i]), 150)
    r2, s2 = ix.dense_topk(np.concatenate([Q[5:12], Q[:1]]), 150)      # batch-invariant, bitwise
    assert np.array_equal(r2[-1], rows[0]) and np.array_equal(s2[-1], scores[0])
print("chain-ok")
'''
    env = dict(os.environ, RR_SCAN_F32_CHAIN="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "chain-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_exact_split_operand_scans_in_a_subprocess():
    """RR_SCAN_EXACT=1 turns the filter scan off: batches run the split-operand scans (16x16x32 tiles up
    to 16 queries, 32x32x16 beyond) with their own two-pass selection and bit-exact rescoring -- the
    path a matrix without a finite row-norm bound takes.  Read once per process: child process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from oracle import dense as OD
from parity import assert_topk_matches
from review_recommender_amd import synth
from review_recommender_amd.index import ProductIndex
for dtype in ("f32", "bf16"):
    V = synth.unit_rows(150_037, 384, 171)
    V[149_990:150_030] = V[7]                     # ties in the ragged tail
    Vo = OD.round_to_bf16(V) if dtype == "bf16" else V
    ix = ProductIndex.from_rows(V, dtype=dtype)
    for batch in (9, 40, 64):
        Q = synth.unit_rows(batch, 384, 172 + batch)
        Q[2] = V[7]
        rows, scores = ix.dense_topk(Q, 150)
        assert ix.select_trace()[0] == 2, ix.select_trace()[:4]       # two-pass, no stored scores
        for i in (0, 2, batch - 1):
            assert_topk_matches(rows[i], scores[i], OD.sims_float64(Vo, Q[i]), 150)
        ix.set_scan_mode(True)
        r1, s1 = ix.dense_topk(Q, 150)                                # the stored-score pass: bit-equal
        ix.set_scan_mode(False)
        assert np.array_equal(r1, rows) and np.array_equal(s1.view(np.uint32), scores.view(np.uint32))
print("exact-ok")
'''
    env = dict(os.environ, RR_SCAN_EXACT="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "exact-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("batch", [40, 130])
def test_filter_scan_ragged_tail_and_ties(batch):
    # just above the 8 x pool tiles the filter path asks for, last tile 37 rows, ties that straddle it
    n = 8 * 150 * 64 + 37
    V = synth.unit_rows(n, 384, 500)
    V[n - 50:n - 5] = V[5]
    Q = synth.unit_rows(batch, 384, 501)
    Q[7] = V[5]
    ix = ProductIndex(V)
    rows, scores = check_against_oracle(V, Q[:9], 150, index=ix)
    rows_all, scores_all = ix.dense_topk(Q, 150)
    assert np.array_equal(rows_all[:9], rows) and np.array_equal(scores_all[:9].view(np.uint32), scores.view(np.uint32))
    dup = [5] + list(range(n - 50, n - 5))
    assert rows_all[7][:46].tolist() == sorted(dup)
    ix.close()


def _scan_info(ix):
    import ctypes as C
    from review_recommender_amd import _lib
    out = (C.c_int32 * 8)()
    _lib.check(_lib.load().rr_index_last_scan_info(ix.handle, out), "rr_index_last_scan_info")
    return tuple(out)


def test_bf16_filter_plane_of_an_fp32_index_gives_bitwise_the_same_answers(hip):
    """An fp32 index scans a once-rounded bf16 copy of its rows in the batched filter path (half the bytes per
    launch) and rescores candidates on the fp32 rows: the answers must equal, bit for bit, those of the scan
    over the fp32 rows themselves (rr_index_set_shadow(0)) and those of the single-query scan; a write to the
    matrix drops the plane."""
    from review_recommender_amd import _lib
    n = 200_000
    V = synth.unit_rows(n, 384, 91)
    V[500:530] = V[7]
    Q = synth.unit_rows(120, 384, 92)            # one filter-scan launch (a tail of <= 4 queries would run the VALU scan last)
    Q[4] = V[7]
    ix = ProductIndex(V)
    rows_s, sc_s = ix.dense_topk(Q, 150)
    info = _scan_info(ix)
    assert info[0] == 5 and info[4] == 2, "the batched scan must have streamed the bf16 plane"
    _lib.check(hip.rr_index_set_shadow(ix.handle, 0), "set_shadow")
    rows_f, sc_f = ix.dense_topk(Q, 150)
    assert _scan_info(ix)[4] == 4
    assert np.array_equal(rows_s, rows_f) and np.array_equal(sc_s.view(np.uint32), sc_f.view(np.uint32))
    _lib.check(hip.rr_index_set_shadow(ix.handle, 1), "set_shadow")
    for i in (0, 4, 77, 119):
        r1, s1 = ix.dense_topk(Q[i:i + 1], 150)
        assert np.array_equal(r1[0], rows_s[i]) and np.array_equal(s1[0].view(np.uint32), sc_s[i].view(np.uint32))
    check_against_oracle(V, Q[:5], 150, index=ix)
    # new rows over the old ones: the plane (and the row-norm bound) must follow
    V2 = synth.unit_rows(n, 384, 93)
    _lib.check(hip.rr_index_upload_rows(ix.handle, 0, n, _lib.ptr(V2)), "upload")
    rows_b, sc_b = ix.dense_topk(Q[:40], 150)
    assert _scan_info(ix)[4] == 2
    for i in (0, 13, 39):
        r1, s1 = ix.dense_topk(Q[i:i + 1], 150)
        assert np.array_equal(r1[0], rows_b[i]) and np.array_equal(s1[0].view(np.uint32), sc_b[i].view(np.uint32))
    check_against_oracle(V2, Q[:3], 150, index=ix)
    ix.close()


def test_store_prefilter_of_the_filter_scan_stays_exact():
    """>= 2M rows: the filter scan only writes back the tile words whose 32 queries can still matter (thresholds from a
    1/64 tile sample).  (a) ordinary data: batched == single-query scan, bitwise, and the prefilter did not push any
    query to the fallback; (b) adversarial layout -- the best rows of one query sit exactly in the sampled tiles, so
    its sampled threshold overshoots: that query must take the exact fallback and still be right; (c) RR_NO_PREFILTER
    semantics are the same answers (checked through (a) against the single-query scan, which has no prefilter)."""
    import torch
    n, pool = 2_200_000, 150
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    mat = torch.randn((n, 384), generator=g, device="cuda")
    mat /= mat.norm(dim=1, keepdim=True)
    Q = synth.unit_rows(40, 384, 4711)
    # (b) 400 rows close to Q[3] placed in sampled tiles (tile 64 i + 32 -> rows (64 i + 32) * 32 + r)
    q3 = torch.from_numpy(Q[3]).cuda()
    gi = torch.Generator(device="cuda")
    gi.manual_seed(5)
    for i in range(400):
        row = (64 * (i + 3) + 32) * 32 + (i % 32)
        v = q3 * (0.9 - 0.0005 * i) + 0.3 * torch.randn(384, generator=gi, device="cuda") / 384 ** 0.5
        mat[row] = v / v.norm()
    ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
    rows, scores = ix.dense_topk(Q, pool)
    assert _scan_info(ix)[0] == 5 and _scan_info(ix)[4] == 2
    for i in (0, 1, 2, 17, 39):
        r1, s1 = ix.dense_topk(Q[i:i + 1], pool)
        assert np.array_equal(r1[0], rows[i]) and np.array_equal(s1[0].view(np.uint32), scores[i].view(np.uint32))
    # query 3: served exactly whichever path it took
    ref = (mat.double() @ q3.double()).cpu().numpy()
    assert_topk_matches(rows[3], scores[3], ref, pool)
    r1, s1 = ix.dense_topk(Q[3:4], pool)
    assert_topk_matches(r1[0], s1[0], ref, pool)
    np.testing.assert_allclose(s1[0], scores[3], atol=2e-6, rtol=0)      # scores ~0.95: chain vs split-operand rounding
    ix.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("batch", [192, 193, 200, 256, 300, 449])
def test_paired_scan_launches_share_one_selection(batch, dtype):
    """193 .. 256 queries go as two filter-scan launches followed by ONE selection / rescoring sequence (the second
    set's maxima, bounds and thresholds live behind the first's).  Every answer must still be the single-query scan's,
    bit for bit -- including queries of the second set, ties across M-tiles, and batches that mix a pair with a tail."""
    V = synth.unit_rows(400_000, 384, 177)
    V[2000:2040] = V[15]
    Q = synth.unit_rows(batch, 384, 178)
    Q[3] = V[15]
    Q[batch - 2] = V[15]                          # a tie-heavy query in the last launch too
    ix = ProductIndex(V) if dtype == "f32" else ProductIndex.from_rows(V, dtype="bf16")
    rows, scores = ix.dense_topk(Q, 150)
    for i in sorted({0, 3, 127, 128, 129, 191, 192, batch - 2, batch - 1} & set(range(batch))):
        r1, s1 = ix.dense_topk(Q[i:i + 1], 150)
        assert np.array_equal(r1[0], rows[i]), i
        assert np.array_equal(s1[0].view(np.uint32), scores[i].view(np.uint32)), i
    if dtype == "f32":
        check_against_oracle(V, Q[[0, 130, batch - 1]], 150, index=ix)
    ix.close()
