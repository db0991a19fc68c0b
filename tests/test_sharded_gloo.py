"""The N > 1 host logic on CPU: two processes, gloo backend.  Covers shard bounds, the packed
payload layout, the single all-gather, and the [rank][query][pool] addressing the fusion kernel
reads the gathered blocks with.  Per-shard candidates come from the oracle here (no GPU); the
merged pool is checked against the oracle's global top-pool."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dense as OD
from review_recommender_amd import synth
from review_recommender_amd.sharded import PayloadLayout, exchange, shard_bounds

N, DIMS, POOL, B = 1003, 64, 40, 3


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_rows_exactly_once():
    for n in (1, 7, 8, 1003, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_payload_layout_is_aligned_and_disjoint():
    lay = PayloadLayout(3, 150)
    offs = [lay.off_rows, lay.off_n, lay.off_avg, lay.off_l1p, lay.off_dense, lay.off_bm25, lay.nbytes]
    assert all(o % 16 == 0 for o in offs) and offs == sorted(offs)
    assert lay.nbytes >= 3 * 150 * 40
    buf = torch.zeros(lay.nbytes, dtype=torch.uint8)
    v = lay.views(buf)
    v["rows"][:] = 7
    v["bm25"][:] = 1.5
    assert int(v["n"].abs().sum()) == 0 and float(v["dense"].abs().sum()) == 0
    assert v["rows"].shape == (3, 150) and v["rows"].dtype == torch.int64


def _worker(rank, world, port, V, Q, n_rev, stars, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(len(V), world, rank)
        lay = PayloadLayout(len(Q), POOL)
        buf = torch.zeros(lay.nbytes, dtype=torch.uint8)
        v = lay.views(buf)
        for b, q in enumerate(Q):
            sims = V[lo:hi] @ q
            rows, sc = OD.topk_reference_order(sims, POOL)
            v["rows"][b] = torch.from_numpy(rows + lo)
            v["dense"][b] = torch.from_numpy(sc)
            v["bm25"][b] = torch.from_numpy((rows + lo).astype(np.float32) * 0.5)
            v["n"][b] = torch.from_numpy(n_rev[rows + lo].astype(np.float64))
            v["avg"][b] = torch.from_numpy(stars[rows + lo])
            v["l1p"][b] = torch.from_numpy(np.log1p(n_rev[rows + lo].astype(np.float64)))
        g = exchange(buf, world)
        assert g.shape == (world, lay.nbytes)
        ret[rank] = g.numpy().copy()
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange_and_merge_addressing():
    world = 2
    V = synth.unit_rows(N, DIMS, 3)
    Q = synth.unit_rows(B, DIMS, 4)
    n_rev, stars = synth.metadata(N, 5)
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, V, Q, n_rev, stars, ret), nprocs=world, join=True)
    g0, g1 = ret[0], ret[1]
    assert np.array_equal(g0, g1), "every rank must end with the same gathered blocks"
    lay = PayloadLayout(B, POOL)
    stride = lay.nbytes

    def at(col_off, dtype, q, i):          # the kernel's rr_cand_addr::get, restated
        r, j = divmod(i, POOL)
        a = g0.reshape(-1)[r * stride + col_off:].view(dtype)
        return a[q * POOL + j]

    for q in range(B):
        cand = [(float(at(lay.off_dense, np.float32, q, i)), int(at(lay.off_rows, np.int64, q, i)), i)
                for i in range(world * POOL)]
        merged = sorted(cand, key=lambda t: (-t[0], t[1]))[:POOL]
        sims = V @ Q[q]
        want_rows, want_sc = OD.topk_reference_order(sims, POOL)
        # per-shard BLAS results can differ in the last bit from the full matvec: compare rows
        # wherever the scores are not within rounding of a neighbour
        got_rows = np.array([t[1] for t in merged])
        diff = got_rows != want_rows
        assert np.all(np.abs(sims[got_rows[diff]] - sims[want_rows[diff]]) < 1e-6)
        for sc, row, i in merged[:5]:
            assert float(at(lay.off_bm25, np.float32, q, i)) == row * 0.5
            assert float(at(lay.off_n, np.float64, q, i)) == float(n_rev[row])
            assert float(at(lay.off_avg, np.float64, q, i)) == float(stars[row])


def _rerank_worker(rank, world, port, n_pairs, ret):
    from review_recommender_amd.sharded import exchange_scores, split_pairs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # first collective: the payload all-gather (as above); second: the scores of this rank's share of the pairs
        buf = torch.full((64,), rank, dtype=torch.uint8)
        g = exchange(buf, world)
        lo, hi = split_pairs(n_pairs, world, rank)
        mine = torch.arange(lo, hi, dtype=torch.float32) * 0.25 - 3.0      # score(pair i) = i / 4 - 3
        full = exchange_scores(mine, n_pairs, world)
        ret[rank] = (g.numpy().copy(), full.numpy().copy(), (lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [64 * 200, 7, 1])
def test_rerank_flow_two_collectives_config5(n_pairs):
    """BASELINE config 5's second exchange: after the merge the B x rr_k reranker pairs are split
    evenly across the ranks, every rank scores its share, one tiny all-gather gives every rank all scores."""
    from review_recommender_amd.sharded import split_pairs
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rerank_worker, args=(world, port, n_pairs, ret), nprocs=world, join=True)
    want = np.arange(n_pairs, dtype=np.float32) * 0.25 - 3.0
    spans = [ret[r][2] for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n_pairs and spans[0][1] == spans[1][0]
    assert abs((spans[0][1] - spans[0][0]) - (spans[1][1] - spans[1][0])) <= 1
    for r in range(world):
        assert np.array_equal(ret[r][1], want)                       # every pair exactly once, in order, on every rank
        assert np.array_equal(ret[r][0][0], np.zeros(64, np.uint8)) and np.array_equal(ret[r][0][1], np.ones(64, np.uint8))
    for w_ in (1, 3, 8):
        cover = [split_pairs(n_pairs, w_, r) for r in range(w_)]
        assert cover[0][0] == 0 and cover[-1][1] == n_pairs and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def _floor_worker(rank, world, port, ret):
    from review_recommender_amd.sharded import exchange_floor
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # per-query bounds of this "shard": rank 1 could not split its call for query 2 (-inf = no floor for it)
        mine = torch.tensor([0.30 + 0.01 * rank, 0.25 - 0.02 * rank, float("-inf") if rank == 1 else 0.4], dtype=torch.float32)
        ret[rank] = exchange_floor(mine, world).numpy().copy()
    finally:
        dist.destroy_process_group()


def test_floor_exchange_is_the_elementwise_minimum():
    """The collective in front of the shards' selection (two-phase K1): every rank ends with the
    per-query minimum of the shards' bounds; one -inf makes the floor -inf (that query is selected as if unsharded)."""
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_floor_worker, args=(world, port, ret), nprocs=world, join=True)
    want = np.array([0.30, 0.23, -np.inf], dtype=np.float32)
    for r in range(world):
        assert np.array_equal(ret[r], want)


def _pipeline_worker(rank, world, port, ret):
    from review_recommender_amd.sharded import exchange_start
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # three batches in flight the way ShardedSearcher.submit / finish pipelines them: batch i + 1's exchange is started
        # (and "K1 of batch i + 1" -- here: filling the next buffer -- runs) before batch i's gathered blocks are waited for
        bufs = [torch.full((48,), 10 * i + rank, dtype=torch.uint8) for i in range(3)]
        got = []
        pend = exchange_start(bufs[0], world)
        for i in range(1, 3):
            nxt = exchange_start(bufs[i], world)
            got.append(pend.wait().numpy().copy())
            pend = nxt
        got.append(pend.wait().numpy().copy())
        assert pend.wait() is pend.out                      # a second wait is a no-op
        ret[rank] = got
    finally:
        dist.destroy_process_group()


def test_started_exchanges_complete_in_order_and_match_the_blocking_form():
    """"pipeline batches so the all-gather of batch i overlaps K1 of batch i+1" -- exchange_start returns at
    once, wait() hands back the gathered blocks; several may be in flight; every rank sees every batch's blocks."""
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_pipeline_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        for i in range(3):
            g = ret[r][i]
            assert g.shape == (world, 48)
            assert np.all(g[0] == 10 * i) and np.all(g[1] == 10 * i + 1)


def _overlap_flow_worker(rank, world, port, ret):
    from review_recommender_amd.sharded import exchange_floor_start, exchange_start
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the collectives of ShardedSearcher.enable_overlap's three stages, in the order its submit / finish issue them:
        #   submit(i):  floor all-reduce of batch i started (behind its scan), then payload all-gather of batch i - 1 started
        #               into that slot's OWN gathered buffer (ring of three);
        #   finish(i - 2): waits for batch i - 2's all-gather.
        n = 5
        rings = [torch.zeros((world, 32), dtype=torch.uint8) for _ in range(3)]
        bounds = [torch.tensor([0.5 + 0.1 * i - 0.05 * rank, float("-inf") if (rank == 1 and i == 2) else 0.1 * i],
                               dtype=torch.float32) for i in range(n)]
        floors, pend, got = {}, {}, {}
        for i in range(n + 2):
            if i < n:
                work = exchange_floor_start(bounds[i], world)          # gloo: done in place, nothing to wait for
                assert work is None
                floors[i] = bounds[i].numpy().copy()
            if 0 <= i - 1 < n:
                buf = torch.full((32,), 16 * (i - 1) + rank, dtype=torch.uint8)
                pend[i - 1] = exchange_start(buf, world, out=rings[(i - 1) % 3])
                assert pend[i - 1].out is rings[(i - 1) % 3]
            if 0 <= i - 2 < n:
                got[i - 2] = pend.pop(i - 2).wait().numpy().copy()   # (copied before the slot's buffer is gathered into again)
        ret[rank] = (floors, got)
    finally:
        dist.destroy_process_group()


def test_overlapped_pipeline_collectives_floor_started_then_gather_into_ring_buffers():
    """The N > 1 side of ShardedSearcher.enable_overlap on CPU (its kernels need a GPU: tests/test_gpu_sharded.py): the
    floor's all-reduce in its started form, the payload all-gather into caller-owned ring buffers three slots deep, two
    batches in flight behind the one being submitted -- every rank ends with every batch's minimum and blocks."""
    world = 2
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_overlap_flow_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        floors, got = ret[r]
        for i in range(5):
            want = np.array([0.5 + 0.1 * i - 0.05, -np.inf if i == 2 else 0.1 * i], dtype=np.float32)
            assert np.array_equal(floors[i], want), (r, i)
            assert np.all(got[i][0] == 16 * i) and np.all(got[i][1] == 16 * i + 1), (r, i)
    assert exchange_floor_start_single_rank()


def exchange_floor_start_single_rank():
    from review_recommender_amd.sharded import exchange_floor_start
    b = torch.tensor([0.25, 0.5])
    return exchange_floor_start(b, 1) is None and b.tolist() == [0.25, 0.5]
