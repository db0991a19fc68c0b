"""Row-sharded search over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  Rank r owns a
contiguous row range of the matrix, of n_reviews / avg_stars, and of the BM25 documents;
idf / avgdl are corpus-wide and replicated.  Per query batch every rank runs K1 and K2 on
its shard, packs the candidate payload

    rows int64 | n_reviews f64 | avg_stars f64 | log1p_n f64 | dense f32 | bm25 f32     (40 B/candidate)

into ONE contiguous device buffer, and a single all-gather moves all ranks' buffers to all
ranks.  K3 then reads the gathered blocks in place ([rank][query][pool] addressing),
merges world x pool candidates to the global top-pool by (dense desc, row asc) and fuses.
Every rank ends with the same result.  The global top-pool is contained in the union of
the local top-pools, and a row's score does not depend on the shard it sits in
(csrc/rr_dense.hip), so the answer equals the single-GPU answer bit for bit.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .engine import FusionWeights, HybridSearcher


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank``: the first n_rows % world ranks get one extra row."""
    if not (0 <= rank < world):
        raise ValueError("rank outside [0, world)")
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass(frozen=True)
class PayloadLayout:
    """Byte offsets of the candidate payload one rank contributes (all 16-B aligned)."""
    n_queries: int
    pool: int

    @property
    def count(self) -> int:
        return self.n_queries * self.pool

    def _al(self, x: int) -> int:
        return (x + 15) // 16 * 16

    @property
    def off_rows(self) -> int: return 0
    @property
    def off_n(self) -> int: return self._al(self.count * 8)
    @property
    def off_avg(self) -> int: return self.off_n + self._al(self.count * 8)
    @property
    def off_l1p(self) -> int: return self.off_avg + self._al(self.count * 8)
    @property
    def off_dense(self) -> int: return self.off_l1p + self._al(self.count * 8)
    @property
    def off_bm25(self) -> int: return self.off_dense + self._al(self.count * 4)
    @property
    def nbytes(self) -> int: return self.off_bm25 + self._al(self.count * 4)

    def views(self, buf):
        """Typed (n_queries, pool) views into one rank's uint8 buffer (torch tensor)."""
        import torch
        c, shape = self.count, (self.n_queries, self.pool)
        v = lambda off, n, dt: buf[off:off + n].view(dt).view(shape)
        return {"rows": v(self.off_rows, c * 8, torch.int64), "n": v(self.off_n, c * 8, torch.float64),
                "avg": v(self.off_avg, c * 8, torch.float64), "l1p": v(self.off_l1p, c * 8, torch.float64),
                "dense": v(self.off_dense, c * 4, torch.float32), "bm25": v(self.off_bm25, c * 4, torch.float32)}


class PendingExchange:
    """An all-gather that has been started (`exchange_start`): `wait()` returns the gathered (world, nbytes) tensor once
    the CURRENT stream may read it (RCCL: the collective runs on the process group's own stream; wait() makes the
    current stream wait for it without blocking the host -- kernels enqueued in between run under the collective)."""

    def __init__(self, out, work=None, finish=None):
        self.out, self.work, self._finish = out, work, finish

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self._finish is not None:
            self._finish()
            self._finish = None
        return self.out


def exchange_start(buf, world: int, group=None, out=None) -> PendingExchange:
    """Starts the path's one collective (all-gather of every rank's payload buffer) without waiting for it: what lets
    batch i's exchange run under K1 of batch i + 1 (ShardedSearcher.submit / finish).  ``out``: a caller-owned
    (world, nbytes) uint8 buffer to gather into."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)
    if world == 1:
        out[0].copy_(buf)
        return PendingExchange(out)
    if dist.get_backend(group) == "nccl":
        return PendingExchange(out, dist.all_gather_into_tensor(out, buf, group=group, async_op=True))   # RCCL over xGMI
    if buf.is_cuda:                                               # gloo rehearsal on a GPU box: gloo gathers host tensors
        host = torch.empty((world, buf.numel()), dtype=torch.uint8)
        work = dist.all_gather([host[r] for r in range(world)], buf.cpu(), group=group, async_op=True)
        return PendingExchange(out, work, lambda: out.copy_(host))
    return PendingExchange(out, dist.all_gather([out[r] for r in range(world)], buf, group=group, async_op=True))


def exchange(buf, world: int, group=None):
    """The path's one collective: all-gather of every rank's payload buffer.
    Returns a (world, nbytes) uint8 tensor, identical on every rank."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)
    if world == 1:
        out[0].copy_(buf)
        return out
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, buf, group=group)       # RCCL over xGMI
    elif buf.is_cuda:                                             # gloo rehearsal on a GPU box: gloo gathers host tensors
        host = torch.empty((world, buf.numel()), dtype=torch.uint8)
        dist.all_gather([host[r] for r in range(world)], buf.cpu(), group=group)
        out.copy_(host)
    else:                                                         # gloo (CPU tests)
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, buf, group=group)
    return out


def exchange_floor(bound, world: int, group=None):
    """The tiny collective in front of the shards' selection: element-wise MINIMUM over the ranks
    of the per-query bounds of HybridSearcher.dense_scan -- a lower bound of the corpus-wide pool-th best score.
    In place; B floats."""
    import torch.distributed as dist
    if world == 1:
        return bound
    if dist.get_backend(group) == "nccl" or not bound.is_cuda:
        dist.all_reduce(bound, op=dist.ReduceOp.MIN, group=group)
    else:                                                         # gloo rehearsal on a GPU box
        host = bound.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
        bound.copy_(host)
    return bound


def exchange_floor_start(bound, world: int, group=None):
    """exchange_floor without waiting for it (RCCL): returns the work handle whose wait() orders the CURRENT stream behind
    the all-reduce, or None when it has already happened (one rank, gloo)."""
    import torch.distributed as dist
    if world == 1:
        return None
    if dist.get_backend(group) == "nccl":
        return dist.all_reduce(bound, op=dist.ReduceOp.MIN, group=group, async_op=True)
    exchange_floor(bound, world, group)
    return None


def split_pairs(n_pairs: int, world: int, rank: int) -> Tuple[int, int]:
    """Even split of the flattened (query, candidate) pair list of the rerank step across ranks
    ("pairs B x 200 are split evenly across ranks after the merge"): the same rule as shard_bounds."""
    return shard_bounds(n_pairs, world, rank)


def exchange_scores(local, n_pairs: int, world: int, group=None):
    """The second, tiny all-gather of the rerank flow (only when rerank_k > 0): every rank contributes the
    reranker scores of its share of the pairs (float32 tensor of split_pairs' length); returns the full
    (n_pairs,) tensor, identical on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    share = (n_pairs + world - 1) // world                     # shares differ by at most one pair: pad to the longest
    buf = torch.zeros(share, dtype=torch.float32, device=local.device)
    buf[:local.numel()] = local
    out = torch.empty(world * share, dtype=torch.float32, device=local.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, buf, group=group)
    elif buf.is_cuda:                                             # gloo rehearsal on a GPU box
        host = torch.empty(world * share, dtype=torch.float32)
        dist.all_gather([host[r * share:(r + 1) * share] for r in range(world)], buf.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather([out[r * share:(r + 1) * share] for r in range(world)], buf, group=group)
    parts = []
    for r in range(world):
        lo, hi = split_pairs(n_pairs, world, r)
        parts.append(out[r * share:r * share + (hi - lo)])
    return torch.cat(parts)


@dataclass
class PendingBatch:
    """What ShardedSearcher.submit leaves for finish: the batch's parameters, its payload and the started all-gather."""
    B: int
    k: int
    pool: int
    pool_local: int
    rr_k: int
    w: FusionWeights
    lay: "PayloadLayout"
    buf: object
    pending: "PendingExchange"
    gate_fn: object = None
    rerank_fn: object = None
    # the overlapped form (ShardedSearcher.enable_overlap): stage 0 = scanned (K1's scan on the scans' stream, the floor's
    # all-reduce started), 1 = payload built and its all-gather started, 2 = finished
    stage: int = 1
    slot: int = -1
    bound: object = None
    floor_work: object = None
    terms: object = None
    bm25_mode: str = "forward"
    plain_q: object = None        # set when K1 could not be split on this rank: the queries of the whole K1 stage 1 then runs


class _Overlap:
    """Streams and ring buffers of the overlapped shard step: scans on a stream masked to the first ``scan_cus`` CUs, every
    batch's tail (selection against the floor, K2, metadata gather, payload all-gather, merge + K3) on a stream masked to the
    rest, three batches deep (include/rr_hip.h: scan slots)."""
    SLOTS = 3

    def __init__(self, searcher: HybridSearcher, scan_cus: int):
        import torch
        lib, dev = searcher.lib, searcher.device
        total = torch.cuda.get_device_properties(dev).multi_processor_count
        if not (0 < scan_cus < total) or scan_cus % 32 or (total - scan_cus) % 32:
            # the dispatcher deals one-per-CU workgroups to the 32 shader engines in turn: a share that is not a multiple of
            # 32 leaves engines with CUs that get two of them (measured: 240 CUs take twice the time of 224)
            raise ValueError(f"scan_cus {scan_cus} must be a multiple of 32 inside (0, {total})")
        self.scan_cus, self.handles = scan_cus, []
        streams = []
        for first, n in ((0, scan_cus), (scan_cus, total - scan_cus)):
            h = C.c_void_p()
            _lib.check(lib.rr_stream_create_cu_range(dev.index, first, n, C.byref(h)), "rr_stream_create_cu_range")
            self.handles.append(h)
            streams.append(torch.cuda.ExternalStream(h.value, device=dev))
        self.scan_stream, self.tail_stream = streams
        _lib.check(lib.rr_index_set_scan_cus(searcher.index.handle, scan_cus), "rr_index_set_scan_cus")
        self.lib, self.index = lib, searcher.index
        self.seq = 0
        self.pending = None            # the ticket whose payload has not been built yet (stage 0)
        self.unfinished = 0
        self.rings = {}                # (B, pool_local, pool, k, world) -> per-slot buffers

    def buffers(self, key, make):
        if key not in self.rings:
            self.rings[key] = [make() for _ in range(self.SLOTS)]
        return self.rings[key]

    def close(self):
        import torch
        torch.cuda.synchronize()
        if self.index.handle:
            _lib.check(self.lib.rr_index_set_scan_cus(self.index.handle, 0), "rr_index_set_scan_cus")
        for h in self.handles:
            _lib.check(self.lib.rr_stream_destroy(h), "rr_stream_destroy")
        self.handles = []


class ShardedSearcher:
    """K1 + K2 per shard, one all-gather, K3 on the merged pool."""

    def __init__(self, searcher: HybridSearcher, n_total_rows: int, rank: int, world: int, group=None):
        lo, hi = shard_bounds(n_total_rows, world, rank)
        ix = searcher.index
        if ix.row_offset != lo or ix.n_rows != hi - lo:
            raise ValueError(f"rank {rank} must hold rows [{lo}, {hi}); its index holds "
                             f"[{ix.row_offset}, {ix.row_offset + ix.n_rows})")
        self.s, self.rank, self.world, self.group, self.n_total = searcher, rank, world, group, n_total_rows
        # diagnostic: run the payload + exchange + merge path even with one rank (bench.py --force-payload)
        self.force_payload = False
        # row shards select against a corpus-wide floor (two-phase K1 + exchange_floor); RR_NO_SHARD_FLOOR=1: every shard
        # selects on its own threshold (one collective less, ~8x the rescoring at 8 shards)
        self.use_floor = os.environ.get("RR_NO_SHARD_FLOOR") is None
        self._ov: Optional[_Overlap] = None

    def enable_overlap(self, scan_cus: Optional[int] = None) -> bool:
        """Runs every batch's tail beside the NEXT batch's scan ("keep K1 -> K2-gather -> allgather -> K3 on-stream
        ... pipeline batches"): the scans get a stream masked to ``scan_cus`` CUs (a multiple of 32), the tails one masked to
        the rest (rr_stream_create_cu_range), and submit / finish become a three-stage pipeline -- scan(i + 1) | payload(i) |
        merge(i - 1).  Answers are bit for bit those of the straight path (tests/test_gpu_sharded.py).
        The masked streams are blocking HIP streams: a caller that enqueues its own commands on the NULL stream (torch's
        default stream) makes every one of them a barrier across both -- run the calling loop on a stream of its own
        (``torch.cuda.set_stream(torch.cuda.Stream())``, as bench.py does).
        ``scan_cus`` None: ON for real row shards (world > 1) with the split the full-step proxy measured best on MI355X
        (tools/shard_step_proxy.py, profiles/r04_overlap_ab.md: per rank-step without the collectives' wire) --
            <= 1.5M rows per shard (8 GPUs at 10M): 128 | 128 CUs   0.433 -> 0.350 ms
            <= 3M   rows           (4 GPUs)       : 160 |  96       0.663 -> 0.584 ms
            <= 6M   rows           (2 GPUs)       : 192 |  64       1.131 -> 1.016 ms
        and OFF for one rank / larger shards (10M rows on one GPU lose: the tail needs the CUs there and whatever runs beside
        the scan slows it).  RR_TAIL_OVERLAP_CUS=n forces a split (also with one rank), RR_NO_TAIL_OVERLAP=1 turns it off.
        Returns whether the overlap is on."""
        if self._ov is not None:
            return True
        if os.environ.get("RR_NO_TAIL_OVERLAP") is not None:
            return False
        forced = os.environ.get("RR_TAIL_OVERLAP_CUS")
        if scan_cus is None and forced:
            scan_cus = int(forced)
        if scan_cus is None and self.world > 1:
            n = self.s.index.n_rows
            scan_cus = 128 if n <= 1_500_000 else 160 if n <= 3_000_000 else 192 if n <= 6_000_000 else None
        if scan_cus is None:
            return False
        self._ov = _Overlap(self.s, scan_cus)
        return True

    def disable_overlap(self) -> None:
        if self._ov is not None:
            self._flush_overlap()
            self._ov.close()
            self._ov = None

    def local_scan(self, q_dev, pool_local: int):
        """Phase 1 of K1 on this rank: the scan and this shard's bound per query (None: the call cannot be split)."""
        # the shards' kth best rows together hold >= pool rows for any kth >= ceil(pool / world).  A shard is only sure of
        # ~8 kth rescored rows (kth opened 8-row M-tiles): with kth >= ceil(pool / 8) + 1 its own list of `pool` rows
        # fills without the exact fallback also at world >= 8 (ceil(150 / 8) = 19 M-tiles = 152 rows would just do)
        self._flush_overlap()
        kth = max((pool_local + self.world - 1) // self.world, (pool_local + 7) // 8 + 1)
        return self.s.dense_scan(q_dev, pool_local, min(kth, pool_local))

    def local_payload(self, q_dev, term_id_lists, pool_local: int, bm25_mode: str = "forward", floor=None, buf=None):
        """K1 (+ the floor exchange when there is more than one shard) + K2 + metadata gather into the payload buffer of
        this rank.  ``floor``: tests that play several shards in one process pass the minimum they formed themselves
        (after calling local_scan on every shard); ``False`` = plain K1."""
        import torch
        self._flush_overlap()
        s = self.s
        B = q_dev.shape[0]
        lay = PayloadLayout(B, pool_local)
        if buf is None:                 # (``buf``: a caller-owned uint8 buffer of lay.nbytes to build the payload in)
            buf = torch.empty(lay.nbytes, dtype=torch.uint8, device=s.device)
        v = lay.views(buf)
        import torch.distributed as dist
        if floor is None and self.world > 1 and self.use_floor and dist.is_available() and dist.is_initialized():
            bound = self.local_scan(q_dev, pool_local)
            # every rank takes the same branch: whether a call can be split depends on the batch and on the shard sizes
            # only through properties all shards share (n_queries, pool; a shard too small for the filter path reports
            # -inf bounds instead) -- a rank that could not split contributes -inf, i.e. no floor at all
            mine = bound if bound is not None else torch.full((B,), float("-inf"), dtype=torch.float32, device=s.device)
            floor = exchange_floor(mine, self.world, self.group)
            if bound is None:
                floor = False
        if floor is None or floor is False:
            s.dense_pool(q_dev, pool_local, out=(v["rows"], v["dense"]))   # K1 writes the payload in place
        else:
            s.dense_select(q_dev, pool_local, floor, out=(v["rows"], v["dense"]))
        s.bm25_at(term_id_lists, v["rows"], bm25_mode, out=v["bm25"])  # K2 too
        _lib.check(s.lib.rr_index_gather_meta_dev(
            s.index.handle, C.c_void_p(v["rows"].data_ptr()), B * pool_local,
            C.c_void_p(v["n"].data_ptr()), C.c_void_p(v["avg"].data_ptr()),
            C.c_void_p(v["l1p"].data_ptr()), s._stream()), "rr_index_gather_meta_dev")
        return lay, buf

    def search_batch_dev(self, q_dev, term_id_lists: Optional[Sequence[Sequence[int]]], k: int,
                         weights: Optional[FusionWeights] = None, pool_floor: int = 150,
                         bm25_mode: str = "forward", rerank_k: int = 0, gate_fn=None, rerank_fn=None):
        """Device-resident queries in, device tensors out: (pool_rows, columns, order).

        ``gate_fn(rows (B, pool) global rows, numpy) -> (B, pool) float32`` and
        ``rerank_fn(query_idx (n,), rows (n,)) -> (n,) float32`` (scores of the (query, product) pairs given) are
        host callables over the MERGED pool (app/app_product_search.py:271-303): the shards' candidates are
        merged first (K3 with k = pool gives the pool order), gate strings are matched on the host, the
        B x rr_k reranker pairs are split evenly across the ranks, their scores gathered with a second tiny
        all-gather, and K3 runs again on the same gathered payload with the pool-aligned columns."""
        import torch
        w = weights or FusionWeights()
        B = q_dev.shape[0]
        pool = min(max(k, rerank_k, pool_floor), self.n_total)
        pool_local = min(pool, self.s.index.n_rows)
        rr_k = min(rerank_k, pool)
        if self.world > 1:
            # every rank must contribute the same count for the strided addressing
            assert pool_local == pool, "each shard needs at least `pool` rows"
        tl = term_id_lists if term_id_lists is not None else [[] for _ in range(B)]
        s = self.s
        two_pass = gate_fn is not None or (rerank_fn is not None and rr_k > 0)
        if self.world == 1 and not self.force_payload:
            # nothing to exchange: K1 -> K2 -> K3 straight through, metadata read from the index
            self._flush_overlap()
            rows, dense = s.dense_pool(q_dev, pool)
            bm = s.bm25_at(tl, rows, bm25_mode)
            gate = rerank = None
            if two_pass:
                gate, rerank = self._pool_columns(rows, B, pool, rr_k, gate_fn, rerank_fn)
            return s.fuse(HybridSearcher.make_params(w, min(k, pool), pool, pool, rr_k), B, rows, dense, bm,
                          None, rerank, None, gate)
        return self.finish(self.submit(q_dev, term_id_lists, k, weights, pool_floor, bm25_mode, rerank_k, gate_fn, rerank_fn))

    def submit(self, q_dev, term_id_lists, k: int, weights: Optional[FusionWeights] = None, pool_floor: int = 150,
               bm25_mode: str = "forward", rerank_k: int = 0, gate_fn=None, rerank_fn=None) -> "PendingBatch":
        """First half of search_batch_dev for row shards: K1 (+ floor exchange) + K2 + metadata gather on this rank, then the
        payload all-gather is STARTED (exchange_start) and the call returns.  `finish(ticket)` waits for the collective on
        the stream and runs the merge (K3).  Calling submit(batch i + 1) before finish(batch i) puts batch i's all-gather
        under batch i + 1's K1 ("pipeline batches so the all-gather of batch i overlaps K1 of batch i+1"):

            t = s.submit(q[0], ...)
            for i in range(1, n):
                t_next = s.submit(q[i], ...)
                res[i - 1] = s.finish(t)
                t = t_next
            res[n - 1] = s.finish(t)

        Every answer is bit for bit the one search_batch_dev gives (same kernels, same order per batch)."""
        w = weights or FusionWeights()
        B = q_dev.shape[0]
        pool = min(max(k, rerank_k, pool_floor), self.n_total)
        pool_local = min(pool, self.s.index.n_rows)
        rr_k = min(rerank_k, pool)
        if self.world > 1:
            assert pool_local == pool, "each shard needs at least `pool` rows"
        tl = term_id_lists if term_id_lists is not None else [[] for _ in range(B)]
        two_pass = gate_fn is not None or (rerank_fn is not None and rr_k > 0)
        if self._ov is not None:
            if not two_pass:
                t = self._submit_overlapped(q_dev, tl, B, k, pool, pool_local, w, bm25_mode)
                if t is not None:
                    return t
            self._flush_overlap()      # a batch the pipeline does not take (host columns, a call K1 cannot split): in order, behind it
        lay, buf = self.local_payload(q_dev, tl, pool_local, bm25_mode)
        pending = exchange_start(buf, self.world, self.group)
        return PendingBatch(B, k, pool, pool_local, rr_k, w, lay, buf, pending, gate_fn, rerank_fn)

    # ------------------------------------------------------------------ the overlapped form
    def _flush_overlap(self) -> None:
        """Builds the payload of the ticket still at stage 0 (its scan is parked in a slot: a plain K1 call would void it)."""
        ov = self._ov
        if ov is not None and ov.pending is not None:
            t, ov.pending = ov.pending, None
            self._build_payload(t)

    def _submit_overlapped(self, q_dev, tl, B, k, pool, pool_local, w, bm25_mode):
        import torch
        import torch.distributed as dist
        ov, s = self._ov, self.s
        if ov.unfinished > 2:
            raise RuntimeError("the overlapped pipeline is three batches deep: finish() earlier tickets before submitting more")
        slot = ov.seq % ov.SLOTS
        with_floor = self.world > 1 and self.use_floor and dist.is_available() and dist.is_initialized()
        kth = 0
        if with_floor:       # (local_scan's rule)
            kth = min(max((pool_local + self.world - 1) // self.world, (pool_local + 7) // 8 + 1), pool_local)
        cur = torch.cuda.current_stream(s.device)
        ready = torch.cuda.Event()
        ready.record(cur)                                   # (the caller's stream produced q_dev)
        with torch.cuda.stream(ov.scan_stream):
            ov.scan_stream.wait_event(ready)
            bound_ring = ov.buffers(("bound", B), lambda: torch.empty((B,), dtype=torch.float32, device=s.device))
            bound = s.dense_scan_slot(slot, q_dev, pool_local, kth, bound_out=bound_ring[slot])
            plain_q = None
            if bound is None:
                if not with_floor:
                    return None                             # (no floor exchange to keep in step: the caller's plain path)
                # Row shards: whether K1 can be split may depend on THIS rank's rows (no finite row-norm bound: NaN / inf rows;
                # a shard a few rows short of the filter path).  The ranks' collectives must stay in the same order, so this
                # rank stays in the pipeline: it contributes "no floor" (-inf) to the all-reduce, and its stage 1 runs the
                # whole K1 in its slot instead of the selection.
                bound = bound_ring[slot]
                bound.fill_(float("-inf"))
                plain_q = q_dev
            work = exchange_floor_start(bound, self.world, self.group) if with_floor else None
        ov.seq += 1
        ov.unfinished += 1
        lay = PayloadLayout(B, pool_local)
        t = PendingBatch(B, k, pool, pool_local, 0, w, lay, None, None, stage=0, slot=slot,
                         bound=bound if with_floor else None, floor_work=work, terms=tl, bm25_mode=bm25_mode, plain_q=plain_q)
        prev, ov.pending = ov.pending, t
        if prev is not None:
            self._build_payload(prev)                       # batch i - 1's tail goes to its stream beside this scan
        return t

    def _build_payload(self, t: "PendingBatch") -> None:
        """Stage 1 on the tails' stream: selection (against the exchanged floor), K2, metadata gather into the slot's payload
        buffer, and the all-gather is started."""
        import torch
        ov, s = self._ov, self.s
        B, pool_local, lay = t.B, t.pool_local, t.lay
        ring = ov.buffers(("payload", lay.nbytes, self.world), lambda: (
            torch.empty(lay.nbytes, dtype=torch.uint8, device=s.device),
            torch.empty((self.world, lay.nbytes), dtype=torch.uint8, device=s.device)))
        buf, gathered = ring[t.slot]
        v = lay.views(buf)
        with torch.cuda.stream(ov.tail_stream):
            if t.floor_work is not None:
                t.floor_work.wait()                         # (orders this stream behind the all-reduce, not the host)
            if t.plain_q is not None:
                s.dense_pool(t.plain_q, pool_local, out=(v["rows"], v["dense"]), slot=t.slot)
            else:
                s.dense_select_slot(t.slot, B, pool_local, floor=t.bound, out=(v["rows"], v["dense"]))
            s.bm25_at(t.terms, v["rows"], t.bm25_mode, out=v["bm25"])
            _lib.check(s.lib.rr_index_gather_meta_dev(
                s.index.handle, C.c_void_p(v["rows"].data_ptr()), B * pool_local,
                C.c_void_p(v["n"].data_ptr()), C.c_void_p(v["avg"].data_ptr()),
                C.c_void_p(v["l1p"].data_ptr()), s._stream()), "rr_index_gather_meta_dev")
            t.pending = exchange_start(buf, self.world, self.group, out=gathered)
        t.buf, t.stage, t.terms, t.plain_q = buf, 1, None, None

    def _finish_overlapped(self, t: "PendingBatch"):
        import torch
        ov, s = self._ov, self.s
        if t.stage == 0:
            if ov.pending is t:
                ov.pending = None
            self._build_payload(t)
        B, k, pool, pool_local, w, lay = t.B, t.k, t.pool, t.pool_local, t.w, t.lay
        k_out = min(k, pool)
        out_rows, cols, order = ov.buffers(("out", B, pool, k_out), lambda: (
            torch.empty((B, pool), dtype=torch.int64, device=s.device),
            torch.empty((B, 8, pool), dtype=torch.float64, device=s.device),
            torch.empty((B, k_out), dtype=torch.int32, device=s.device)))[t.slot]
        cur = torch.cuda.current_stream(s.device)
        with torch.cuda.stream(ov.tail_stream):
            gathered = t.pending.wait()
            base = gathered.data_ptr()
            ptr = lambda off: C.c_void_p(base + off)
            params = HybridSearcher.make_params(w, k_out, pool, self.world * pool_local, 0,
                                                cand_per_rank=pool_local, stride_bytes=lay.nbytes)
            _lib.check(s.lib.rr_fuse_topk_dev(
                s.index.handle, C.byref(params), B, ptr(lay.off_rows), ptr(lay.off_dense), ptr(lay.off_bm25),
                ptr(lay.off_n), ptr(lay.off_avg), ptr(lay.off_l1p), None, None, None,
                C.c_void_p(out_rows.data_ptr()), C.c_void_p(cols.data_ptr()), C.c_void_p(order.data_ptr()), s._stream()),
                "rr_fuse_topk_dev")
            done = torch.cuda.Event()
            done.record(ov.tail_stream)
        cur.wait_event(done)                                # the caller's stream may read the answer
        t.stage = 2
        ov.unfinished -= 1
        # (the answer lives in the slot's ring buffers: valid until three more batches have been submitted)
        return out_rows, cols, order

    def finish(self, t: "PendingBatch"):
        """Second half: the gathered payload is merged and fused (K3; with a gate / reranker: merge, host columns, K3 again)."""
        import torch
        if t.slot >= 0:
            return self._finish_overlapped(t)
        s = self.s
        B, k, pool, pool_local, rr_k, w, lay = t.B, t.k, t.pool, t.pool_local, t.rr_k, t.w, t.lay
        gate_fn, rerank_fn = t.gate_fn, t.rerank_fn
        two_pass = gate_fn is not None or (rerank_fn is not None and rr_k > 0)
        gathered = t.pending.wait()
        base = gathered.data_ptr()
        ptr = lambda off: C.c_void_p(base + off)
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None

        def fuse(k_out, rerank, gate, rr):
            params = HybridSearcher.make_params(w, k_out, pool, self.world * pool_local, rr,
                                                cand_per_rank=pool_local, stride_bytes=lay.nbytes)
            out_rows = torch.empty((B, pool), dtype=torch.int64, device=s.device)
            cols = torch.empty((B, 8, pool), dtype=torch.float64, device=s.device)
            order = torch.empty((B, params.k), dtype=torch.int32, device=s.device)
            _lib.check(s.lib.rr_fuse_topk_dev(
                s.index.handle, C.byref(params), B, ptr(lay.off_rows), ptr(lay.off_dense), ptr(lay.off_bm25),
                ptr(lay.off_n), ptr(lay.off_avg), ptr(lay.off_l1p), p(rerank), None, p(gate),
                p(out_rows), p(cols), p(order), s._stream()), "rr_fuse_topk_dev")
            return out_rows, cols, order

        if not two_pass:
            res = fuse(min(k, pool), None, None, 0)
        else:
            merged_rows, _, _ = fuse(pool, None, None, 0)           # pass A: the merge alone fixes the pool order
            gate, rerank = self._pool_columns(merged_rows, B, pool, rr_k, gate_fn, rerank_fn)
            res = fuse(min(k, pool), rerank, gate, rr_k)             # pass B: same payload + pool-aligned columns
        self._keep = (gathered, t.buf)   # alive until the stream has consumed them
        return res

    def _pool_columns(self, rows_dev, B: int, pool: int, rr_k: int, gate_fn, rerank_fn):
        """Gate factors (every rank, all pairs: string work on replicated texts) and reranker scores (this rank's
        share of the B x rr_k pairs + the score all-gather) for the merged pool; device tensors or None."""
        import torch
        s = self.s
        rows_h = rows_dev.cpu().numpy()
        gate = rerank = None
        if gate_fn is not None:
            g = np.ascontiguousarray(gate_fn(rows_h), dtype=np.float32)
            gate = torch.from_numpy(g).to(s.device)
        if rerank_fn is not None and rr_k > 0:
            n_pairs = B * rr_k
            lo, hi = split_pairs(n_pairs, self.world, self.rank)
            idx = np.arange(lo, hi)
            qi, ci = idx // rr_k, idx % rr_k
            mine = np.asarray(rerank_fn(qi, rows_h[qi, ci]), dtype=np.float32).reshape(-1)
            if mine.shape[0] != hi - lo:
                raise ValueError("rerank_fn must return one score per pair")
            full = exchange_scores(torch.from_numpy(mine).to(s.device), n_pairs, self.world, self.group)
            rerank = torch.zeros((B, pool), dtype=torch.float32, device=s.device)
            rerank[:, :rr_k] = full.view(B, rr_k)
        return gate, rerank
