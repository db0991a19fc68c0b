"""The drop-in search engine: the reference's Python call boundary over the HIP kernels.

Boundary kept:
  cosine_similarity_search(query_vector, embeddings_matrix, top_k)   utils.py:111-124
  run_search(query, k, rerank_k, w_dense, w_bm25, w_rerank, w_prior, w_best, prior_C,
             use_snips, max_scan, min_reviews, gate_penalty)
             -> (DataFrame, snips, dbg)                 app/app_product_search.py:245-317
  search(query, k, alpha)                               north-star sugar over run_search
  CLI flavour (pool floor 100, no trust factor)         app/test.py:228-342

Per query batch the device does K1 (dense scan + exact top-pool), K2 (BM25 at the
pool), K3 (min-max, priors, trust, blend, gate, top-k) back to back on one HIP stream;
the host only tokenises, matches gate strings and builds the result frame.
torch is used for device buffers and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from . import _lib, text
from .bm25 import BM25Corpus, BM25Index
from .index import MAX_POOL, ProductIndex

APP_POOL_FLOOR = 150    # app/app_product_search.py:253
CLI_POOL_FLOOR = 100    # app/test.py:238
APP_TRUST_SAT = 80      # app/app_product_search.py:303
COLUMN_NAMES = ("_dense", "_bm25", "_prior", "_rerank", "_best", "_gate", "_trust", "_final")
_FLOAT32_COLUMNS = {"_dense", "_bm25", "_best", "_gate", "_trust", "_final"}


@dataclass
class FusionWeights:
    """run_search's scoring parameters (defaults: config.py:64-72, prior_C app/...:402)."""
    w_dense: float = 0.55
    w_bm25: float = 0.20
    w_rerank: float = 0.20
    w_prior: float = 0.20
    w_best: float = 0.10
    prior_C: float = 20.0
    min_reviews: int = 8
    gate_penalty: float = 0.5
    trust_sat: int = APP_TRUST_SAT
    apply_trust: bool = True


@dataclass
class BatchResult:
    """Raw result of one query batch (numpy, host)."""
    pool_rows: np.ndarray      # (B, pool) global rows in pool order (dense desc, row asc)
    columns: np.ndarray        # (B, 8, pool) float64: COLUMN_NAMES
    order: np.ndarray          # (B, k) pool positions of the top-k, best first
    dense_raw: np.ndarray      # (B, pool) float32 raw cosine scores
    bm25_raw: np.ndarray       # (B, pool) float32 raw BM25 scores
    pool: int
    k: int
    best_ids: Optional[np.ndarray] = None    # (B, pool) best review id per candidate, -1 = none
    best_raw: Optional[np.ndarray] = None    # (B, pool) its raw score

    def topk_rows(self) -> np.ndarray:
        return np.take_along_axis(self.pool_rows, self.order.astype(np.int64), axis=1)

    def topk_column(self, name: str) -> np.ndarray:
        c = self.columns[:, COLUMN_NAMES.index(name), :]
        return np.take_along_axis(c, self.order.astype(np.int64), axis=1)


def _torch():
    import torch
    return torch


# RR_NO_KERNEL_COPY=1 (A/B): the batch's small transfers as copy commands (hipMemcpyAsync) instead of through
# rr_copy_segments_dev / K1 reading pinned queries
_KERNEL_COPIES = os.environ.get("RR_NO_KERNEL_COPY") is None


class StagedTerms:
    """Token ids of a batch already on the device (HybridSearcher.stage_batch)."""
    __slots__ = ("ids", "off", "slot", "n_ids")

    def __init__(self, ids, off, slot, n_ids):
        self.ids, self.off, self.slot, self.n_ids = ids, off, slot, int(n_ids)


class StagedBatch:
    """One batch's inputs on the device: ``q`` (B, dim) float32, ``terms`` (StagedTerms or None), ``ready`` (event)."""
    __slots__ = ("q", "terms", "ready", "slot")

    def __init__(self, q, terms, ready, slot):
        self.q, self.terms, self.ready, self.slot = q, terms, ready, slot


class HybridSearcher:
    """K1 -> K2 -> K3 on one GPU over a ProductIndex (+ optional BM25Index)."""

    def __init__(self, index: ProductIndex, bm25: Optional[BM25Index] = None):
        if not index.has_meta:
            raise ValueError("the index needs metadata (ProductIndex.set_meta) before searching")
        self.index, self.bm25 = index, bm25
        self.lib = _lib.load()
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("no GPU visible: the search path runs on the device only")
        self.device = torch.device("cuda", index.device)
        self._stage_slots = [{"host": None, "dev": None, "event": None, "consumed": None} for _ in range(8)]
        self._stage_next = 0
        self._q_slots = [{"dev": None, "free": None} for _ in range(4)]
        self._q_next = 0
        self._in = None                       # input stream of stage_batch (created on first use)
        # the staging rings are shared Python state: one searcher serves every Streamlit session thread (frontend.py), and
        # the C library's per-handle mutex does not cover slot selection / the pinned buffers
        self._ring_lock = threading.RLock()

    # -------------------------------------------------------------- device steps
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def copy_segments(self, pairs) -> None:
        """ONE kernel launch on the current stream that copies ``src`` into ``dst`` for every (dst, src) pair of tensors
        (at most four; same shape and dtype; contiguous, or 2-D with contiguous rows).  Either side may be a PINNED host
        tensor: the bytes then cross PCIe as the kernel's own loads / stores (rr_copy_segments_dev) instead of as one
        copy command per tensor -- a batch's token ids in, its rows / order / final scores out."""
        segs = (_lib.CopySeg * len(pairs))()
        for i, (dst, src) in enumerate(pairs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("copy_segments: shape / dtype mismatch")
            for t in (dst, src):
                if not t.is_cuda and not t.is_pinned():
                    raise ValueError("copy_segments: host tensors must be pinned")
            esz = dst.element_size()
            if dst.is_contiguous() and src.is_contiguous():
                rows, row_bytes, dp, sp = 1, dst.numel() * esz, dst.numel() * esz, dst.numel() * esz
            else:
                if dst.dim() != 2 or dst.stride(1) != 1 or src.stride(1) != 1:
                    raise ValueError("copy_segments: tensors must be contiguous or 2-D with contiguous rows")
                rows, row_bytes = dst.shape[0], dst.shape[1] * esz
                dp, sp = dst.stride(0) * esz, src.stride(0) * esz
            segs[i] = _lib.CopySeg(dst.data_ptr(), src.data_ptr(), row_bytes, rows, dp, sp)
        _lib.check(self.lib.rr_copy_segments_dev(segs, len(pairs), self.device.index, self._stream()), "rr_copy_segments_dev")

    def dense_pool(self, q_dev, pool: int, out=None, slot: int = 0):
        """K1 on device tensors: (rows int64 (B,pool), scores float32 (B,pool)).  ``q_dev``: (B, dim) float32 on the
        device -- or in PINNED host memory (K1's first kernel then reads the queries over PCIe itself: no copy command).
        ``out`` = (rows, scores) tensors to write into (e.g. views of a shard payload).  ``slot``: the scan slot whose state
        the call uses (rr_dense_topk_slot_dev; 0 unless other batches have scans parked)."""
        torch = _torch()
        B = q_dev.shape[0]
        if out is not None:
            rows, dense = out
        else:
            rows = torch.empty((B, pool), dtype=torch.int64, device=self.device)
            dense = torch.empty((B, pool), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.rr_dense_topk_slot_dev(self.index.handle, int(slot), C.c_void_p(q_dev.data_ptr()), B, pool,
                                                   C.c_void_p(rows.data_ptr()), C.c_void_p(dense.data_ptr()),
                                                   self._stream()), "rr_dense_topk_dev")
        return rows, dense

    def dense_scan(self, q_dev, pool: int, kth: int):
        """Phase 1 of the two-phase K1 of row shards (rr_dense_scan_dev): the scan without its selection.  Returns a
        float32 (B,) device tensor -- per query a lower bound of the score of this shard's kth-best row -- or None when
        the call cannot be split (then use dense_pool)."""
        torch = _torch()
        B = q_dev.shape[0]
        bound = torch.empty((B,), dtype=torch.float32, device=self.device)      # (written in full whenever `applied`)
        applied = C.c_int32(0)
        _lib.check(self.lib.rr_dense_scan_dev(self.index.handle, C.c_void_p(q_dev.data_ptr()), B, pool, int(kth),
                                              C.c_void_p(bound.data_ptr()), C.byref(applied), self._stream()),
                   "rr_dense_scan_dev")
        return bound if applied.value else None

    def dense_select(self, q_dev, pool: int, floor, out=None):
        """Phase 2 (rr_dense_select_dev): the selection of the scan dense_scan left behind, with the floor the shards
        agreed on (float32 (B,) device tensor).  Lists of `pool` rows: see include/rr_hip.h."""
        torch = _torch()
        B = q_dev.shape[0]
        if out is not None:
            rows, dense = out
        else:
            rows = torch.empty((B, pool), dtype=torch.int64, device=self.device)
            dense = torch.empty((B, pool), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.rr_dense_select_dev(self.index.handle, C.c_void_p(q_dev.data_ptr()), B, pool,
                                                C.c_void_p(floor.data_ptr()), C.c_void_p(rows.data_ptr()),
                                                C.c_void_p(dense.data_ptr()), self._stream()), "rr_dense_select_dev")
        return rows, dense

    def dense_scan_slot(self, slot: int, q_dev, pool: int, kth: int = 0, bound_out=None):
        """Phase 1 of the pipelined K1 (rr_dense_scan_slot_dev) on the CURRENT stream: the scan of ``q_dev`` into scan slot
        ``slot`` (0 .. 2).  ``kth`` > 0 (row shards): also the per-query bound of dense_scan, returned as a float32 (B,)
        tensor (``bound_out`` if given); ``kth`` = 0: returns True.  None when the call cannot be split (then use dense_pool)."""
        torch = _torch()
        B = q_dev.shape[0]
        bound = None
        if kth > 0:
            bound = bound_out if bound_out is not None else torch.empty((B,), dtype=torch.float32, device=self.device)
        applied = C.c_int32(0)
        _lib.check(self.lib.rr_dense_scan_slot_dev(self.index.handle, int(slot), C.c_void_p(q_dev.data_ptr()), B, pool, int(kth),
                                                   C.c_void_p(bound.data_ptr()) if kth > 0 else None, C.byref(applied),
                                                   self._stream()), "rr_dense_scan_slot_dev")
        if not applied.value:
            return None
        return bound if kth > 0 else True

    SELECT_LIST, SELECT_RESCORE, SELECT_ORDER = 1, 2, 4        # include/rr_hip.h: RR_SELECT_*

    def dense_select_slot(self, slot: int, B: int, pool: int, floor=None, out=None, parts: int = 7):
        """Phase 2 (rr_dense_select_part_dev) on the CURRENT stream -- which may be another one than the scan's, and another
        one per part (SELECT_LIST | SELECT_RESCORE | SELECT_ORDER): the library orders them with events.  Returns
        (rows, dense) when SELECT_ORDER is among the parts, else None."""
        torch = _torch()
        rows = dense = None
        if parts & self.SELECT_ORDER:
            if out is not None:
                rows, dense = out
            else:
                rows = torch.empty((B, pool), dtype=torch.int64, device=self.device)
                dense = torch.empty((B, pool), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self.lib.rr_dense_select_part_dev(self.index.handle, int(slot), int(parts), B, pool, p(floor), p(rows), p(dense),
                                                     self._stream()), "rr_dense_select_part_dev")
        return (rows, dense) if parts & self.SELECT_ORDER else None

    def bm25_at(self, term_id_lists: Sequence[Sequence[int]], rows_dev, mode: str = "forward", out=None):
        """K2 on device tensors: float32 (B, pool) raw BM25 at the candidate rows."""
        torch = _torch()
        B, pool = rows_dev.shape
        staged = isinstance(term_id_lists, StagedTerms)
        flat_form = isinstance(term_id_lists, tuple) and len(term_id_lists) == 2 and isinstance(term_id_lists[1], np.ndarray)
        empty = (term_id_lists.n_ids == 0) if staged else \
            (len(term_id_lists[0]) == 0) if flat_form else not any(len(t) for t in term_id_lists)
        if self.bm25 is None or empty:
            # no index / no tokens -> zeros (app/app_product_search.py:202,204)
            if out is None:
                return torch.zeros((B, pool), dtype=torch.float32, device=self.device)
            out.zero_()
            return out
        if out is None:
            out = torch.empty((B, pool), dtype=torch.float32, device=self.device)
        ids_dev, off_dev = (term_id_lists.ids, term_id_lists.off) if staged else self._stage_terms(term_id_lists)[:2]
        _lib.check(self.lib.rr_bm25_scores_at_dev(
            self.bm25.handle, C.c_void_p(ids_dev.data_ptr()), C.c_void_p(off_dev.data_ptr()), B,
            C.c_void_p(rows_dev.data_ptr()), pool, {"forward": 0, "postings": 1}[mode],
            C.c_void_p(out.data_ptr()), self._stream()), "rr_bm25_scores_at_dev")
        return out

    def _stage_terms(self, term_id_lists):
        """Token ids of a batch as device arrays (flat ids + offsets), uploaded on every call through a
        small ring of pinned host buffers (one async copy per batch; nothing is cached across calls, so a
        caller may reuse and mutate its lists).  ``term_id_lists`` is a sequence of per-query id
        sequences, or an already flattened ``(ids int32, offsets int32[B+1])`` pair of numpy arrays."""
        torch = _torch()
        if isinstance(term_id_lists, tuple) and len(term_id_lists) == 2 and isinstance(term_id_lists[1], np.ndarray):
            flat = np.ascontiguousarray(term_id_lists[0], dtype=np.int32).reshape(-1)
            off = np.ascontiguousarray(term_id_lists[1], dtype=np.int32).reshape(-1)
        else:
            B = len(term_id_lists)
            off = np.zeros(B + 1, dtype=np.int32)
            np.cumsum([len(t) for t in term_id_lists], out=off[1:])
            flat = (np.concatenate([np.asarray(t, dtype=np.int32).reshape(-1) for t in term_id_lists])
                    if off[-1] else np.zeros(0, dtype=np.int32))
        n_off, n_ids = off.shape[0], flat.shape[0]
        need = n_off + max(n_ids, 1)
        with self._ring_lock:
            return self._stage_terms_locked(off, flat, n_off, n_ids, need)

    def _stage_terms_locked(self, off, flat, n_off, n_ids, need):
        torch = _torch()
        slot = self._stage_slots[self._stage_next % len(self._stage_slots)]
        self._stage_next += 1
        if slot["host"] is None or slot["host"].numel() < need:
            # (pinning host memory is slow and synchronises: every slot of the ring is sized at once, on first use)
            cap = max(2 * need, 16384)
            torch.cuda.synchronize(self.device)      # (buffers change hands between streams only when nothing is in flight)
            for sl in self._stage_slots:
                if sl["host"] is None or sl["host"].numel() < need:
                    if sl["event"] is not None:
                        sl["event"].synchronize()
                    sl["host"] = torch.empty(cap, dtype=torch.int32).pin_memory()
                    sl["dev"] = torch.empty(cap, dtype=torch.int32, device=self.device)
                    sl["event"] = torch.cuda.Event()
        else:
            slot["event"].synchronize()          # the copy that last used this pinned buffer has finished
        h = slot["host"].numpy()
        h[:n_off] = off
        h[n_off:n_off + n_ids] = flat
        cur = torch.cuda.current_stream(self.device)
        if slot["consumed"] is not None:         # (input stream: the K2 launch that read this device buffer last)
            cur.wait_event(slot["consumed"])
            slot["consumed"] = None
        if _KERNEL_COPIES:
            self.copy_segments([(slot["dev"][:need], slot["host"][:need])])
        else:
            slot["dev"][:need].copy_(slot["host"][:need], non_blocking=True)
        slot["event"].record(cur)
        return slot["dev"][n_off:n_off + max(n_ids, 1)], slot["dev"][:n_off], slot, n_ids

    def stage_batch(self, q_host, term_id_lists=None) -> "StagedBatch":
        """Uploads one batch's inputs on the searcher's INPUT stream: the query vectors (``q_host``: pinned float32
        (B, dim) tensor) into a device buffer of a small ring, the token ids through the pinned staging ring.  The
        caller makes its compute stream wait for ``.ready``, runs the batch with ``.q`` / ``.terms`` and then calls
        ``release(batch)`` (records when the buffers may be overwritten).  With the answer copied back on a third
        stream, batch i + 1's uploads and batch i - 1's downloads run under batch i's kernels instead of between
        them (~90 us per 256-query batch on one stream)."""
        torch = _torch()
        with self._ring_lock:
            return self._stage_batch_locked(q_host, term_id_lists)

    def _stage_batch_locked(self, q_host, term_id_lists):
        torch = _torch()
        if self._in is None:
            self._in = torch.cuda.Stream(device=self.device)
        slot = self._q_slots[self._q_next % len(self._q_slots)]
        self._q_next += 1
        if slot["dev"] is None or slot["dev"].shape != q_host.shape:
            # a fresh block of the caching allocator may still be read by kernels queued on the compute stream (it is
            # free in THAT stream's order only): the input stream must not write it before they have run
            slot["dev"] = torch.empty(tuple(q_host.shape), dtype=torch.float32, device=self.device)
            slot["free"] = torch.cuda.Event()
            slot["free"].record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._in):
            if slot["free"] is not None:
                self._in.wait_event(slot["free"])
            slot["dev"].copy_(q_host, non_blocking=True)
            terms = None
            if term_id_lists is not None:
                ids, off, tslot, n_ids = self._stage_terms(term_id_lists)
                terms = StagedTerms(ids, off, tslot, n_ids)
            ready = torch.cuda.Event()
            ready.record(self._in)
        return StagedBatch(slot["dev"], terms, ready, slot)

    def release(self, batch: "StagedBatch") -> None:
        """After the batch's kernels are enqueued on the current stream: its buffers are free once they have run."""
        torch = _torch()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        batch.slot["free"] = ev
        if batch.terms is not None:
            batch.terms.slot["consumed"] = ev

    def fuse(self, params: "_lib.FuseParams", B: int, rows, dense, bm25, meta=None,
             rerank=None, best=None, gate=None):
        """K3 on device tensors -> (out_rows (B,pool), cols (B,8,pool) f64, order (B,k))."""
        torch = _torch()
        pool, k = params.pool, params.k
        out_rows = torch.empty((B, pool), dtype=torch.int64, device=self.device)
        cols = torch.empty((B, 8, pool), dtype=torch.float64, device=self.device)
        order = torch.empty((B, k), dtype=torch.int32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        n, avg, l1p = meta if meta is not None else (None, None, None)
        _lib.check(self.lib.rr_fuse_topk_dev(
            self.index.handle, C.byref(params), B, p(rows), p(dense), p(bm25), p(n), p(avg), p(l1p),
            p(rerank), p(best), p(gate), p(out_rows), p(cols), p(order), self._stream()),
            "rr_fuse_topk_dev")
        return out_rows, cols, order

    # -------------------------------------------------------------- one batch
    @staticmethod
    def make_params(w: FusionWeights, k: int, pool: int, n_candidates: int, rerank_k: int,
                    cand_per_rank: int = 0, stride_bytes: int = 0, bm25_f64: bool = False) -> "_lib.FuseParams":
        return _lib.FuseParams(
            w_dense=w.w_dense, w_bm25=w.w_bm25, w_rerank=w.w_rerank, w_prior=w.w_prior,
            w_best=w.w_best, prior_C=w.prior_C, min_reviews=int(w.min_reviews),
            trust_sat=int(w.trust_sat), apply_trust=int(bool(w.apply_trust)),
            rerank_active=int(rerank_k > 0), rerank_k=int(rerank_k), k=int(k),
            n_candidates=int(n_candidates), pool=int(pool), cand_per_rank=int(cand_per_rank),
            bm25_f64=int(bool(bm25_f64)), cand_rank_stride_bytes=int(stride_bytes))

    def search_batch(self, qvecs: np.ndarray, term_id_lists: Optional[Sequence[Sequence[int]]],
                     k: int, rerank_k: int = 0, weights: Optional[FusionWeights] = None,
                     pool_floor: int = APP_POOL_FLOOR,
                     gate_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                     rerank_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                     bm25_mode: str = "forward", reviews=None, max_scan: int = 0,
                     bm25_f64: bool = False) -> BatchResult:
        """qvecs (B, dim) float32; term_id_lists: per-query BM25 token ids (None = no BM25).
        gate_fn / rerank_fn map the (B, pool) pool rows to (B, pool) float32 gate factors /
        (B, rr_k) raw reranker scores; they run on the host between K2 and K3.
        ``bm25_f64``: the CLI flavour without a BM25 artefact (float64 zeros column, app/test.py:252)."""
        torch = _torch()
        w = weights or FusionWeights()
        q = np.ascontiguousarray(qvecs, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.index.dim:
            raise ValueError(f"qvecs must be (B, {self.index.dim}); got {q.shape}")
        B = q.shape[0]
        if k < 1:
            raise ValueError("k must be >= 1")
        pool = min(max(k, rerank_k, pool_floor), self.index.n_rows)
        if pool > MAX_POOL:
            raise ValueError(f"pool {pool} exceeds the kernels' limit {MAX_POOL}")
        k_eff = min(k, pool)
        rr_k = min(rerank_k, pool)
        with torch.cuda.device(self.device):
            q_dev = torch.from_numpy(q).to(self.device)
            rows, dense = self.dense_pool(q_dev, pool)
            tl = term_id_lists if term_id_lists is not None else [[] for _ in range(B)]
            bm = self.bm25_at(tl, rows, bm25_mode)
            gate = rerank = best = None
            best_ids = None
            rows_h = None
            if reviews is not None:
                # best review per candidate (csrc/rr_reviews.hip), the whole batch in one call; the
                # reference's iloc[:max_rows] cut is evaluated per query on the device
                best = torch.empty((B, pool), dtype=torch.float32, device=self.device)
                best_ids = torch.empty((B, pool), dtype=torch.int32, device=self.device)
                _lib.check(self.lib.rr_reviews_best_cut_dev(
                    reviews.handle, C.c_void_p(q_dev.data_ptr()), B, C.c_void_p(rows.data_ptr()), pool,
                    self.index.row_offset, int(max_scan), C.c_void_p(best.data_ptr()),
                    C.c_void_p(best_ids.data_ptr()), self._stream()), "rr_reviews_best_cut_dev")
            if gate_fn is not None or (rerank_fn is not None and rr_k > 0):
                rows_h = rows.cpu().numpy() if rows_h is None else rows_h
                if gate_fn is not None:
                    g = np.ascontiguousarray(gate_fn(rows_h), dtype=np.float32)
                    gate = torch.from_numpy(g).to(self.device)
                if rerank_fn is not None and rr_k > 0:
                    r = np.zeros((B, pool), dtype=np.float32)
                    r[:, :rr_k] = np.asarray(rerank_fn(rows_h[:, :rr_k]), dtype=np.float32)
                    rerank = torch.from_numpy(r).to(self.device)
            params = self.make_params(w, k_eff, pool, pool, rr_k, bm25_f64=bm25_f64 and self.bm25 is None)
            out_rows, cols, order = self.fuse(params, B, rows, dense, None if params.bm25_f64 else bm, None,
                                              rerank, best, gate)
            res = BatchResult(out_rows.cpu().numpy(), cols.cpu().numpy(), order.cpu().numpy(),
                              dense.cpu().numpy(), bm.cpu().numpy(), pool, k_eff)
            if best_ids is not None:
                res.best_ids, res.best_raw = best_ids.cpu().numpy(), best.cpu().numpy()
        return res


class SearchEngine:
    """The reference's search API over one GPU.

    meta: DataFrame with ``sku, n_reviews, avg_stars, agg_text`` (+ any other column),
    row-aligned with ``embeddings`` (product_emb_meta.parquet / product_emb.npy).
    bm25_blob: the ``{"skus", "corpus"}`` dict of product_bm25.pkl, or None.
    encoder: object with ``encode([query], normalize_embeddings=True)`` (SentenceTransformer API).
    cross_encoder: object with ``predict(pairs, batch_size=64, show_progress_bar=False)``.
    """

    def __init__(self, meta: pd.DataFrame, embeddings: np.ndarray, bm25_blob: Optional[dict] = None,
                 *, encoder=None, cross_encoder=None, device: int = 0, normalize: bool = True,
                 flavour: str = "app", dtype: str = "f32", reviews: Optional[Tuple] = None):
        if flavour not in ("app", "cli"):
            raise ValueError("flavour must be 'app' or 'cli'")
        if len(meta) != embeddings.shape[0]:
            # app/app_product_search.py:104-107, app/test.py:141-142: hard error
            raise ValueError(f"metadata has {len(meta)} rows but embeddings have "
                             f"{embeddings.shape[0]} rows")
        if "sku" not in meta.columns or "agg_text" not in meta.columns:
            raise ValueError("metadata must have 'sku' and 'agg_text' columns")  # app/test.py:138-139
        self.flavour = flavour
        self.meta = meta.reset_index(drop=True)
        self.encoder, self.cross_encoder = encoder, cross_encoder
        # chunked upload: the matrix may be a memory-mapped product_emb.npy (app/test.py:140)
        self.index = ProductIndex.from_rows(embeddings, device=device, normalize=normalize, dtype=dtype)
        nan = pd.Series([np.nan] * len(self.meta))
        n = pd.to_numeric(self.meta.get("n_reviews", nan), errors="coerce").fillna(0).values
        r = pd.to_numeric(self.meta.get("avg_stars", nan), errors="coerce").values
        self.index.set_meta(n, r)
        self._texts = self.meta["agg_text"].astype(str)
        self.bm25_corpus: Optional[BM25Corpus] = None
        bm25_index = None
        if bm25_blob:
            self.bm25_corpus = BM25Corpus.from_corpus(bm25_blob["corpus"])
            aligned = self.bm25_corpus.select(self._align_bm25([str(s) for s in bm25_blob["skus"]]))
            bm25_index = aligned.to_device(device)
        self.searcher = HybridSearcher(self.index, bm25_index)
        # reviews = (frame with sku/text/stars, (n_reviews, dim) embeddings): reviews_with_embeddings.parquet
        self.reviews = None
        if reviews is not None:
            from .reviews import ReviewIndex
            self.reviews = ReviewIndex(reviews[0], reviews[1], self.meta["sku"].astype(str).tolist(), device=device)

    @classmethod
    def from_artifacts(cls, data_dir, **kw) -> "SearchEngine":
        """Loads product_emb.npy / product_emb_meta.parquet / product_bm25.pkl from ``data_dir``
        (the reference's data/processed layout, app/test.py:21-26) and l2-normalises the rows on
        the GPU like the reference's loaders do on the host (app/test.py:144)."""
        from .artifacts import load_artifacts, load_reviews
        meta, emb, blob = load_artifacts(data_dir)
        kw.setdefault("normalize", True)
        if "reviews" not in kw:
            kw["reviews"] = load_reviews(data_dir)     # None when reviews_with_embeddings.parquet is absent
        return cls(meta, emb, blob, **kw)

    # app: sku -> last position, missing -> 0.0 score (app/app_product_search.py:207-208)
    # cli: same map, but if ANY meta sku is missing the scores are used unpermuted
    #      (ensure_same_order returns None, app/test.py:159-173)
    def _align_bm25(self, bm25_skus: List[str]) -> np.ndarray:
        pos = {s: i for i, s in enumerate(bm25_skus)}
        meta_skus = self.meta["sku"].astype(str).tolist()
        order = np.array([pos.get(s, -1) for s in meta_skus], dtype=np.int64)
        if self.flavour == "cli" and (order < 0).any():
            if len(bm25_skus) < len(meta_skus):
                raise IndexError("BM25 corpus is shorter than the metadata and cannot be used unpermuted")
            order = np.arange(len(meta_skus), dtype=np.int64)
        return order

    # ------------------------------------------------------------------ pieces
    def encode(self, query: str) -> np.ndarray:
        if self.encoder is None:
            raise ValueError("no query encoder was given; use search_by_vector / pass qvec")
        return np.asarray(self.encoder.encode([query], normalize_embeddings=True)[0], dtype=np.float32)

    def _gate_fn(self, query: str, penalty: float):
        groups = text.build_gate_groups(query)

        def fn(rows: np.ndarray) -> np.ndarray:
            out = np.ones(rows.shape, dtype=np.float32)
            if not groups:
                return out
            for b in range(rows.shape[0]):
                texts = self._texts.iloc[rows[b]].str.slice(0, 6000).tolist()
                out[b] = np.array([text.calculate_gate_factor(t, groups, penalty)[0] for t in texts],
                                  dtype=np.float32)
            return out
        return groups, fn

    def _rerank_fn(self, query: str):
        if self.cross_encoder is None:
            return None   # model missing -> zeros (app/app_product_search.py:275)

        def fn(rows: np.ndarray) -> np.ndarray:
            # every (query, text[:2000]) pair of the batch in ONE predict call: the packed forward has no use for
            # per-query mini-batches (app/app_product_search.py:272-278 builds the pairs the same way, per query)
            texts = self._texts.iloc[rows.reshape(-1)].str.slice(0, 2000).tolist()
            pairs = [(query, t) for t in texts]
            scores = np.asarray(self.cross_encoder.predict(pairs, batch_size=64, show_progress_bar=False), dtype=np.float32)
            return scores.reshape(rows.shape)
        return fn

    # ------------------------------------------------------------------ API
    def run_search(self, query: str, k: int, rerank_k: int, w_dense: float, w_bm25: float,
                   w_rerank: float, w_prior: float, w_best: float, prior_C: float,
                   use_snips: bool = False, max_scan: int = 0, min_reviews: int = 8,
                   gate_penalty: float = 0.5, *, qvec: Optional[np.ndarray] = None
                   ) -> Tuple[pd.DataFrame, Dict, Dict]:
        """Same positional / keyword signature and return triple as the reference's
        run_search (app/app_product_search.py:245-248, 312-317); ``qvec`` lets a caller
        supply the query embedding when no encoder is loaded.  With ``use_snips`` and a review
        index (``reviews=`` of the constructor) ``snips`` and ``_best`` are filled like
        _best_snippets does (app/app_product_search.py:285-294, 320-370); otherwise ``snips`` is {}
        and ``_best`` zeros, as in the reference when the review file is absent."""
        app = self.flavour == "app"
        qv = self.encode(query) if qvec is None else np.asarray(qvec, dtype=np.float32)
        toks = text.tokenize_query(query)
        term_ids = None
        if self.searcher.bm25 is not None:
            term_ids = [self.searcher.bm25.term_ids(toks)]
        groups, gate_fn = self._gate_fn(query, gate_penalty)
        w = FusionWeights(w_dense, w_bm25, w_rerank, w_prior, w_best, prior_C, min_reviews,
                          gate_penalty, APP_TRUST_SAT, app)
        res = self.searcher.search_batch(
            qv[None, :], term_ids, k, rerank_k, w,
            pool_floor=APP_POOL_FLOOR if app else CLI_POOL_FLOOR,
            gate_fn=gate_fn if groups else None,
            rerank_fn=self._rerank_fn(query) if rerank_k > 0 else None,
            reviews=self.reviews if use_snips else None, max_scan=max_scan,
            bm25_f64=not app)
        frame = self._frame(res, 0, rerank_k > 0)
        snips = {}
        if res.best_ids is not None:
            skus = self.meta["sku"].astype(str).iloc[res.pool_rows[0]].tolist()
            snips = self.reviews.snippets(skus, res.best_ids[0], res.best_raw[0], 600 if app else 400)
        dbg = {"bm25_active": self.searcher.bm25 is not None, "tokens": toks,
               "groups": [list(g) for g in groups], "pool": max(k, rerank_k,
                                                              APP_POOL_FLOOR if app else CLI_POOL_FLOOR)}
        return frame, snips, dbg

    def _frame(self, res: BatchResult, b: int, rerank_active: bool = False) -> pd.DataFrame:
        top = res.order[b].astype(np.int64)
        out = self.meta.iloc[res.pool_rows[b][top]].reset_index(drop=True).copy()
        for j, name in enumerate(COLUMN_NAMES):
            col = res.columns[b, j, top]
            if name == "_trust" and self.flavour == "cli":
                continue   # the CLI has no trust column (app/test.py:308)
            # `_rerank` is the float32 array z when rerank_k > 0, else the float64 scalar column 0.0
            # (app/app_product_search.py:279-282); `_bm25` is float64 zeros in the CLI without an artefact
            f32 = name in _FLOAT32_COLUMNS or (name == "_rerank" and rerank_active)
            if name == "_bm25" and self.flavour == "cli" and self.searcher.bm25 is None:
                f32 = False
            out[name] = col.astype(np.float32) if f32 else col
        return out

    def search(self, query: str, k: int = 10, alpha: float = 0.5, *,
               qvec: Optional[np.ndarray] = None) -> Tuple[pd.DataFrame, Dict, Dict]:
        """search(query, k, alpha) of BASELINE.json: dense weight alpha, BM25 weight
        1 - alpha, no rerank / prior / best, gate off."""
        return self.run_search(query, k, 0, alpha, 1.0 - alpha, 0.0, 0.0, 0.0, 20.0, False, 0,
                               8, 1.0, qvec=qvec)

    def search_by_vector(self, qvec: np.ndarray, query: str = "", k: int = 10, alpha: float = 0.5):
        return self.search(query, k, alpha, qvec=qvec)

    def cosine_similarity_search(self, query_vector: np.ndarray, top_k: int):
        rows, scores = self.index.dense_topk(np.asarray(query_vector, dtype=np.float32)[None, :], top_k)
        return rows[0], scores[0]


def cosine_similarity_search(query_vector: np.ndarray, embeddings_matrix: np.ndarray,
                             top_k: int, device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """utils.py:111-124 with the matrix copied to the GPU for this call; use
    ProductIndex / SearchEngine to keep it resident across calls."""
    ix = ProductIndex(np.asarray(embeddings_matrix, dtype=np.float32), device=device)
    try:
        rows, scores = ix.dense_topk(np.asarray(query_vector, dtype=np.float32)[None, :], top_k)
    finally:
        ix.close()
    return rows[0], scores[0]


def cli_rows(frame: pd.DataFrame, snips: Optional[Dict] = None) -> List[Dict]:
    """The CLI's JSON row schema (app/test.py:312-328), 4-dp rounding; ``snips`` is run_search's
    second return value (sku -> best review)."""
    rows = []
    for _, r in frame.iterrows():
        n = r.get("n_reviews", np.nan)
        a = r.get("avg_stars", np.nan)
        snip = snips.get(str(r["sku"])) if snips else None
        rows.append({
            "sku": str(r["sku"]), "score": round(float(r["_final"]), 4),
            "dense": round(float(r["_dense"]), 4), "bm25": round(float(r["_bm25"]), 4),
            "rerank": round(float(r["_rerank"]), 4), "prior": round(float(r["_prior"]), 4),
            "bestrev": round(float(r["_best"]), 4),
            "n_reviews": int(n) if pd.notna(n) else 0,
            "avg_stars": round(float(a), 2) if pd.notna(a) else None,
            "snippet_stars": float(snip["stars"]) if snip and snip.get("stars") is not None else None,
            "snippet": snip["text"] if snip else None})
    return rows
