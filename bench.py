#!/usr/bin/env python3
"""Headline benchmark: queries/sec of the hybrid search path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): hybrid alpha=0.5 (w_dense=0.5, w_bm25=0.5, gate off), top-k=100,
pool=150, over a synthetic corpus of --docs products x 384-d fp32 unit vectors with a BM25
corpus of ~40 tokens per document (vocabulary 200k, Zipf 1.07); queries arrive in batches of
--batch.  The corpus is row-sharded across the N GPUs (strong scaling: the corpus is fixed, a
GPU holds docs/N rows); one RCCL all-gather per batch merges the per-shard candidates.

A "step" = one batch end to end (SURVEY 8d timing protocol): H2D of the query vectors and of the
BM25 token ids (pinned host buffers) -> K1 (dense scan + top-pool) -> K2 (BM25 at the pool) ->
[all-gather] -> K3 (fusion + top-k) -> D2H of rows / order / finals into pinned host memory.
The corpus itself is resident in HBM.  value = batch * steps / wall time of the K steps between two
barrier + synchronize fences, max over ranks; per-step HIP-event deltas give median / p10 / p90.

The JSON line also carries `roofline` (the dense scan kernel, timed with HIP events around every
launch on the stream it runs on) and, at N=1, `cpu_baseline` (the oracle = port of the
reference's numpy / rank_bm25-style CPU path, timed on the host cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 384
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
BF16_MFMA_PEAK_PF = 2.5      # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--doc-len", type=int, default=40)
    ap.add_argument("--no-bm25", action="store_true", help="dense-only (BASELINE configs[1])")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="matrix storage (bf16: BASELINE configs[3]; queries and arithmetic stay fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ce-precision", choices=["fp32", "bf16"], default="fp32",
                    help="arithmetic of the K5 cross-encoder with --rerank-k: fp32 = the reference's (logits within 1e-5 of "
                         "transformers), bf16 = the fast path (2.5e-2)")
    ap.add_argument("--rerank-k", type=int, default=0,
                    help="BASELINE config 5: cross-encoder rerank of the first RERANK_K pool rows per query (K5, seeded random "
                         "weights of the ms-marco-MiniLM-L-6 shape, synthetic product token ids built on the device); "
                         "use with --batch 64 --k 20")
    ap.add_argument("--force-payload", action="store_true",
                    help="diagnostic: with one rank, still pack / exchange / merge the shard payload")
    return ap.parse_args()


def make_queries(torch, args, stats, n_sets=4):
    """Host-side query sets: pinned (batch, 384) fp32 vectors + flattened BM25 token ids."""
    from review_recommender_amd import synth
    sets = []
    for i in range(n_sets):
        q = torch.from_numpy(synth.unit_rows(args.batch, DIM, 4321 + i)).pin_memory()
        terms = None
        if not args.no_bm25:
            lists = synth.query_terms(args.batch, args.vocab, 99 + i, stats["df"])
            off = np.zeros(args.batch + 1, dtype=np.int32)
            np.cumsum([len(t) for t in lists], out=off[1:])
            terms = (np.concatenate(lists).astype(np.int32), off)
            # realised postings per query ("record the realised sum of df(t) per query")
            stats["sum_df_per_query"] = float(np.mean([stats["df"][t].sum() for t in lists]))
        sets.append((q, terms))
    return sets


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(torch, args, shard, query_sets):
    """The reference's CPU path restated in oracle/ (`kind: "port"`), timed on this box's host cores on
    a bounded sample of the same workload (SURVEY 8d row "CPU baseline"):
      dense    numpy matvec + argpartition + argsort (utils.py:114-122) on <= 1M rows, BLAS threads = physical cores
               (median of 9 after 3 warm-up calls, min / max reported) and 1 thread;
      bm25     rank_bm25-style per-token Python loop over per-document dicts + the sku dict of
               app/app_product_search.py:206-208 on 200k documents; vectorised CSR numpy on <= 1M documents;
      pipeline the whole restated run_search at 10k products (BASELINE config 1) and at <= 1M.
    `value` = the reference's own path (numpy dense on all cores + Python-loop BM25 + sku dict), every
    part scaled linearly in documents to --docs.  Reported beside the GPU figure, never the target."""
    import pandas as pd
    from threadpoolctl import threadpool_limits
    from oracle.bm25 import BM25OkapiOracle
    from oracle.dense import cosine_similarity_search
    from oracle.pipeline import run_search_oracle
    from review_recommender_amd import synth

    n_s = min(shard.n_local, 1_000_000)
    V = shard.matrix[:n_s].float().cpu().numpy()
    qs = query_sets[0][0].numpy()
    scale = args.docs / n_s
    out = {}

    def timed(fn, reps, warm=1):
        """MEDIAN of `reps` calls after `warm` untimed ones (a mean over a few calls of a memory-bound sgemv on hundreds of
        BLAS threads moved 5x between runs of the same box class: VERDICT r2)."""
        for i in range(warm):
            fn(i)
        ts = []
        for i in range(reps):
            t0 = time.perf_counter()
            fn(warm + i)
            ts.append(time.perf_counter() - t0)
        timed.last = ts
        return float(np.median(ts))

    # BLAS threads = PHYSICAL cores: the matvec is memory-bound, SMT siblings only add contention
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except ImportError:
        phys = os.cpu_count()
    with threadpool_limits(limits=phys):
        t_dense_all = timed(lambda i: cosine_similarity_search(qs[i % len(qs)], V, 150), 9, warm=3)
    spread_all = [round(x * scale * 1e3, 2) for x in (min(timed.last), max(timed.last))]
    with threadpool_limits(limits=1):
        t_dense_1 = timed(lambda i: cosine_similarity_search(qs[i % len(qs)], V, 150), 3)
    out["dense_all_cores_ms"] = round(t_dense_all * scale * 1e3, 2)
    out["dense_all_cores_min_max_ms"] = spread_all
    out["dense_blas_threads"] = int(phys)
    out["dense_1_core_ms"] = round(t_dense_1 * scale * 1e3, 2)
    t_loop = t_csr = 0.0
    if not args.no_bm25:
        a = shard.bm25_arrays
        flat, off = query_sets[0][1]
        tl = [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]
        # (a) Python loop over per-document dicts, 200k documents (documents as token-id lists): x50 to 10M
        n_l = min(200_000, n_s)
        ip = a["doc_indptr"][:n_l + 1].cpu().numpy()
        dt = a["doc_terms"][:int(ip[-1])].cpu().numpy()
        df_ = a["doc_tf"][:int(ip[-1])].cpu().numpy()
        corpus = [np.repeat(dt[ip[d]:ip[d + 1]], df_[ip[d]:ip[d + 1]]).tolist() for d in range(n_l)]
        bm = BM25OkapiOracle(corpus)
        skus = synth.skus(n_l)

        def loop(i):
            scores = np.array(bm.get_scores(tl[i % len(tl)].tolist()), dtype=np.float32)
            by_sku = {skus[j]: scores[j] for j in range(n_l)}                # app/app_product_search.py:207
            return [by_sku.get(s, 0.0) for s in skus[:150]]
        t_loop = timed(loop, 3, warm=1) * (args.docs / n_l)
        out["bm25_python_loop_ms"] = round(t_loop * 1e3, 1)
        out["bm25_python_loop_docs"] = n_l
        # (b) vectorised CSR numpy over the sample's postings (sorted on the GPU: setup, not timed)
        csr = _csr_oracle(torch, a, n_s, args.vocab)
        rows150 = np.arange(150)
        t_csr = timed(lambda i: csr.get_scores(tl[i % len(tl)].tolist())[rows150].astype(np.float32), 5) * scale
        out["bm25_csr_numpy_ms"] = round(t_csr * 1e3, 2)

        # (c) the whole restated run_search (app flavour: sku dict over all documents) at 10k and at the sample
        class _Ids:
            def get_scores(self, toks):
                return csr.get_scores([int(t[1:]) for t in toks])
        n_rev, stars = synth.metadata(n_s, 2)
        all_skus = synth.skus(n_s)
        for n_p, reps in ((10_000, 5), (n_s, 2)):
            meta = pd.DataFrame({"sku": all_skus[:n_p], "n_reviews": n_rev[:n_p], "avg_stars": stars[:n_p],
                                 "agg_text": ""})
            small = _csr_oracle(torch, a, n_p, args.vocab) if n_p < n_s else csr

            class _B:
                def get_scores(self, toks, _c=small):
                    return _c.get_scores([int(t[1:]) for t in toks])

            def full(i, _m=meta, _n=n_p):
                q = " ".join(f"t{int(t)}" for t in tl[i % len(tl)])
                return run_search_oracle(query=q, qvec=qs[i % len(qs)], meta=_m, V=V[:_n], bm25=_B(),
                                         bm25_skus=all_skus[:_n], k=args.k, rerank_k=0, w_dense=0.5, w_bm25=0.5,
                                         w_rerank=0.0, w_prior=0.0, w_best=0.0, prior_C=20.0, min_reviews=8,
                                         gate_penalty=1.0)
            out[f"run_search_{n_p}_products_ms"] = round(timed(full, reps) * 1e3, 1)
    qps = 1.0 / (t_dense_all * scale + t_loop)
    return {"value": round(qps, 4), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "cpu_model": _cpu_model(), "blas_threads": int(phys), "timing": "median after warm-up",
            "sample": (f"reference-shaped path = numpy matvec+argpartition on {n_s} of {args.docs} rows ({phys} BLAS "
                       f"threads = physical cores, x{scale:.0f}) + rank_bm25-style Python scoring + sku dict on "
                       f"{out.get('bm25_python_loop_docs', 0)} docs (1 thread, scaled linearly); "
                       "variants_ms are per query at --docs documents except run_search_*_products_ms "
                       "(per query at that many products, unscaled)"),
            "variants_ms": out}


def _csr_oracle(torch, a, n_docs, vocab):
    """BM25CsrOracle over the first n_docs documents of the shard's forward arrays (postings = the forward
    entries sorted by (term, doc), on the GPU; corpus-wide idf / avgdl as the shard itself uses)."""
    from oracle.bm25 import BM25CsrOracle
    ip = a["doc_indptr"][:n_docs + 1]
    nnz = int(ip[-1].item())
    doc_of = torch.repeat_interleave(torch.arange(n_docs, device=ip.device), ip[1:] - ip[:-1])
    terms = a["doc_terms"][:nnz].long()
    key, perm = torch.sort(terms * n_docs + doc_of)
    post_docs = (key % n_docs).int().cpu().numpy()
    post_tf = a["doc_tf"][:nnz][perm].cpu().numpy()
    pip = np.zeros(vocab + 1, dtype=np.int64)
    np.cumsum(torch.bincount(terms, minlength=vocab).cpu().numpy(), out=pip[1:])
    return BM25CsrOracle(pip, post_docs, post_tf, a["doc_len"][:n_docs].cpu().numpy(), a["idf"].cpu().numpy(),
                         a["avgdl"])


class SyntheticPairScorer:
    """The rerank step of BASELINE config 5 with synthetic text: `rerank_fn(query_idx, rows) -> scores` for
    ShardedSearcher.search_batch_dev.  Every product row has a deterministic pseudo-random token sequence of 54..501 ids
    (the reference cuts texts at 2000 characters, ~400-500 WordPiece tokens), every query 8; the packed
    [CLS] q [SEP] text [SEP] sequences are built with torch ops on the device and scored by csrc/rr_ce.hip
    (seeded random weights: real ms-marco weights are not available offline).  The pool rows arrive on the host and the
    scores go back through it, as in the reference's flow (texts are host data)."""

    def __init__(self, torch, dev, seed=7, chunk_tokens=1 << 20, precision="fp32"):
        from review_recommender_amd import synth
        from review_recommender_amd.cross_encoder import CrossEncoder
        self.torch, self.dev, self.chunk = torch, dev, chunk_tokens
        self.ce = CrossEncoder(synth.bert_state_dict(seed, n_layers=6, n_labels=1), device=dev.index, precision=precision)
        self.pairs = 0
        self.tokens = 0
        self.ms = 0.0

    def __call__(self, qi, rows):
        torch = self.torch
        n = len(rows)
        out = torch.empty(n, dtype=torch.float32, device=self.dev)
        r = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(self.dev)
        qq = torch.from_numpy(np.ascontiguousarray(qi, dtype=np.int64)).to(self.dev)
        ltxt = 54 + (r * 2654435761 % 448)                         # text tokens per product
        length = ltxt + 11                                         # [CLS] 8 query tokens [SEP] text [SEP]
        per = max(1, self.chunk // 512)
        for s0 in range(0, n, per):
            ln = length[s0:s0 + per]
            m = ln.numel()
            pos = torch.arange(512, device=self.dev).expand(m, 512)
            mask = pos < ln[:, None]
            hq = (qq[s0:s0 + per, None] * 40503 + pos * 9973) % 29000 + 1000
            ht = (r[s0:s0 + per, None] * 7919 + pos * 104729) % 29000 + 1000
            tok = torch.where(pos == 0, 101, torch.where(pos <= 8, hq, torch.where((pos == 9) | (pos == ln[:, None] - 1), 102, ht)))
            typ = (pos > 9).int()
            cu = torch.zeros(m + 1, dtype=torch.int32, device=self.dev)
            cu[1:] = torch.cumsum(ln, 0).int()
            logits = self.ce.model.forward_packed_dev(tok[mask].int().contiguous(), typ[mask].contiguous(),
                                                      pos[mask].int().contiguous(), cu, m, 512, 0)
            out[s0:s0 + m] = logits[:, 0]
            self.tokens += int(cu[-1].item())
        self.pairs += n
        res = out.cpu().numpy()
        if not np.isfinite(res).all():                             # (NaN logits = the encoder's fp16-range flag, include/rr_hip.h)
            raise RuntimeError("cross-encoder logits are not finite: rr_ce_range_status / set_wide_range")
        return res


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the search path has no CPU fallback")
    # rehearsal switch (never set by the driver): RR_BENCH_REHEARSE=gloo runs the N > 1 code path with every rank on
    # cuda:0 and the gloo backend (RCCL refuses two ranks on one device), to exercise sharding, the setup collectives,
    # the payload exchange and the merge on a one-GPU box.  Timings of such a run mean nothing.
    rehearse = os.environ.get("RR_BENCH_REHEARSE", "")
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(rehearse)
        else:
            dist.init_process_group("nccl", device_id=dev)

    from review_recommender_amd.device_corpus import build_device_shard
    from review_recommender_amd.engine import FusionWeights
    shard = build_device_shard(torch, dist, docs=args.docs, rank=rank, world=world, dev=dev, vocab=args.vocab,
                               doc_len=args.doc_len, want_bm25=not args.no_bm25, dtype=args.dtype,
                               force_payload=args.force_payload)
    sharded, index, stats, n_local = shard.sharded, shard.index, shard.stats, shard.n_local
    qsets = make_queries(torch, args, stats)
    w = FusionWeights(w_dense=0.5, w_bm25=0.0 if args.no_bm25 else 0.5, w_rerank=0.0, w_prior=0.0,
                      w_best=0.0, gate_penalty=1.0)
    pool = max(args.k, 150)
    q_dev = [torch.empty((args.batch, DIM), dtype=torch.float32, device=dev) for _ in range(2)]
    pin_rows = torch.empty((args.batch, pool), dtype=torch.int64).pin_memory()
    pin_order = torch.empty((args.batch, args.k), dtype=torch.int32).pin_memory()
    pin_final = torch.empty((args.batch, pool), dtype=torch.float64).pin_memory()

    rerank_fn = None
    if args.rerank_k > 0:
        rerank_fn = SyntheticPairScorer(torch, dev, seed=7, precision=args.ce_precision)
        w = FusionWeights(w_dense=0.4, w_bm25=0.2, w_rerank=0.3, w_prior=0.1, w_best=0.0, gate_penalty=1.0)
        pool = max(args.k, args.rerank_k, 150)
        pin_rows = torch.empty((args.batch, pool), dtype=torch.int64).pin_memory()
        pin_final = torch.empty((args.batch, pool), dtype=torch.float64).pin_memory()

    # RR_BENCH_STREAMS=3 (A/B): batch i + 1's uploads on the searcher's input stream and batch i - 1's downloads on
    # out_stream, under batch i's kernels.  Measured SLOWER than one stream on this stack (r02: 1.25M-row shard 0.79
    # vs 0.69 ms per step, 10M rows 2.75 vs 2.72 ms: the cross-stream events cost more than the ~90 us of copies they
    # hide), so the default keeps every copy on the compute stream, in order.
    one_stream = os.environ.get("RR_BENCH_STREAMS") != "3"
    kernel_copies = os.environ.get("RR_NO_KERNEL_COPY") is None      # (A/B: copy commands instead)
    out_stream = torch.cuda.Stream(device=dev)
    pins = [(pin_rows, pin_order, pin_final),
            (torch.empty_like(pin_rows).pin_memory(), torch.empty_like(pin_order).pin_memory(),
             torch.empty_like(pin_final).pin_memory())]
    last_pins = [pins[0]]
    keep = []

    # Row shards (world > 1, or --force-payload): batch i's payload all-gather is started and batch i + 1's K1 enqueued
    # before batch i's merge (ShardedSearcher.submit / finish: SURVEY 8e's "all-gather of batch i under K1 of batch i + 1").
    # Every batch is still submitted AND finished inside the timed region (drain() in front of the closing fence).
    pipelined = ((world > 1 or args.force_payload) and args.rerank_k == 0 and one_stream and kernel_copies
                 and os.environ.get("RR_BENCH_NO_PIPELINE") is None)
    # Row shards (world > 1; RR_TAIL_OVERLAP_CUS=n forces it with one rank, RR_NO_TAIL_OVERLAP=1 turns it off): submit / finish
    # are a three-stage pipeline on two CU-masked streams -- scan(i + 1) | selection + K2 + payload all-gather(i) | merge +
    # K3(i - 1) -- and two batches stay in flight behind the one being submitted (profiles/r04_overlap_ab.md: a rank's step of
    # eight shards 0.433 -> 0.350 ms on the one-GPU proxy; one GPU at 10M rows: no gain, hence off there).
    overlap = pipelined and sharded.enable_overlap()
    depth = 2 if overlap else 1
    inflight = []
    if overlap:
        # the masked streams are ordinary (blocking) HIP streams: every command on the NULL stream -- torch's default stream --
        # would wait for both of them and hold both back.  The caller's side of the loop (answer copies, events) therefore
        # runs on a non-blocking stream of its own.
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    def drain(keep=0):
        while len(inflight) > keep:
            rows, cols, order = sharded.finish(inflight.pop(0))
            p_rows, p_order, p_final = pins[0]
            sharded.s.copy_segments([(p_rows, rows), (p_order, order), (p_final, cols[:, 7, :])])

    def step(i):
        q_pin, terms = qsets[i % len(qsets)]
        cur = torch.cuda.current_stream(dev)
        if pipelined:
            t = sharded.submit(q_pin, terms, args.k, w)
            drain(depth - 1)
            inflight.append(t)
            return
        if one_stream and kernel_copies:
            # H2D: K1's first kernel reads the pinned query vectors over PCIe itself; the token ids go through the
            # searcher's pinned staging ring and one copy kernel (inside search); D2H: ONE kernel writes rows / order /
            # final scores into the pinned answer buffers (rr_copy_segments_dev) -- the same bytes cross PCIe inside the
            # step, without the fixed latency of five copy commands
            rows, cols, order = sharded.search_batch_dev(q_pin, terms, args.k, w, rerank_k=args.rerank_k, rerank_fn=rerank_fn)
            p_rows, p_order, p_final = pins[0]
            sharded.s.copy_segments([(p_rows, rows), (p_order, order), (p_final, cols[:, 7, :])])
            return
        if one_stream:
            q = q_dev[i % 2]
            q.copy_(q_pin, non_blocking=True)                    # H2D: query vectors (token ids: inside search)
            rows, cols, order = sharded.search_batch_dev(q, terms, args.k, w, rerank_k=args.rerank_k, rerank_fn=rerank_fn)
            p_rows, p_order, p_final = pins[0]
            p_rows.copy_(rows, non_blocking=True)                # D2H: the answer
            p_order.copy_(order, non_blocking=True)
            p_final.copy_(cols[:, 7, :], non_blocking=True)
            return
        batch = sharded.s.stage_batch(q_pin, terms)              # H2D: query vectors + token ids, on the input stream
        cur.wait_event(batch.ready)
        rows, cols, order = sharded.search_batch_dev(batch.q, batch.terms, args.k, w, rerank_k=args.rerank_k,
                                                     rerank_fn=rerank_fn)
        sharded.s.release(batch)
        done = torch.cuda.Event()
        done.record(cur)
        p_rows, p_order, p_final = last_pins[0] = pins[i % 2]
        with torch.cuda.stream(out_stream):                      # D2H: the answer
            out_stream.wait_event(done)
            p_rows.copy_(rows, non_blocking=True)
            p_order.copy_(order, non_blocking=True)
            p_final.copy_(cols[:, 7, :], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(out_stream)
        # the answer's device tensors stay referenced until their download has run (no record_stream: the caching
        # allocator would then hold every block back behind events and go to hipMalloc)
        keep.append((copied, rows, cols, order))
        if len(keep) > 3:
            keep.pop(0)[0].synchronize()

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not measurement: the first batched searches build the index's lazy parts (row-norm bound, bf16 filter
    # plane, scratch), size the pinned staging ring and let torch's caching allocator reach its steady state (a
    # one-off 74 ms allocator stall was seen as late as the 14th call of a 1.25M-row run)
    for i in range(16):
        step(i)
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    _scan_stats(index)                         # drain warm-up launches
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # (no Python garbage collection inside the timed region: a generation-2 pass over the setup's objects was seen as one
    #  75 ms step in a 0.45 ms-per-step run)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    marks[0].record()
    host_s = 0.0
    prof = None
    if os.environ.get("RR_BENCH_HOST_PROFILE"):      # where the host's share of a step goes (cProfile of the timed loop)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    for i in range(args.steps):
        h0 = time.perf_counter()
        step(i)
        host_s += time.perf_counter() - h0       # the host's share: enqueueing one step (no device wait inside)
        marks[i + 1].record()
    if prof is not None:
        prof.disable()
        import pstats
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(28)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    total_ms, launches = _scan_stats(index)
    info = _scan_info(index)
    # the host's cost of ENQUEUEING a step, measured where nothing can block it: six steps on an idle stream (inside the
    # depth of the pinned staging ring, so no slot is waited for), timed on the host alone.  (host_loop_ms_per_step below is
    # the timed loop's own host time: it includes the ring's back-pressure once the GPU is the bottleneck, i.e. GPU time.)
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    for i in range(6):
        step(i)
    host_enqueue_ms = (time.perf_counter() - h0) / 6 * 1e3
    fence()
    _scan_stats(index)
    # the answer of the last step, as it sits in the pinned buffers: cheap invariants (parity itself is the tests' job:
    # tests/test_gpu_bench_config.py checks this very configuration against the oracle)
    r_h, o_h, f_h = last_pins[0][0].numpy(), last_pins[0][1].numpy().astype(np.int64), last_pins[0][2].numpy()
    top = np.take_along_axis(f_h, o_h, axis=1)
    ok = (r_h.min() >= 0 and r_h.max() < args.docs and np.all(np.diff(top, axis=1) <= 0)
          and all(len(set(o.tolist())) == o_h.shape[1] for o in o_h[:8]) and np.isfinite(top).all())
    if not ok:
        raise SystemExit("bench.py: the last step's answer violates its invariants (rows in range, top-k sorted, distinct)")
    per_step = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    if os.environ.get("RR_BENCH_DEBUG"):
        print("per-step ms:", np.round(per_step, 3).tolist(), file=sys.stderr, flush=True)

    # the same filter scan with ONE query set per launch (128 queries per pass over the stream): the HBM-bound form of
    # the kernel -- the two-set launch above reads the stream once for 256 queries and is paced by the matrix side
    one_set = None
    q_dev[0].copy_(qsets[0][0])                  # (the side measurements below search device-resident queries)
    if args.batch > 128 and info["queries_per_launch"] > 128:
        q128 = q_dev[0][:128].contiguous()
        for _ in range(2):
            sharded.s.dense_pool(q128, pool)
        torch.cuda.synchronize()
        _scan_stats(index)
        for _ in range(10):
            sharded.s.dense_pool(q128, pool)
        torch.cuda.synchronize()
        ms128, n128 = _scan_stats(index)
        i128 = _scan_info(index)
        b128 = n_local * DIM * i128["elem_bytes"]
        gbs128 = b128 / (ms128 / max(n128, 1) * 1e-3) / 1e9
        one_set = {"bound": "hbm", "achieved": round(gbs128, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(gbs128 / HBM_PEAK_GBS, 4), "kernel": i128["kernel"], "launches": int(n128),
                   "avg_launch_ms": round(ms128 / max(n128, 1), 5), "bytes_per_launch": b128,
                   "queries_per_launch": i128["queries_per_launch"]}

    # the HBM-bound end of the path: one query per matrix read (rr_scan_f32<6,1>), same shard
    q1 = q_dev[0][:1].contiguous()
    for _ in range(2):
        sharded.s.dense_pool(q1, pool)
    torch.cuda.synchronize()
    _scan_stats(index)
    for _ in range(10):
        sharded.s.dense_pool(q1, pool)
    torch.cuda.synchronize()
    ms1, n1 = _scan_stats(index)
    esz = 2 if args.dtype == "bf16" else 4
    bytes_per_launch = n_local * DIM * esz            # algorithmic: the shard's matrix, read once per launch
    gbs1 = bytes_per_launch / (ms1 / max(n1, 1) * 1e-3) / 1e9
    single = {"bound": "hbm", "achieved": round(gbs1, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(gbs1 / HBM_PEAK_GBS, 4),
              "kernel": "rr_scan_bf16<1>" if args.dtype == "bf16" else "rr_scan_f32<6,1>", "launches": int(n1),
              "avg_launch_ms": round(ms1 / max(n1, 1), 5), "bytes_per_launch": bytes_per_launch}

    if rank == 0:
        avg_ms = total_ms / max(launches, 1)
        bytes_per_launch = n_local * DIM * info["elem_bytes"]      # what the batched scan streams once per launch
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        qpl, terms_per_dim = info["queries_per_launch"], info["mfma_terms"]
        # HBM traffic per launch: NOT measured in this run -- replayed from the PMC passes kept in profiles/ (FETCH_SIZE /
        # WRITE_SIZE collected and corrected as MI355X_MICROARCH.md prescribes, tools/profile_pmc.sh) for this kernel at this
        # many queries per launch on 10M rows, the same bytes-per-row ratio applied to this run's rows.  The line says so:
        # `traffic` stays null (nothing was counted here), `traffic_from_profile` carries the figure and its file.
        traffic_from_profile = None
        # (r03: the 256-query rr_scan_fltq on 16x16x32 MFMAs; r02: the 128-query scan over the bf16 filter plane with the store
        #  prefilter; r01: the same scan over fp32 rows -- the newest summary of this launch shape wins)
        for tag in (("r04", "r03", "r02") if info["elem_bytes"] == 2 else ("r01",)):
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_scan_b{qpl}_10M.json")
            if os.path.exists(pmc) and args.dtype == "f32":
                with open(pmc) as f:
                    m = json.load(f)
                traffic_from_profile = {"bytes": int(m["hbm_bytes_per_launch"] / m["algorithmic_bytes_per_launch"] * bytes_per_launch),
                                        "ratio_to_algorithmic": round(m["hbm_bytes_per_launch"] / m["algorithmic_bytes_per_launch"], 4),
                                        "file": os.path.relpath(pmc, ROOT)}
                break
        pf = terms_per_dim * 2.0 * n_local * DIM * qpl / (avg_ms * 1e-3) / 1e15 if launches else 0.0
        hbm_frac, mfma_frac = achieved / HBM_PEAK_GBS, pf / BF16_MFMA_PEAK_PF
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(hbm_frac, 4), "traffic": None, "traffic_from_profile": traffic_from_profile, "kernel": info["kernel"],
                "launches": int(launches), "avg_launch_ms": round(avg_ms, 5), "bytes_per_launch": bytes_per_launch,
                "queries_per_launch": qpl, "launches_per_step": round(launches / max(args.steps, 1), 3)}
        if terms_per_dim:
            roof.update({"matrix_core_pflops": round(pf, 4), "matrix_core_frac_of_2.5PF": round(mfma_frac, 4)})
            if mfma_frac > hbm_frac:      # both fractions are printed; the larger one names the bound
                roof.update({"bound": "mfma", "achieved": round(pf * 1e3, 2), "peak": BF16_MFMA_PEAK_PF * 1e3,
                             "unit": "TFLOP/s", "frac": round(mfma_frac, 4), "hbm_read_gbs": round(achieved, 2)})
        out = {
            "metric": "queries/sec at top-k=100 (hybrid alpha=0.5)" if not args.no_bm25
                      else "queries/sec at top-k=100 (dense only)",
            "value": round(args.batch * args.steps / dt, 3), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),
            "host_loop_ms_per_step": round(host_s / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "storage": args.dtype,
            "ms_per_step_median": round(float(np.median(per_step)), 4),
            "ms_per_step_p10": round(float(np.percentile(per_step, 10)), 4),
            "ms_per_step_p90": round(float(np.percentile(per_step, 90)), 4),
            "ms_per_step_max": round(float(per_step.max()), 4),
            "timed_region": "H2D queries + token ids -> K1 -> K2 -> [all-gather] -> K3 -> D2H rows/order/finals" + ("" if one_stream else " (uploads / downloads of neighbouring batches on their own streams)"),
            "scan_ms_per_step": round(total_ms / max(args.steps, 1), 4),
            "config": {"workload": (f"{'hybrid BM25+dense alpha=0.5' if not args.no_bm25 else 'dense-only cosine'} "
                                    f"top-k={args.k} pool={pool}, {args.docs} products x {DIM} {'bf16-stored' if args.dtype == 'bf16' else 'fp32'}"
                                    + (f", BM25 ~{args.doc_len} tokens/doc vocab {args.vocab} "
                                       f"({stats['nnz']} postings on rank 0, {stats.get('sum_df_per_query', 0):.0f} "
                                       f"postings per query)" if not args.no_bm25 else "")
                                    + f", batches of {args.batch} queries"),
                       # four seeded query sets are cycled (step i searches set i % 4): the library keeps nothing between calls
                       # (tests: "the same batch again gives the same bits", token lists staged on every call)
                       "query_sets": len(qsets),
                       "docs": args.docs, "docs_per_gpu": n_local, "batch": args.batch, "k": args.k,
                       "pool": pool, "parallelism": ("single GPU: no collective" if world == 1 else
                                                       f"row-shard x{world} + 1 all-reduce(min, B floats) + 1 all-gather"
                                                       + (" started under the next batch's K1" if pipelined else "")
                                                       + (f"; every batch's tail on {torch.cuda.get_device_properties(dev).multi_processor_count - sharded._ov.scan_cus} CUs beside the next batch's scan on {sharded._ov.scan_cus}" if overlap else ""))},
            "roofline": roof,
            # the single-query scan on the same shard: the HBM-bound end of the same path
            "roofline_single_query": single,
            **({"roofline_one_query_set": one_set} if one_set else {}),
        }
        if rerank_fn is not None:
            out["metric"] = f"queries/sec, hybrid + cross-encoder rerank top-{args.rerank_k} -> top-{args.k}"
            out["config"]["workload"] += (f"; rerank of the first {args.rerank_k} pool rows per query by the K5 cross-encoder "
                                          "(MiniLM-L6 shape, seeded weights, synthetic token ids 65..512 per pair)")
            out["rerank"] = {"precision": args.ce_precision, "pairs_per_step": args.batch * args.rerank_k,
                             "mean_tokens_per_pair": round(rerank_fn.tokens / max(rerank_fn.pairs, 1), 1)}
        if world == 1 and not args.no_cpu_baseline and rerank_fn is None:
            out["cpu_baseline"] = cpu_baseline(torch, args, shard, qsets)
        print(json.dumps(out), flush=True)
    if overlap:
        sharded.disable_overlap()          # (the CU-masked streams are destroyed before the process ends)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _scan_stats(index):
    from review_recommender_amd import _lib
    tot, n = C.c_double(), C.c_int64()
    _lib.check(_lib.load().rr_index_scan_stats(index.handle, C.byref(tot), C.byref(n)), "rr_index_scan_stats")
    return tot.value, n.value


_KERNELS = {1: "rr_scan_f32<6,NB>", 2: "rr_scan_bf16<NB>", 3: "rr_scan_mfma_x3<1>", 4: "rr_scan_x3w<NQT>",
            5: "rr_scan_flt<NQ2>", 6: "rr_scan_mfma_f32<NQT>", 7: "rr_scan_mfma_bf16"}


def _scan_info(index):
    """Which scan kernel the last batched launch used (rr_index_last_scan_info)."""
    from review_recommender_amd import _lib
    out = (C.c_int32 * 8)()
    _lib.check(_lib.load().rr_index_last_scan_info(index.handle, out), "rr_index_last_scan_info")
    kid, variant, qpl, terms, ebytes = out[0], out[1], out[2], out[3], out[4]
    name = _KERNELS.get(kid, "unknown").replace("NQ2", str(variant)).replace("NQT", str(variant)).replace("NB", str(variant))
    if kid == 5 and ebytes == 2:          # a bf16 stream is scanned on 16x16x32 tiles; variant 8 = one launch for two 128-query sets
        name = ("rr_scan_fltq (two query sets, query-stationary)" if variant == 9 else
                "rr_scan_flt16<4, two query sets>" if variant == 8 else f"rr_scan_flt16<{variant}>")
    stream = "bf16 rows" if index.dtype == "bf16" else ("bf16 filter plane of the fp32 rows" if ebytes == 2 else "fp32 rows")
    return {"kernel": f"{name} over {stream}", "queries_per_launch": int(qpl), "mfma_terms": int(terms),
            "elem_bytes": int(ebytes) or (2 if index.dtype == "bf16" else 4)}


if __name__ == "__main__":
    main()
