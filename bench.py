#!/usr/bin/env python3
"""Headline benchmark: queries/sec of the hybrid search path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): hybrid alpha=0.5 (w_dense=0.5, w_bm25=0.5, gate off), top-k=100,
pool=150, over a synthetic corpus of --docs products x 384-d fp32 unit vectors with a BM25
corpus of ~40 tokens per document (vocabulary 200k, Zipf 1.07); queries arrive in batches of
--batch.  The corpus is row-sharded across the N GPUs (strong scaling: the corpus is fixed, a
GPU holds docs/N rows); one RCCL all-gather per batch merges the per-shard candidates.
A "step" = one batch through K1 (dense scan + top-pool) -> K2 (BM25 at the pool) ->
[all-gather] -> K3 (fusion + top-k), queries resident in HBM, top-k copied back to pinned host
memory asynchronously.  value = batch * steps / time, max time over ranks.

The JSON line also carries `roofline` (the dense scan kernel, timed with HIP events around every
launch on the stream it runs on) and, at N=1, `cpu_baseline` (the oracle = port of the
reference's numpy / rank_bm25-style CPU path, timed on the host cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 384
SEED = 1234
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
F32_MATRIX_PEAK_TF = 157.3   # dense f32-input MFMA peak = f32 vector peak (MI355X_MICROARCH.md)
N_BLOCKS = 8                 # the corpus is generated in 8 seeded blocks so any N in {1,2,4,8} sees the same data


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--doc-len", type=int, default=40)
    ap.add_argument("--no-bm25", action="store_true", help="dense-only (BASELINE configs[1])")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="matrix storage (bf16: BASELINE configs[3]; queries and arithmetic stay fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-payload", action="store_true",
                    help="diagnostic: with one rank, still pack / exchange / merge the shard payload")
    return ap.parse_args()


def zipf_cdf(vocab, s=1.07):
    w = 1.0 / np.power(np.arange(1, vocab + 1, dtype=np.float64), s)
    return np.cumsum(w / w.sum())


def gen_block(torch, dev, block, rows, vocab, doc_len, cdf_dev, want_bm25):
    """One seeded corpus block on the device: unit rows, metadata, forward BM25 entries."""
    g = torch.Generator(device=dev)
    g.manual_seed(SEED * 1000 + block)
    x = torch.randn((rows, DIM), generator=g, device=dev, dtype=torch.float32)
    n_rev = torch.clamp(torch.floor(torch.exp(torch.randn(rows, generator=g, device=dev, dtype=torch.float64)
                                              * 1.2 + 2.5)), 1, 5000)
    stars = torch.round(torch.clamp(torch.randn(rows, generator=g, device=dev, dtype=torch.float64) * 0.6 + 4.1,
                                    1.0, 5.0) * 1000) / 1000
    out = {"x": x, "n": n_rev, "stars": stars}
    if want_bm25:
        dl = torch.clamp(torch.poisson(torch.full((rows,), float(doc_len), device=dev), generator=g), min=1).long()
        total = int(dl.sum().item())
        u = torch.rand(total, generator=g, device=dev, dtype=torch.float64)
        tok = torch.clamp(torch.searchsorted(cdf_dev, u), max=vocab - 1)
        doc = torch.repeat_interleave(torch.arange(rows, device=dev), dl)
        key, _ = torch.sort(doc * vocab + tok)
        uniq, cnt = torch.unique_consecutive(key, return_counts=True)
        out.update(doc_len=dl.int(), e_doc=(uniq // vocab), e_term=(uniq % vocab).int(), e_tf=cnt.int())
    return out


def build_shard(torch, dist, args, rank, world, dev):
    from review_recommender_amd import _lib
    from review_recommender_amd.bm25 import idf_with_floor
    from review_recommender_amd.engine import HybridSearcher
    from review_recommender_amd.index import ProductIndex
    from review_recommender_amd.sharded import ShardedSearcher, shard_bounds

    assert args.docs % N_BLOCKS == 0 and N_BLOCKS % world == 0, "docs %% 8 == 0 and gpus in {1,2,4,8}"
    per_block = args.docs // N_BLOCKS
    blocks = range(rank * N_BLOCKS // world, (rank + 1) * N_BLOCKS // world)
    lo, hi = shard_bounds(args.docs, world, rank)
    n_local = hi - lo
    want_bm25 = not args.no_bm25
    cdf_dev = torch.from_numpy(zipf_cdf(args.vocab)).to(dev) if want_bm25 else None

    mat = torch.empty((n_local, DIM), device=dev, dtype=torch.float32)
    n_rev = torch.empty(n_local, device=dev, dtype=torch.float64)
    stars = torch.empty(n_local, device=dev, dtype=torch.float64)
    parts = []
    for j, b in enumerate(blocks):
        blk = gen_block(torch, dev, b, per_block, args.vocab, args.doc_len, cdf_dev, want_bm25)
        s = j * per_block
        mat[s:s + per_block] = blk["x"]
        n_rev[s:s + per_block] = blk["n"]
        stars[s:s + per_block] = blk["stars"]
        if want_bm25:
            parts.append((blk["doc_len"], blk["e_doc"] + s, blk["e_term"], blk["e_tf"]))
        del blk
    if args.dtype == "bf16":
        # normalise in fp32, round once to bf16 (nearest even): SURVEY section 8d
        mat = (mat / torch.clamp(mat.norm(dim=1, keepdim=True), min=1e-12)).to(torch.bfloat16).contiguous()
        index = ProductIndex(None, n_rows=n_local, dim=DIM, device=dev.index, row_offset=lo,
                             device_ptr=mat.data_ptr(), keepalive=mat, dtype="bf16")
    else:
        index = ProductIndex(None, n_rows=n_local, dim=DIM, device=dev.index, row_offset=lo,
                             device_ptr=mat.data_ptr(), keepalive=mat)
        index.l2_normalize()                          # utils.py:40-44 on the device
    index.set_meta(n_rev.cpu().numpy(), stars.cpu().numpy())

    bm25 = None
    keep = [mat]
    if want_bm25:
        doc_len = torch.cat([p[0] for p in parts])
        e_doc = torch.cat([p[1] for p in parts])
        e_term = torch.cat([p[2] for p in parts])
        e_tf = torch.cat([p[3] for p in parts])
        del parts
        nnz = int(e_doc.numel())
        doc_indptr = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
        doc_indptr[1:] = torch.cumsum(torch.bincount(e_doc, minlength=n_local), 0)
        # postings: entries sorted by (term, doc)
        pkey, perm = torch.sort(e_term.long() * n_local + e_doc)
        post_docs = (pkey % n_local).int()
        post_tf = e_tf[perm].contiguous()
        df_local = torch.bincount(e_term.long(), minlength=args.vocab)
        post_indptr = torch.zeros(args.vocab + 1, dtype=torch.int64, device=dev)
        post_indptr[1:] = torch.cumsum(df_local, 0)
        del pkey, perm
        # corpus-wide statistics (setup-time collectives; not on the query path)
        df = df_local.clone()
        tot_len = doc_len.long().sum().reshape(1)
        if world > 1:
            dist.all_reduce(df)
            dist.all_reduce(tot_len)
        avgdl = int(tot_len.item()) / args.docs
        idf = torch.from_numpy(idf_with_floor(df.cpu().numpy(), args.docs)).to(dev)
        e_term32, e_doc = e_term.contiguous(), None
        h = C.c_void_p()
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_lib.load().rr_bm25_create_dev(
            dev.index, n_local, args.vocab, nnz, p(post_indptr), p(post_docs), p(post_tf), p(doc_indptr),
            p(e_term32), p(e_tf), p(doc_len), p(idf), avgdl, 1.5, 0.75, lo, C.byref(h)), "rr_bm25_create_dev")

        class _Dev:                                   # minimal BM25Index look-alike over adopted arrays
            handle = h
        bm25 = _Dev()
        keep += [post_indptr, post_docs, post_tf, doc_indptr, e_term32, e_tf, doc_len, idf]
        stats = dict(nnz=nnz, avgdl=avgdl, df=df.cpu().numpy())
    else:
        stats = dict(nnz=0, avgdl=0.0, df=None)
    searcher = HybridSearcher(index, bm25)
    sharded = ShardedSearcher(searcher, args.docs, rank, world)
    sharded.force_payload = args.force_payload
    return sharded, index, keep, stats, n_local


def make_queries(torch, dev, args, stats, n_sets=4):
    from review_recommender_amd import synth
    sets = []
    for i in range(n_sets):
        q = torch.from_numpy(synth.unit_rows(args.batch, DIM, 4321 + i)).to(dev)
        if args.no_bm25:
            terms = None
        else:
            terms = synth.query_terms(args.batch, args.vocab, 99 + i, stats["df"])
        sets.append((q, terms))
    return sets


def cpu_baseline(args, index_matrix_sample, stats, query_sets):
    """The reference's CPU path (its numpy cosine search + rank_bm25-style scoring + the sku dict
    of app/app_product_search.py:206-208), restated in oracle/, timed on this box's host cores on
    a bounded sample and scaled linearly in the number of documents.  Reported, never the target."""
    from oracle.bm25 import BM25OkapiOracle
    from oracle.dense import cosine_similarity_search
    V = index_matrix_sample
    qs = query_sets[0][0].cpu().numpy()
    cosine_similarity_search(qs[0], V, 150)                      # warm BLAS threads
    t0 = time.perf_counter()
    reps = 5
    for i in range(reps):
        cosine_similarity_search(qs[i % len(qs)], V, 150)
    t_dense = (time.perf_counter() - t0) / reps * (args.docs / V.shape[0])
    t_bm25, bm_docs = 0.0, 0
    if not args.no_bm25:
        bm_docs = 20_000
        rng = np.random.default_rng(7)
        cdf = zipf_cdf(args.vocab)
        corpus = [np.minimum(np.searchsorted(cdf, rng.random(max(1, rng.poisson(args.doc_len)))),
                             args.vocab - 1).tolist() for _ in range(bm_docs)]
        bm = BM25OkapiOracle(corpus)
        skus = [f"B{i:09d}" for i in range(bm_docs)]
        terms = query_sets[0][1]
        t0 = time.perf_counter()
        reps_b = 3
        for i in range(reps_b):
            scores = np.array(bm.get_scores(terms[i].tolist()), dtype=np.float32)
            by_sku = {skus[j]: scores[j] for j in range(bm_docs)}    # app/app_product_search.py:207
            _ = [by_sku.get(s, 0.0) for s in skus[:150]]
        t_bm25 = (time.perf_counter() - t0) / reps_b * (args.docs / bm_docs)
    qps = 1.0 / (t_dense + t_bm25)
    return {"value": round(qps, 4), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "sample": (f"dense: numpy matvec+argpartition on {V.shape[0]} of {args.docs} rows x5 queries "
                       f"(all BLAS threads); bm25: rank_bm25-style Python scoring + sku dict on {bm_docs} docs "
                       f"x3 queries (1 thread); both scaled linearly to {args.docs} docs; "
                       f"dense {t_dense * 1e3:.1f} ms + bm25 {t_bm25 * 1e3:.1f} ms per query")}


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
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from review_recommender_amd.engine import FusionWeights
    sharded, index, keep, stats, n_local = build_shard(torch, dist, args, rank, world, dev)
    qsets = make_queries(torch, dev, args, stats)
    w = FusionWeights(w_dense=0.5, w_bm25=0.0 if args.no_bm25 else 0.5, w_rerank=0.0, w_prior=0.0,
                      w_best=0.0, gate_penalty=1.0)
    pool = max(args.k, 150)
    pin_rows = torch.empty((args.batch, pool), dtype=torch.int64).pin_memory()
    pin_order = torch.empty((args.batch, args.k), dtype=torch.int32).pin_memory()
    pin_final = torch.empty((args.batch, pool), dtype=torch.float64).pin_memory()

    def step(i):
        q, terms = qsets[i % len(qsets)]
        rows, cols, order = sharded.search_batch_dev(q, terms, args.k, w)
        pin_rows.copy_(rows, non_blocking=True)
        pin_order.copy_(order, non_blocking=True)
        pin_final.copy_(cols[:, 7, :], non_blocking=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    index_scan = _scan_stats(index)           # drain warm-up launches
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    total_ms, launches = _scan_stats(index)

    # the HBM-bound end of the path: one query per matrix read (rr_scan_f32<6,1>), same shard
    q1 = qsets[0][0][:1].contiguous()
    for _ in range(2):
        sharded.s.dense_pool(q1, pool)
    torch.cuda.synchronize()
    _scan_stats(index)
    for _ in range(10):
        sharded.s.dense_pool(q1, pool)
    torch.cuda.synchronize()
    ms1, n1 = _scan_stats(index)
    gbs1 = n_local * DIM * (2 if args.dtype == "bf16" else 4) / (ms1 / max(n1, 1) * 1e-3) / 1e9
    single = {"bound": "hbm", "achieved": round(gbs1, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(gbs1 / HBM_PEAK_GBS, 4), "kernel": "rr_scan_bf16<1>" if args.dtype == "bf16" else "rr_scan_f32<6,1>", "launches": int(n1),
              "avg_launch_ms": round(ms1 / max(n1, 1), 5),
              "bytes_per_launch": n_local * DIM * (2 if args.dtype == "bf16" else 4)}

    if rank == 0:
        # HBM traffic per launch from the PMC passes kept in profiles/ (FETCH_SIZE x2 on gfx950 +
        # WRITE_SIZE, MI355X_MICROARCH.md), measured on this kernel at this many queries per launch
        # on 10M rows; the same bytes-per-row ratio is applied to this run's rows.  null when no
        # PMC summary for this launch shape is present.
        exact = bool(os.environ.get("RR_SCAN_EXACT"))
        qpl = min(args.batch, 64 if exact else 128) if args.batch > 4 else args.batch   # queries sharing one matrix read
        traffic = None
        chain_tag = ("chain_" if os.environ.get("RR_SCAN_F32_CHAIN") else "exact_" if exact else "") if qpl > 4 else ""
        pmc = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_scan_{chain_tag}b{qpl}_10M.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                m = json.load(f)
            if args.dtype == "f32":
                traffic = int(m["hbm_bytes_per_launch"] / m["algorithmic_bytes_per_launch"] * n_local * DIM * 4)
        esz = 2 if args.dtype == "bf16" else 4
        bytes_per_launch = n_local * DIM * esz    # algorithmic: the shard's matrix, read once per launch
        avg_ms = total_ms / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        chain = bool(os.environ.get("RR_SCAN_F32_CHAIN"))
        if qpl > 4 and not chain:
            # 5..128 queries per read: bf16 filter scan on the matrix cores (rr_dense_flt.hip: one MFMA term per
            # dim, candidates rescored exactly) -- or, with RR_SCAN_EXACT=1, the split-operand scans (6 terms for
            # fp32 storage, 3 for bf16; rr_dense_x3.hip up to 16 queries, rr_dense_x3w.hip up to 64).  Both the HBM
            # fraction and the matrix-core fraction are reported; the larger one names the bound.
            terms = 1 if not exact else (3 if args.dtype == "bf16" else 6)
            pf = terms * 2.0 * n_local * DIM * qpl / (avg_ms * 1e-3) / 1e15 if launches else 0.0
            hbm_frac, mfma_frac = achieved / HBM_PEAK_GBS, pf / 2.5
            roof = {"bound": "hbm" if hbm_frac >= mfma_frac else "mfma", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4), "traffic": traffic,
                    "kernel": (f"rr_scan_flt<{1 if qpl <= 32 else 2 if qpl <= 64 else 4},{args.dtype}>" if not exact else
                               f"rr_scan_mfma_x3<1,{args.dtype}>" if qpl <= 16 else
                               f"rr_scan_x3w<{1 if qpl <= 32 else 2},{args.dtype}>"),
                    "matrix_core_pflops": round(pf, 4), "matrix_core_frac_of_2.5PF": round(mfma_frac, 4)}
            if mfma_frac > hbm_frac:
                roof.update({"achieved": round(pf * 1e3, 2), "peak": 2500.0, "unit": "TFLOP/s",
                             "frac": round(mfma_frac, 4)})
        elif qpl > 32:
            # RR_SCAN_F32_CHAIN: f32-input MFMA kernels; > 32 queries per read is past the f32 ridge
            flops = 2.0 * n_local * DIM * qpl
            tf = flops / (avg_ms * 1e-3) / 1e12 if launches else 0.0
            roof = {"bound": "mfma", "achieved": round(tf, 2), "peak": F32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(tf / F32_MATRIX_PEAK_TF, 4), "traffic": traffic,
                    "kernel": "rr_scan_mfma_bf16<4>" if args.dtype == "bf16" else "rr_scan_mfma_f32<4>",
                    "hbm_read_gbs": round(achieved, 2)}
        else:
            roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": ("rr_scan_" if qpl <= 4 else "rr_scan_mfma_") + args.dtype}
        roof.update({"launches": int(launches), "avg_launch_ms": round(avg_ms, 5),
                     "bytes_per_launch": bytes_per_launch, "queries_per_launch": qpl})
        out = {
            "metric": "queries/sec at top-k=100 (hybrid alpha=0.5)" if not args.no_bm25
                      else "queries/sec at top-k=100 (dense only)",
            "value": round(args.batch * args.steps / dt, 3), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "storage": args.dtype,
            "config": {"workload": (f"{'hybrid BM25+dense alpha=0.5' if not args.no_bm25 else 'dense-only cosine'} "
                                    f"top-k={args.k} pool={pool}, {args.docs} products x {DIM} {'bf16-stored' if args.dtype == 'bf16' else 'fp32'}"
                                    + (f", BM25 ~{args.doc_len} tokens/doc vocab {args.vocab} "
                                       f"({stats['nnz']} postings on rank 0)" if not args.no_bm25 else "")
                                    + f", batches of {args.batch} queries"),
                       "docs": args.docs, "docs_per_gpu": n_local, "batch": args.batch, "k": args.k,
                       "pool": pool, "parallelism": f"row-shard x{world} + 1 all-gather"},
            "roofline": roof,
            # the single-query scan on the same shard: the HBM-bound end of the same path
            "roofline_single_query": single,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample_rows = min(n_local, 1_000_000)
            sample = keep[0][:sample_rows].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(args, sample, stats, qsets)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _scan_stats(index):
    from review_recommender_amd import _lib
    tot, n = C.c_double(), C.c_int64()
    _lib.check(_lib.load().rr_index_scan_stats(index.handle, C.byref(tot), C.byref(n)), "rr_index_scan_stats")
    return tot.value, n.value


if __name__ == "__main__":
    main()
