"""A full sharded step with ALL 8 shards of the 10M-product corpus on ONE GPU (the builder bench.py uses, per rank):
per batch every "rank" runs phase 1 of K1 (scan + bound), the floor is the minimum of the 8 bounds (what the all-reduce
returns), every rank selects against it, scores BM25 at its candidates and gathers their metadata into its payload buffer,
the 8 buffers are laid side by side (what the all-gather leaves on every rank) and K3 merges and fuses them; the answer is
written to pinned host memory.  One rank's share of that = (the 8 ranks' local work) / 8 + one merge: the time of a rank's
step on an 8-GPU node WITHOUT the two collectives' wire / latency (not measurable on one GPU).
    python tools/shard_step_proxy.py [docs_total] [shards] [reps]        -> one JSON line
BM25 statistics (idf, avgdl) are per shard here (no process group on one GPU): timing only, not an answer to compare."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np, torch
from review_recommender_amd import synth
from review_recommender_amd.device_corpus import build_device_shard
from review_recommender_amd.engine import FusionWeights
from review_recommender_amd.sharded import PayloadLayout, PendingBatch, PendingExchange

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
B, K, POOL, VOCAB = 256, 100, 150, 200_000


class LocalStats:                      # stands in for torch.distributed at build time: statistics stay per shard
    @staticmethod
    def get_backend():
        return "nccl"

    @staticmethod
    def all_reduce(t):
        return t


dev = torch.device("cuda", 0)
shards = [build_device_shard(torch, LocalStats, docs=docs, rank=r, world=world, dev=dev, vocab=VOCAB, doc_len=40)
          for r in range(world)]
w = FusionWeights(w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0, w_best=0.0, gate_penalty=1.0)
q_pin = torch.from_numpy(synth.unit_rows(B, 384, 4321)).pin_memory()
terms = synth.query_terms(B, VOCAB, 99, shards[0].stats["df"])
_off = np.zeros(B + 1, dtype=np.int32)
np.cumsum([len(t) for t in terms], out=_off[1:])
terms = (np.concatenate(terms).astype(np.int32), _off)          # (flattened once, like bench.py's query sets)
pins = (torch.empty((B, POOL), dtype=torch.int64).pin_memory(), torch.empty((B, K), dtype=torch.int32).pin_memory(),
        torch.empty((B, POOL), dtype=torch.float64).pin_memory())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]


NO_FLOOR = os.environ.get("RR_NO_SHARD_FLOOR") is not None     # every shard selects on its own threshold: one collective less


def step(timed=None):
    if timed is not None: ev[0].record()
    if NO_FLOOR:
        floor = False                                                               # (local_payload then runs the plain K1)
    else:
        bounds = [sh.sharded.local_scan(q_pin, POOL) for sh in shards]             # phase 1 of K1 on every rank
        floor = torch.stack(bounds).min(dim=0).values                               # (= the all-reduce(MIN) of B floats)
    if timed is not None: ev[1].record()
    lay = PayloadLayout(B, POOL)
    gathered = torch.empty((world, lay.nbytes), dtype=torch.uint8, device=dev)     # (= what the all-gather leaves)
    for r, sh in enumerate(shards):                                                 # phase 2 + K2 + metadata -> payload
        sh.sharded.local_payload(q_pin, terms, POOL, floor=floor, buf=gathered[r])
    if timed is not None: ev[2].record()
    s0 = shards[0].sharded
    rows, cols, order = s0.finish(PendingBatch(B, K, POOL, POOL, 0, w, lay, gathered[0], PendingExchange(gathered)))
    s0.s.copy_segments([(pins[0], rows), (pins[1], order), (pins[2], cols[:, 7, :])])
    if timed is not None:
        ev[3].record()
        torch.cuda.synchronize()
        timed.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])


OVERLAP = os.environ.get("PROXY_OVERLAP")        # "1" / a CU count: rank 0's step with enable_overlap, the other ranks' parts replayed


def overlapped_rank_step():
    """ONE rank's pipelined step (ShardedSearcher.enable_overlap) with everything the other seven ranks contribute replayed
    from a first, straight pass: the corpus-wide floor of the query set (what the all-reduce returns) and their payload
    blocks (what the all-gather brings).  Rank 0's scan + bound, its selection against that floor, K2, metadata gather, the
    merge of all eight blocks and K3 run live, two batches in flight behind the one being submitted; no wire."""
    import review_recommender_amd.sharded as S
    bounds = [sh.sharded.local_scan(q_pin, POOL) for sh in shards]
    floor = torch.stack(bounds).min(dim=0).values.clone()
    lay = PayloadLayout(B, POOL)
    others = torch.empty((world, lay.nbytes), dtype=torch.uint8, device=dev)
    for r, sh in enumerate(shards):
        sh.sharded.local_payload(q_pin, terms, POOL, floor=floor, buf=others[r])
    torch.cuda.synchronize()
    s0 = shards[0].sharded
    s0.force_payload = True
    kth = min(max((POOL + world - 1) // world, (POOL + 7) // 8 + 1), POOL)

    # the two collectives, replayed: rank 0's own bound is computed live and then overwritten by the corpus-wide minimum; its
    # payload block is gathered beside the seven recorded ones
    real_start = S.exchange_start

    def fake_start(buf, w_, group=None, out=None):
        g = out if out is not None else torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)
        g.copy_(others)
        g[0].copy_(buf)
        return S.PendingExchange(g)
    S.exchange_start = fake_start
    orig_submit = s0._submit_overlapped

    def submit(q, tl, k):
        # (world is 1 for this searcher: ask for the bound ourselves and plant the recorded floor in it)
        ov = s0._ov
        t = None
        slot = ov.seq % ov.SLOTS
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.stream(ov.scan_stream):
            ring = ov.buffers(("bound", B), lambda: torch.empty((B,), dtype=torch.float32, device=dev))
            bound = s0.s.dense_scan_slot(slot, q, POOL, kth, bound_out=ring[slot])
            bound.copy_(floor)                                   # (= the all-reduce(MIN)'s result)
        ov.seq += 1
        ov.unfinished += 1
        t = PendingBatch(B, k, POOL, POOL, 0, w, lay, None, None, stage=0, slot=slot, bound=bound, floor_work=None, terms=tl,
                         bm25_mode="forward")
        prev, ov.pending = ov.pending, t
        if prev is not None:
            s0._build_payload(prev)
        return t
    cus = 160 if OVERLAP == "1" else int(OVERLAP)
    assert s0.enable_overlap(cus)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))       # (not the NULL stream: it synchronises with the masked streams)
    s0.world = world                                            # the merge sees eight blocks of POOL candidates
    inflight = []

    def one():
        inflight.append(submit(q_pin, terms, K))
        while len(inflight) > 2:
            rows, cols, order = s0.finish(inflight.pop(0))
            s0.s.copy_segments([(pins[0], rows), (pins[1], order), (pins[2], cols[:, 7, :])])
    for _ in range(8):
        one()
    torch.cuda.synchronize()
    if os.environ.get("PROXY_STAGES"):
        # the three stages one after the other, each alone on its masked stream (HIP events on that stream)
        while inflight:
            s0.finish(inflight.pop(0))
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        acc = np.zeros(3)
        ov = s0._ov
        for _ in range(reps):
            with torch.cuda.stream(ov.scan_stream):
                ev[0].record()
            t = submit(q_pin, terms, K)
            with torch.cuda.stream(ov.scan_stream):
                ev[1].record()
            torch.cuda.synchronize()
            with torch.cuda.stream(ov.tail_stream):
                ev[2].record()
            ov.pending = None
            s0._build_payload(t)
            with torch.cuda.stream(ov.tail_stream):
                ev[3].record()
            torch.cuda.synchronize()
            with torch.cuda.stream(ov.tail_stream):
                ev[4].record()
            s0.finish(t)
            with torch.cuda.stream(ov.tail_stream):
                ev[5].record()
            torch.cuda.synchronize()
            acc += [ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]), ev[4].elapsed_time(ev[5])]
        acc /= reps
        print(json.dumps({"stages_alone_ms": {"scan_and_bound_on_%d_cus" % ov.scan_cus: round(acc[0], 4),
                                              "select_k2_meta_payload_on_the_rest": round(acc[1], 4),
                                              "merge_k3_on_the_rest": round(acc[2], 4)}}), flush=True)
    import time
    prof = None
    if os.environ.get("PROXY_PROFILE"):
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    if os.environ.get("PROXY_HOST_SPLIT"):
        # where the host's time per step goes (wall-clock around each call; the GPU is not waited for inside them)
        acc = {}
        def timed(name, fn):
            def w(*a_, **k_):
                h0 = time.perf_counter()
                r = fn(*a_, **k_)
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - h0
                return r
            return w
        s0._build_payload = timed("build_payload", s0._build_payload)
        s0._finish_overlapped = timed("finish", s0._finish_overlapped)
        s0.s.dense_scan_slot = timed("  scan_slot", s0.s.dense_scan_slot)
        s0.s.dense_select_slot = timed("  select_slot", s0.s.dense_select_slot)
        s0.s.bm25_at = timed("  bm25_at", s0.s.bm25_at)
        s0.s.copy_segments = timed("  copy_segments", s0.s.copy_segments)
        S.exchange_start = timed("  exchange_start(fake)", S.exchange_start)
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    host = (time.perf_counter() - t0) / reps * 1e3
    if os.environ.get("PROXY_HOST_SPLIT"):
        print(json.dumps({"host_us_per_step": {k: round(v / reps * 1e6, 1) for k, v in acc.items()}, "total_us": round(host * 1e3, 1)}), flush=True)
    if prof is not None:
        prof.disable()
        import pstats
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(22)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    while inflight:
        s0.finish(inflight.pop(0))
    torch.cuda.synchronize()
    print(json.dumps({"docs_total": docs, "shards": world, "rows_per_shard": shards[0].n_local, "batch": B, "pool": POOL, "k": K,
                      "overlapped_rank_step_ms_without_collectives": round(ms, 4), "host_enqueue_ms_per_step": round(host, 4),
                      "scan_cus": s0._ov.scan_cus, "implied_queries_per_s_8_gpus_without_collectives": round(B / ms * 1e3, 1),
                      "note": "rank 0's pipelined step (scan | selection + K2 + metadata + payload | merge of 8 blocks + K3 + answer "
                              "out), the other ranks' floor and payload blocks replayed; median-free wall time over %d steps" % reps}))
    S.exchange_start = real_start
    s0.disable_overlap()                                        # (the masked streams are destroyed before the process ends)


if OVERLAP:
    overlapped_rank_step()
    sys.exit(0)

for _ in range(5):
    step()
torch.cuda.synchronize()
t = []
for _ in range(reps):
    step(t)
t = np.median(np.array(t), axis=0)
scan_rank, payload_rank, merge = t[0] / world, t[1] / world, t[2]
tr = shards[0].index.select_trace()
print(json.dumps({"docs_total": docs, "shards": world, "rows_per_shard": shards[0].n_local, "batch": B, "pool": POOL, "k": K,
                  "per_rank_scan_and_bound_ms": round(float(scan_rank), 4),
                  "per_rank_select_k2_meta_payload_ms": round(float(payload_rank), 4),
                  "merge_k3_and_answer_out_ms": round(float(merge), 4),
                  "per_rank_step_ms_without_collectives": round(float(scan_rank + payload_rank + merge), 4),
                  "implied_queries_per_s_8_gpus_without_collectives": round(B / float(scan_rank + payload_rank + merge) * 1e3, 1),
                  "shard0_query0_mtiles_opened": tr[2], "shard0_query0_rows_kept": tr[3], "corpus_wide_floor": not NO_FLOOR,
                  "note": "median of %d steps; all %d shards on one GPU; collectives (1 all-reduce of B floats, 1 all-gather of "
                          "%d x %d bytes) not included" % (reps, world, world, PayloadLayout(B, POOL).nbytes)}))
