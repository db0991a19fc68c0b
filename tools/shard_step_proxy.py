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
pins = (torch.empty((B, POOL), dtype=torch.int64).pin_memory(), torch.empty((B, K), dtype=torch.int32).pin_memory(),
        torch.empty((B, POOL), dtype=torch.float64).pin_memory())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]


def step(timed=None):
    if timed is not None: ev[0].record()
    bounds = [sh.sharded.local_scan(q_pin, POOL) for sh in shards]                 # phase 1 of K1 on every rank
    floor = torch.stack(bounds).min(dim=0).values                                   # (= the all-reduce(MIN) of B floats)
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
                  "shard0_query0_mtiles_opened": tr[2], "shard0_query0_rows_kept": tr[3],
                  "note": "median of %d steps; all %d shards on one GPU; collectives (1 all-reduce of B floats, 1 all-gather of "
                          "%d x %d bytes) not included" % (reps, world, world, PayloadLayout(B, POOL).nbytes)}))
