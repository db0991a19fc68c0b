"""Does a batch's tail (selection -> rescoring -> K2 -> K3 -> answer copy) hide under the NEXT batch's scan when the two
run on CU-masked streams?  One GPU, the 10M-product corpus of bench.py, 256 queries per batch.  For a scan share of S CUs:
  scan_ms      rr_scan_fltq alone on a stream masked to CUs [0, S), its resident grid sized for S (rr_index_set_scan_cus)
  tail_ms      the tail alone on a stream masked to CUs [S, 256)
  step_ms      batch i + 1's scan on the first stream while batch i's tail runs on the second (wall time per batch)
S = 256: no masks, everything on one stream (today's path).
    python tools/overlap_probe.py [docs] [reps] [S,S,...]          -> one JSON line per S"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import json
import time
import numpy as np, torch
from review_recommender_amd import _lib, synth
from review_recommender_amd.device_corpus import build_device_shard
from review_recommender_amd.engine import FusionWeights, HybridSearcher

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
shares = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 240, 232, 224, 208]
B, K, POOL, VOCAB = 256, 100, 150, 200_000
dev = torch.device("cuda", 0)
lib = _lib.load()


class NoDist:
    @staticmethod
    def get_backend():
        return "nccl"


shard = build_device_shard(torch, NoDist, docs=docs, rank=0, world=1, dev=dev, vocab=VOCAB, doc_len=40)
s = shard.sharded.s
index = shard.index
w = FusionWeights(w_dense=0.5, w_bm25=0.5, w_rerank=0.0, w_prior=0.0, w_best=0.0, gate_penalty=1.0)
qsets = []
for i in range(4):
    q = torch.from_numpy(synth.unit_rows(B, 384, 4321 + i)).pin_memory()
    lists = synth.query_terms(B, VOCAB, 99 + i, shard.stats["df"])
    off = np.zeros(B + 1, dtype=np.int32)
    np.cumsum([len(t) for t in lists], out=off[1:])
    qsets.append((q, (np.concatenate(lists).astype(np.int32), off)))
pins = (torch.empty((B, POOL), dtype=torch.int64).pin_memory(), torch.empty((B, K), dtype=torch.int32).pin_memory(),
        torch.empty((B, POOL), dtype=torch.float64).pin_memory())
params = HybridSearcher.make_params(w, K, POOL, POOL, 0)


def scan_stats():
    tot, n = C.c_double(), C.c_int64()
    _lib.check(lib.rr_index_scan_stats(index.handle, C.byref(tot), C.byref(n)))
    return tot.value, n.value


def masked(first, n):
    out = C.c_void_p()
    _lib.check(lib.rr_stream_create_cu_range(0, first, n, C.byref(out)), "rr_stream_create_cu_range")
    return torch.cuda.ExternalStream(out.value, device=dev), out


SPLIT = os.environ.get("PROBE_SPLIT") is not None      # the rescoring on the scans' stream, the rest of the tail on the few CUs
NS = 3                                                 # scan slots
# PROBE_FLOOR=1: a row shard's tail -- the selection against a corpus-wide floor (here: the shard's own bound of its 20th best
# row, what the all-reduce(MIN) returns up to a hair), ~60 M-tiles per query instead of ~400
FLOOR = os.environ.get("PROBE_FLOOR") is not None
KTH = int(os.environ.get("PROBE_KTH", "20"))
floors = {}


def scan(slot, i):
    r = s.dense_scan_slot(slot, qsets[i % 4][0], POOL, kth=KTH if FLOOR else 0)
    assert r is not None
    if FLOOR:
        floors[slot] = r
    return r


def tail(slot, i, parts=7):
    rows, dense = s.dense_select_slot(slot, B, POOL, floor=floors.get(slot) if FLOOR else None, parts=parts)
    bm = s.bm25_at(qsets[i % 4][1], rows, "forward")
    out_rows, cols, order = s.fuse(params, B, rows, dense, bm)
    s.copy_segments([(pins[0], out_rows), (pins[1], order), (pins[2], cols[:, 7, :])])
    return out_rows, cols, order


# warm up the lazy parts on the default stream
for i in range(4):
    shard.sharded.search_batch_dev(qsets[i][0], qsets[i][1], K, w)
torch.cuda.synchronize()
scan_stats()
t = index.select_trace()
print(json.dumps({"query 0 of the last batch": {"groups opened": int(t[1]), "M-tiles opened": int(t[2]), "rows kept": int(t[3]),
                                                  "select_mtiles cycles": [int(t[4]), int(t[5])]}}), flush=True)

for S in shares:
    out = {"docs": docs, "batch": B, "scan_cus": S, "tail_cus": 256 - S if S < 256 else 256, "reps": reps}
    if S >= 256:
        _lib.check(lib.rr_index_set_scan_cus(index.handle, 0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            if FLOOR:
                scan(0, i)
                tail(0, i)
                continue
            rows, cols, order = shard.sharded.search_batch_dev(qsets[i % 4][0], qsets[i % 4][1], K, w)
            s.copy_segments([(pins[0], rows), (pins[1], order), (pins[2], cols[:, 7, :])])
        torch.cuda.synchronize()
        out["step_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        tot, n = scan_stats()
        out["scan_ms"] = round(tot / max(n, 1), 4)
        out["tail_ms"] = round(out["step_ms"] - out["scan_ms"], 4)
        print(json.dumps(out), flush=True)
        continue
    A, hA = masked(0, S)
    T, hT = masked(S, 256 - S)
    _lib.check(lib.rr_index_set_scan_cus(index.handle, S))
    keep = []
    # (a) scan alone on the masked stream, tail alone on its masked stream: strictly one after the other
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    tails = []
    for i in range(reps + 3):
        with torch.cuda.stream(A):
            scan(i % NS, i)
        A.synchronize()
        with torch.cuda.stream(T):
            e[0].record()
            keep.append(tail(i % NS, i))
            e[1].record()
        T.synchronize()
        if i >= 3:
            tails.append(e[0].elapsed_time(e[1]))
        keep = keep[-4:]
    tot, n = scan_stats()
    out["scan_ms"] = round(tot / max(n, 1), 4)
    out["tail_ms"] = round(float(np.median(tails)), 4)
    if os.environ.get("PROBE_ALONE"):          # (under rocprofv3: per-kernel times of the tail on its few CUs)
        print(json.dumps(out), flush=True)
        _lib.check(lib.rr_index_set_scan_cus(index.handle, 0))
        torch.cuda.synchronize()
        keep.clear()
        _lib.check(lib.rr_stream_destroy(hA))
        _lib.check(lib.rr_stream_destroy(hT))
        continue
    # (b) pipelined: scan(i + 1) on A while tail(i) runs on T
    def run(n_steps):
        with torch.cuda.stream(A):
            scan(0, 0)
        for i in range(n_steps):
            with torch.cuda.stream(A):
                scan((i + 1) % NS, i + 1)
            if SPLIT:
                fl = floors.get(i % NS) if FLOOR else None
                with torch.cuda.stream(T):
                    s.dense_select_slot(i % NS, B, POOL, floor=fl, parts=1)
                with torch.cuda.stream(A):
                    s.dense_select_slot(i % NS, B, POOL, floor=fl, parts=2)
                with torch.cuda.stream(T):
                    keep.append(tail(i % NS, i, parts=4))
            else:
                with torch.cuda.stream(T):
                    keep.append(tail(i % NS, i))
            del keep[:-4]
        with torch.cuda.stream(T):
            keep.append(tail(n_steps % NS, n_steps))
    run(4)
    torch.cuda.synchronize()
    scan_stats()
    t0 = time.perf_counter()
    run(reps)
    torch.cuda.synchronize()
    out["step_ms"] = round((time.perf_counter() - t0) / (reps + 1) * 1e3, 4)
    tot, n = scan_stats()
    out["scan_ms_overlapped"] = round(tot / max(n, 1), 4)
    out["split"] = SPLIT
    # the last batch's answer against the straight path's, bit for bit
    got = [t.clone() for t in keep[-1]]
    _lib.check(lib.rr_index_set_scan_cus(index.handle, 0))
    want = shard.sharded.search_batch_dev(qsets[reps % 4][0], qsets[reps % 4][1], K, w)
    torch.cuda.synchronize()
    out["bitwise_equal_to_straight_path"] = None if FLOOR else all(torch.equal(a, b) for a, b in zip(got, want))
    out["floor"] = FLOOR
    print(json.dumps(out), flush=True)
    _lib.check(lib.rr_index_set_scan_cus(index.handle, 0))
    torch.cuda.synchronize()
    keep.clear()
    _lib.check(lib.rr_stream_destroy(hA))
    _lib.check(lib.rr_stream_destroy(hT))
