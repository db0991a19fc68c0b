"""One rank's K1 at 8 row shards, with and without the corpus-wide floor, on ONE GPU: all 8 shards
of a 10M-row corpus are built on the device, phase 1 runs on every shard to form the true floor (what the all-reduce
would return), then shard 0's K1 is timed three ways: plain (own threshold), scan + select with the floor, scan alone.
The collective itself is not in these numbers.    python tools/shard_floor_proxy.py [rows_total] [shards]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np, torch
from review_recommender_amd.index import ProductIndex
from review_recommender_amd.engine import HybridSearcher
from review_recommender_amd.sharded import ShardedSearcher, shard_bounds

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, pool, reps = 256, 150, 30
shards = []
for r in range(world):
    lo, hi = shard_bounds(n, world, r)
    g = torch.Generator(device="cuda"); g.manual_seed(100 + r)
    m = torch.randn((hi - lo, 384), device="cuda", generator=g); m /= m.norm(dim=1, keepdim=True)
    ix = ProductIndex(None, n_rows=hi - lo, dim=384, device_ptr=m.data_ptr(), keepalive=m, row_offset=lo)
    ix.set_meta(np.ones(hi - lo), np.full(hi - lo, 4.0))
    shards.append(ShardedSearcher(HybridSearcher(ix, None), n, r, world))
q = torch.from_numpy(np.random.default_rng(3).standard_normal((B, 384)).astype(np.float32)).cuda()
q /= q.norm(dim=1, keepdim=True)
s0 = shards[0].s
for _ in range(3):
    s0.dense_pool(q, pool)
bounds = [sh.local_scan(q, pool) for sh in shards]
floor = torch.stack(bounds).min(dim=0).values
for sh in shards[1:]:
    sh.s.dense_pool(q[:1], pool)           # (drop the parked scans)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def two_phase():
    shards[0].local_scan(q, pool)
    return s0.dense_select(q, pool, floor)


plain_ms = timed(lambda: s0.dense_pool(q, pool)); own = s0.index.select_trace()
floor_ms = timed(two_phase); fl = s0.index.select_trace()
scan_ms = timed(lambda: shards[0].local_scan(q, pool))
a = s0.dense_pool(q, pool); b = two_phase(); torch.cuda.synchronize()
print(json.dumps({"rows_total": n, "shards": world, "rows_per_shard": shards[0].s.index.n_rows, "batch": B, "pool": pool,
                  "k1_plain_ms": round(plain_ms, 4), "k1_scan_plus_select_with_floor_ms": round(floor_ms, 4),
                  "k1_scan_only_ms": round(scan_ms, 4),
                  "query0_mtiles_opened": {"own_threshold": own[2], "with_floor": fl[2]},
                  "query0_rows_kept": {"own_threshold": own[3], "with_floor": fl[3]},
                  "floor_min_max": [float(floor.min()), float(floor.max())]}))
