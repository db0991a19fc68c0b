"""Dev check: how does the filter path behave on clustered embeddings (dense score cuts)?
python tools/clustered_check.py [rows] [clusters] [noise] -- rows = unit(centre + noise * g); queries near centres."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from review_recommender_amd import _lib
from review_recommender_amd.index import ProductIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
lib = _lib.load()
g = torch.Generator(device="cuda"); g.manual_seed(1)
centres = torch.randn((k, 384), generator=g, device="cuda"); centres /= centres.norm(dim=1, keepdim=True)
mat = torch.empty((n, 384), device="cuda")
for s in range(0, n, 1_000_000):
    m = min(1_000_000, n - s)
    who = torch.randint(0, k, (m,), generator=g, device="cuda")
    blk = centres[who] + noise / 384 ** 0.5 * torch.randn((m, 384), generator=g, device="cuda")
    mat[s:s + m] = blk / blk.norm(dim=1, keepdim=True)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
B, pool = 128, 150
q = centres[torch.randint(0, k, (B,), generator=g, device="cuda")] + 0.3 / 384 ** 0.5 * torch.randn((B, 384), generator=g, device="cuda")
q /= q.norm(dim=1, keepdim=True)
rows = torch.empty((B, pool), dtype=torch.int64, device="cuda"); sc = torch.empty((B, pool), device="cuda")
def call():
    _lib.check(lib.rr_dense_topk_dev(ix.handle, C.c_void_p(q.data_ptr()), B, pool, C.c_void_p(rows.data_ptr()),
                                     C.c_void_p(sc.data_ptr()), None), "rr_dense_topk_dev")
for _ in range(2): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): call()
e1.record(); torch.cuda.synchronize()
# per-query traces live in the handle's trace buffer: read all 128 through a private copy
tr = ix.select_trace()
ref = (mat.double() @ q[0].double())
top = torch.topk(ref, pool)
ok = set(top.indices.tolist()) == set(rows[0].tolist())
print(f"rows {n} clusters {k} noise {noise}: whole call {e0.elapsed_time(e1) / 5:.3f} ms for {B} queries; trace of query 0 {tr[:6]}; "
      f"score range of its pool {sc[0, 0].item():.4f} .. {sc[0, -1].item():.4f}; top-pool set equals float64 oracle: {ok}", flush=True)
