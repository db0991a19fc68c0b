"""Dev tool: phases of rr_select_mtiles for query 0 of a 256-query call (cycle stamps of the kernel's trace words).
python tools/sel_trace.py [rows]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from review_recommender_amd.index import ProductIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
m = torch.empty((n, 384), device="cuda")
for s in range(0, n, 1_250_000):
    b = torch.randn((min(1_250_000, n - s), 384), device="cuda", generator=g); m[s:s + b.shape[0]] = b / b.norm(dim=1, keepdim=True)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=m.data_ptr(), keepalive=m)
q = np.random.default_rng(0).standard_normal((256, 384)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
for _ in range(3):
    ix.dense_topk(q, 150)
t = ix.select_trace()
print("path", t[0], "groups opened", t[1], "M-tiles opened", t[2], "rows kept", t[3], "cycles: threshold + group list", t[4], "| M-tile list", t[5])
