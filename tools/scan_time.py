"""Times the dense scan alone (dev tool): python tools/scan_time.py <rows> <batch> [reps]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from review_recommender_amd.index import ProductIndex
n, b = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
mat = torch.randn((n, 384), device='cuda'); mat /= mat.norm(dim=1, keepdim=True)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
q = np.random.default_rng(0).standard_normal((b, 384)).astype(np.float32)
for _ in range(3): ix.dense_topk(q, 150)
import ctypes as C
from review_recommender_amd import _lib
tot, cnt = C.c_double(), C.c_int64(); _lib.load().rr_index_scan_stats(ix.handle, C.byref(tot), C.byref(cnt))
for _ in range(reps): ix.dense_topk(q, 150)
_lib.load().rr_index_scan_stats(ix.handle, C.byref(tot), C.byref(cnt))
ms = tot.value / cnt.value
print(f"rows {n} batch {b}: scan {ms:.4f} ms/launch, {n*1536/ms/1e6:.0f} GB/s, launches {cnt.value}", flush=True)
