"""Dev tool for profiling: python tools/x3_run.py <rows> <dtype> <batch> [reps] -- repeated rr_dense_topk_dev calls."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from review_recommender_amd import _lib
from review_recommender_amd.index import ProductIndex
n, dtype, b = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = _lib.load()
mat = torch.randn((n, 384), device="cuda"); mat /= mat.norm(dim=1, keepdim=True)
if dtype == "bf16":
    mat = mat.to(torch.bfloat16)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat, dtype=dtype)
q = torch.randn((b, 384), device="cuda")
rows = torch.empty((b, 150), dtype=torch.int64, device="cuda"); sc = torch.empty((b, 150), device="cuda")
for _ in range(reps):
    _lib.check(lib.rr_dense_topk_dev(ix.handle, C.c_void_p(q.data_ptr()), b, 150, C.c_void_p(rows.data_ptr()),
                                     C.c_void_p(sc.data_ptr()), None), "rr_dense_topk_dev")
torch.cuda.synchronize()
print("done", ix.select_trace()[:8])
