"""Dev tool: batched dense top-pool, two-pass default vs stored-score pass.
python tools/x3_modes.py <rows> [dtype] -- scan kernel ms (HIP events in the library) and whole
rr_dense_topk_dev ms (torch events on the current stream) for 16 / 32 / 64 queries."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from review_recommender_amd import _lib
from review_recommender_amd.index import ProductIndex
n = int(sys.argv[1]); dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
lib = _lib.load()
mat = torch.randn((n, 384), device="cuda"); mat /= mat.norm(dim=1, keepdim=True)
if len(sys.argv) > 3 and sys.argv[3] == "zeros":     # power experiment: same instruction stream, no bit toggling
    mat.zero_()
if dtype == "bf16":
    mat = mat.to(torch.bfloat16)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat, dtype=dtype)
pool = 150
for b in (16, 32, 64, 128):
    q = torch.randn((b, 384), device="cuda")
    rows = torch.empty((b, pool), dtype=torch.int64, device="cuda"); sc = torch.empty((b, pool), device="cuda")
    def call():
        _lib.check(lib.rr_dense_topk_dev(ix.handle, C.c_void_p(q.data_ptr()), b, pool, C.c_void_p(rows.data_ptr()),
                                         C.c_void_p(sc.data_ptr()), None), "rr_dense_topk_dev")
    for stored in (False, True):
        ix.set_scan_mode(stored)
        for _ in range(3): call()
        torch.cuda.synchronize()
        tot, cnt = C.c_double(), C.c_int64(); lib.rr_index_scan_stats(ix.handle, C.byref(tot), C.byref(cnt))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        lib.rr_index_scan_stats(ix.handle, C.byref(tot), C.byref(cnt))
        tr = ix.select_trace()
        print(f"rows {n} {dtype} batch {b} {'stored  ' if stored else 'two-pass'}: scan {tot.value / cnt.value:.3f} ms, "
              f"whole call {e0.elapsed_time(e1) / 20:.3f} ms, trace {tr[:4]}", flush=True)
