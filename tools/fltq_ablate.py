"""Dev tool: what paces rr_scan_fltq (256 queries, query-stationary)?  python tools/fltq_ablate.py [rows] [variants]
Needs librr_hip_dbg.so (python review-recommender_amd/build.py --debug).  Variants: 3000 = the hand-scheduled loop (the
product's kernel), 3000 + bits = its generated ablations with the loop's cycle count (gen_fltq_loop.py --abl, the list
build.FLTQ_ABLATIONS: 128 nothing ablated, 1 no LDS-DMA, 2 no epilogue / stores, 3 both, 4 no wait + barrier, 8 no A reads,
64 pieces from cache, 66; 1025 .. 1030 schedule variants).  (The C++-body variants 1000 + bits / 2000 + bits of round 3's first
half are gone from the harness.)"""
import os, sys; os.environ["RR_DEBUG_HARNESS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from review_recommender_amd import _lib
from review_recommender_amd.index import ProductIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lib = _lib.load()
mat = torch.randn((n, 384), device="cuda"); mat /= mat.norm(dim=1, keepdim=True)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
ix.dense_topk(np.random.default_rng(0).standard_normal((256, 384)).astype(np.float32), 150)
cmp = (C.c_int64 * 8)()
_lib.check(lib.rr_debug_fltq_compare(ix.handle, cmp), "rr_debug_fltq_compare")
print(f"asm loop vs C++ bodies: {cmp[0]} of {cmp[2]} tile words differ, {cmp[1]} of {cmp[3]} group maxima differ"
      + (f"; first at word {cmp[4]}: {cmp[5]:#x} vs {cmp[6]:#x}" if cmp[0] else ""), flush=True)
only = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3000, 3128, 3000, 3128, 3002, 3001, 3003, 3000]
for v in only:
    ms = C.c_float()
    _lib.check(lib.rr_debug_scan_flt(ix.handle, v, int(os.environ.get("RR_ABLATE_REPS", "5")), C.byref(ms)), "rr_debug_scan_flt")
    print(f"variant {v:3d}: {ms.value:.3f} ms  {n * 768 / ms.value / 1e6:.0f} GB/s", flush=True)
