"""Dev tool: what paces the 64-query fp32 batched scan?  python tools/x3w_ablate.py [rows]
Variant bits: 1 no operand split, 2 no B-fragment LDS reads, 8 no lane swap, 16 in-kernel clock stamps."""
import os, sys; os.environ["RR_DEBUG_HARNESS"] = "1"   # librr_hip_dbg.so (python review-recommender_amd/build.py --debug)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from review_recommender_amd import _lib
from review_recommender_amd.index import ProductIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lib = _lib.load()
mat = torch.randn((n, 384), device="cuda"); mat /= mat.norm(dim=1, keepdim=True)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=mat.data_ptr(), keepalive=mat)
ix.dense_topk(np.random.default_rng(0).standard_normal((64, 384)).astype(np.float32), 150)
names = {0: "full kernel", 16: "full kernel + clock stamps", 48: "stamps, one wave per SIMD", 59: "stamps, 1 wave/SIMD, MFMA+loads only", 1: "no operand split", 2: "no B-fragment reads",
         3: "no split, no B reads", 9: "no split, no lane swap", 11: "MFMAs + ring loads only",
         15: "ring loads only", 47: "ring loads only, 1 wave/SIMD", 79: "ring loads only, no epilogue",
         143: "ring loads only, no LDS fill", 207: "ring loads only, no epilogue, no LDS fill", 64: "full kernel, no epilogue"}
order = list(names) if len(sys.argv) < 3 else [int(a) for a in sys.argv[2].split(",")]
names[32] = "full kernel, one wave per SIMD"
names[75] = "MFMAs + ring loads, no epilogue"; names[65] = "no operand split, no epilogue"; names[66] = "no B reads, no epilogue"
for v in order:
    name = names[v]
    ms = C.c_float()
    _lib.check(lib.rr_debug_scan_x3w(ix.handle, v, 5, C.byref(ms)), "rr_debug_scan_x3w")
    print(f"variant {v:2d} {name:28s}: {ms.value:.3f} ms  {n * 1536 / ms.value / 1e6:.0f} GB/s", flush=True)
