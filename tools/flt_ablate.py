"""Dev tool: what paces the 128-query fp32 filter scan?  python tools/flt_ablate.py [rows]
Variant bits: 1 no epilogue, 2 no B-fragment LDS reads, 4 no MFMA, 8 no lane swaps / conversions."""
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
ix.dense_topk(np.random.default_rng(0).standard_normal((128, 384)).astype(np.float32), 150)
# (every key must be a case of rr_debug_scan_flt's switch: tests/test_ring_register_contract.py checks it.  15 / 31 -- ring
#  loads with nothing reading them -- broke the register-ring contract and are gone from the harness.)
names = {0: "full kernel", 1: "no epilogue", 2: "no B-fragment reads", 4: "no MFMA", 8: "no lane swaps / conversions",
         3: "no epilogue, no B reads", 7: "loads + swaps + conversions only",
         16: "no M-tile maxima stores", 32: "maxima stores non-temporal",
         64: "bf16 plane: full kernel", 128: "bf16 plane: full kernel, ring loads from cache",
         256: "bf16 plane: stamped", 320: "bf16 plane: stamped, ring loads from cache",
         400: "stamped, cached, no B-fragment reads"}
only = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else None
for v in (only or [0, 0] + list(names)[1:] + [0]):
    ms = C.c_float()
    _lib.check(lib.rr_debug_scan_flt(ix.handle, v, 5, C.byref(ms)), "rr_debug_scan_flt")
    nbytes = n * (768 if v >= 64 else 1536)
    print(f"variant {v:2d} {names[v]:44s}: {ms.value:.3f} ms  {nbytes / ms.value / 1e6:.0f} GB/s", flush=True)
