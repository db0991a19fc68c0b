// rr_debug.h -- timing-only ablation harness of the scan kernels.  NOT part of the product ABI (include/rr_hip.h):
// compiled only with -DRR_DEBUG_HARNESS into librr_hip_dbg.so (build.py: build_library(debug=True)), which the
// tools/ scripts load through RR_DEBUG_HARNESS=1.  The ablated kernels leave garbage in the scan scratch and some of
// them give up the register-liveness contract of the asynchronous ring loads -- never part of a search.
#pragma once
#ifdef RR_DEBUG_HARNESS
#include "rr_common.h"
extern "C" {
// tools/x3w_ablate.py: variants of the 64-query fp32 split-operand scan (bit 0: no operand split, bit 1: no B-fragment
// reads, bit 2: no MFMA, bit 3: no lane swap)
int rr_debug_scan_x3w(rr_index* ix, int32_t variant, int32_t reps, float* out_ms);
// tools/flt_ablate.py: variants of the filter scans (the list is in rr_dense_flt.hip)
int rr_debug_scan_flt(rr_index* ix, int32_t variant, int32_t reps, float* out_ms);
// the hand-scheduled loop of rr_scan_fltq against its C++ bodies, tile word by tile word (out: 7 values, see the definition)
int rr_debug_fltq_compare(rr_index* ix, int64_t* out);
// tools/k5_stamps.py: in-kernel phase clocks of the last ce_ffn_fused launch (rr_ce.hip)
int rr_debug_ce_ffn_stamps(unsigned long long* out20);
// tools/k5_h2_stamps.py: phase clocks of one workgroup of the last FFN1 ce_gemm_h2 launch (rr_ce_h2.hip)
int rr_debug_ce_h2_stamps(unsigned long long* out16);
// tests/test_gpu_k5.py: one ce_gemm_h2 product on caller data (pack -> GEMM -> fp32), see the definition
int rr_debug_ce_h2_gemm(int32_t epi, int32_t M, int32_t N, int32_t K, const float* d_x, const float* d_w, const float* d_bias, int32_t qcols,
                        float* d_out, int32_t* flag_out);
}
#endif
