// rr_common.h -- shared host/device helpers of librr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <string>

#include "../../include/rr_hip.h"

// ---------------------------------------------------------------- errors
void rr_set_error(const char* fmt, ...);

#define RR_HIP_TRY(expr)                                                        \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) {                                                 \
            rr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                   \
            return RR_E_HIP;                                                    \
        }                                                                       \
    } while (0)

#define RR_REQUIRE(cond, ...)             \
    do {                                  \
        if (!(cond)) {                    \
            rr_set_error(__VA_ARGS__);    \
            return RR_E_INVALID;          \
        }                                 \
    } while (0)

// ---------------------------------------------------------------- handles
// What ONE batch's scan writes and its selection reads (the staged queries, their bf16 planes and error bounds, the tile /
// group maxima, the prefilter's thresholds, the parked phase-1 state).  An index has RR_SCAN_SLOTS of them
// ("slots"): the pipelined K1 (rr_dense_scan_slot_dev / rr_dense_select_part_dev) scans batch i + 1 into one slot on one
// stream while batch i's selection reads another on another stream (three: a batch's last selection part, its fallbacks
// included, may still hold its slot when the scan after next starts).  The ACTIVE slot's pointers live in the rr_index
// fields of the same names (every launch site reads them there); rr_slot_activate swaps them under the handle's mutex.
#define RR_SCAN_SLOTS 3
struct rr_scan_slot {
    float* d_q = nullptr;
    void* d_qplanes = nullptr;
    float* d_eps = nullptr;
    float* d_gmax = nullptr;
    uint32_t* d_smax = nullptr;
    int32_t maxima_q = 0;
    float* d_flt_samp = nullptr;
    float* d_flt_sigma = nullptr;
    uint32_t* d_flt_prog = nullptr;
    uint32_t flt_seq = 0;
    bool flt_prep_fresh = false;
    void* flt_pending = nullptr;
};

struct rr_index {
    int device = 0;
    int64_t n_rows = 0;
    int32_t dim = 0;        // caller's dimension
    int32_t dim_pad = 0;    // multiple of 64
    int32_t dtype = RR_DTYPE_F32;
    int64_t row_offset = 0;
    void* d_matrix = nullptr;   // n_rows x dim_pad
    bool owns_matrix = true;
    double* d_n_reviews = nullptr;
    double* d_avg_stars = nullptr;
    double* d_log1p_n = nullptr;
    bool has_meta = false;
    // scratch for K1 (grown on demand)
    float* d_sims = nullptr;     // [qcap][n_pad]
    float* d_gmax = nullptr;     // [qcap][n_tiles]
    int32_t* d_sel_trace = nullptr;  // [8][4] path trace of the last selection launch
    int32_t* d_flag_list = nullptr;  // [16] count + the (at most 8) flagged queries of a filter call served by the single-query chain
    uint32_t* d_smax = nullptr;  // [qcap][n_super] ordered keys of super-tile maxima
    int32_t scratch_q = 0;       // query slots d_sims was sized for
    int32_t maxima_q = 0;        // ... and d_gmax / d_smax (per slot)
    float* d_eps = nullptr;      // [RR_SEL_MAXQ] error bounds of the filter scan's scores (rr_x3_scratch.eps; per slot)
    float* d_q = nullptr;        // staged queries [RR_MAX_BATCH][dim_pad]
    float norm_bound = -1.f;     // upper bound of the largest row norm; < 0: not computed (any write to the matrix resets it)
    float delta_bound = 0.f;     // ... and of the largest ||row - bf16(row)||
    int32_t last_scan[5] = {0, 0, 0, 0, 0};   // kernel id, template variant, queries in the launch, bf16 MFMA terms per dim, bytes per scanned element
    // bf16 filter plane of an fp32 matrix (rr_dense_flt.hip): every element rounded once to nearest even.  The batched
    // filter scan reads it instead of the fp32 rows (half the bytes per launch, the same approximate scores: the scan
    // rounds to bf16 anyway); candidates are rescored on the fp32 rows.  Built lazily, dropped by any write to the matrix.
    float* d_flt_samp = nullptr;     // [<= RR_FLT_SAMP_CAP][RR_FLT_MAXQ] sampled tile maxima (store prefilter of rr_scan_flt)
    float* d_flt_sigma = nullptr;    // [RR_FLT_MAXQ] per-query store threshold
    uint32_t* d_flt_prog = nullptr;  // [2][n_waves] progress words of the two-set scan launch (pairs of waves keep in step)
    uint32_t flt_seq = 0;            // launch counter of the two-set scan (epoch of the progress words)
    bool scratch_small = false;      // no room for one score slice per 64 queries of a call: the fallback goes block by block
    bool flt_prep_fresh = false;     // rr_flt_pad_prep has written the planes / bounds of the queries in d_q (slots 0 ..): the next filter call skips its own preparation launch
    void* flt_pending = nullptr;     // rr_flt_pending: what a scan-only call (row shards, phase 1) left for its selection
    unsigned short* d_shadow = nullptr;
    bool shadow_valid = false;
    int32_t use_shadow = 1;
    int32_t scan_mode = 0;       // RR_SCAN_MODE_* (rr_index_set_scan_mode)
    void* d_x3 = nullptr;        // two-pass selection scratch of the split-operand scan (rr_x3_scratch)
    void* d_qplanes = nullptr;   // [3][64][384] bf16: one launch's queries split in three bf16 terms
    int64_t* d_rows_out = nullptr;  // host-API staging [RR_MAX_BATCH][RR_MAX_POOL]
    float* d_scores_out = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // last scan
    // the K1 scratch is per handle: a call on another stream than the previous one first waits for the
    // previous call's last kernel (event), so two host threads on different streams cannot overlap on it
    hipEvent_t ev_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_done = false;
    // pipelined K1: the inactive slot's state (rr_scan_slot) and which slot the fields above hold
    rr_scan_slot parked[RR_SCAN_SLOTS];
    int cur_slot = 0;
    // the last launch sequence that touched a slot (its scan, or a part of its selection), and the stream it went to: the
    // next one on ANOTHER stream waits for it.  (ev_done / last_stream above: the same for the selection scratch all slots share.)
    hipEvent_t slot_ev[RR_SCAN_SLOTS] = {};
    hipStream_t slot_stream[RR_SCAN_SLOTS] = {};
    bool slot_has[RR_SCAN_SLOTS] = {};
    int32_t n_cus = 0;           // CUs of the device
    int32_t scan_cus = 0;        // rr_index_set_scan_cus: CUs the scans' stream may use (0 = all): sizes every scan's resident grid
    bool timing_valid = false;
    static const int kRing = 512;               // event pairs around every scan launch
    hipEvent_t ring0[kRing] = {}, ring1[kRing] = {};
    int64_t ring_head = 0, ring_tail = 0;       // [tail, head) recorded, not yet drained
    std::mutex mu;
};

struct rr_bm25 {
    int device = 0;
    int64_t n_docs = 0, n_terms = 0, nnz = 0;
    int64_t row_offset = 0;
    int64_t* d_post_indptr = nullptr;
    int32_t* d_post_docs = nullptr;
    int32_t* d_post_tf = nullptr;
    int64_t* d_doc_indptr = nullptr;
    int32_t* d_doc_terms = nullptr;
    int32_t* d_doc_tf = nullptr;
    int32_t* d_doc_len = nullptr;
    double* d_idf = nullptr;
    double avgdl = 1.0, k1 = 1.5, b = 0.75;
    bool owns_arrays = true;
    double* d_scores = nullptr;   // n_docs, scratch of get_scores (all zeros outside the slices d_dirty marks)
    unsigned char* d_dirty = nullptr;   // per RR_SLICE documents: the last get_scores left something there
    hipStream_t stream = nullptr;
    std::mutex mu;
};

static inline int64_t rr_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Order-preserving map float -> uint32 (a < b  <=>  key(a) < key(b)); NaN is
// canonicalised by the callers before keys are taken.
__device__ __forceinline__ uint32_t rr_f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rr_key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// Sum over the 16 lanes of a DPP row; every lane of the row ends with the same
// value.  Fixed association: ((p0+p1)+(p2+p3)) per quad, quads paired by
// row_half_mirror, halves by row_mirror -- the order oracle/kernel_order.c mirrors.
__device__ __forceinline__ float rr_row16_sum(float v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v = v + __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v = v + __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true);  // row_half_mirror
    v = v + __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true);  // row_mirror
    v = v + __int_as_float(t);
    return v;
}

__device__ __forceinline__ float rr_wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

#endif  // __HIPCC__
