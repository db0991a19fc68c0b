// rr_x3.h -- split-operand pieces shared by the two matrix-core scans (rr_dense_x3.hip: 16x16x32
// tiles, up to 16 queries; rr_dense_x3w.hip: 32x32x16 tiles, 17..64 queries).
#pragma once
#include "rr_common.h"
#include "rr_dense.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define RR_X3_UNITS 48   // 16-byte units (8 bf16) per 384-d query / bf16 row

__device__ __forceinline__ unsigned int rr_pack_hi(float lo_elem, float hi_elem) {
    // {hi16(hi_elem), hi16(lo_elem)}: element order inside a bf16 pair is low half first
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}

// The arithmetic of one K-block (32 dims) of one 16-row M-tile x 16-query tile.  The scan kernel and
// the rescoring kernel both go through these two functions, in the same K-block order, so a score
// recomputed by rr_rescore_x3 equals the scan's bit for bit.
struct rr_x3_afrag { bf16x8 a1, a2, a3; };

template <bool A_BF16>
__device__ __forceinline__ rr_x3_afrag rr_x3_split(u32x4 lo_unit, u32x4 hi_unit) {
    rr_x3_afrag f;
    if (A_BF16) {
        f.a1 = __builtin_bit_cast(bf16x8, lo_unit);        // the stored element is one exact bf16 term
        f.a2 = f.a1;
        f.a3 = f.a1;
        return f;
    }
    const f32x4 lo = __builtin_bit_cast(f32x4, lo_unit);
    const f32x4 hi = __builtin_bit_cast(f32x4, hi_unit);
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    float h1[8], h2[8], h3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h1[e] = __uint_as_float(__float_as_uint(x[e]) & 0xFFFF0000u);
        const float r1 = x[e] - h1[e];
        h2[e] = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
        h3[e] = r1 - h2[e];
    }
    u32x4 p1, p2, p3;
    p1.x = rr_pack_hi(h1[0], h1[1]); p1.y = rr_pack_hi(h1[2], h1[3]);
    p1.z = rr_pack_hi(h1[4], h1[5]); p1.w = rr_pack_hi(h1[6], h1[7]);
    p2.x = rr_pack_hi(h2[0], h2[1]); p2.y = rr_pack_hi(h2[2], h2[3]);
    p2.z = rr_pack_hi(h2[4], h2[5]); p2.w = rr_pack_hi(h2[6], h2[7]);
    p3.x = rr_pack_hi(h3[0], h3[1]); p3.y = rr_pack_hi(h3[2], h3[3]);
    p3.z = rr_pack_hi(h3[4], h3[5]); p3.w = rr_pack_hi(h3[6], h3[7]);
    f.a1 = __builtin_bit_cast(bf16x8, p1);
    f.a2 = __builtin_bit_cast(bf16x8, p2);
    f.a3 = __builtin_bit_cast(bf16x8, p3);
    return f;
}

// Query planes for a scan launch: planes[3][slots][384] bf16 (truncating split, exact sum), with the
// k order inside every 32-dim group that the scan's A-fragment loads imply (RR_X3_ORDER_*).
#define RR_X3_ORDER_NATURAL 0    // rr_scan_mfma_x3 on a bf16 matrix; rr_scan_x3w on an fp32 matrix
#define RR_X3_ORDER_PAIR64 1     // rr_scan_mfma_x3 on an fp32 matrix: slot (kg, j) = dim 4kg + j | 16 + 4kg + (j - 4)
#define RR_X3_ORDER_WIDE_BF16 2  // rr_scan_x3w on a bf16 matrix: K-step v, slot (h, j) = dim 16h + 8v + j
void rr_launch_split_queries(const float* d_q, unsigned short* planes, int slots, int order, hipStream_t st);
// RR_SCAN_MODE_STORED on the handle (or RR_X3_STORED=1 in the environment): single stored-score pass
bool rr_x3_stored_path(const rr_index* ix);
// 32x32x16-tile scan for 17..64 queries (rr_dense_x3w.hip)
int rr_dense_chunk_x3w(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                       float* d_scores, hipStream_t st);

// Stored-score pass of rr_scan_x3w + rr_select for the queries (<= 64) whose flag in `flags` is up;
// every launch in it returns at once when none is.
int rr_dense_chunk_x3w_fallback(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                                float* d_scores, const int32_t* flags, hipStream_t st);

// the same for a whole filter call (<= 256 queries) in two launches (slices of 64 queries, rr_dense_x3w.hip)
int rr_dense_x3w_fallback_all(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows, float* d_scores,
                              const int32_t* flags, hipStream_t st);

// Resolution of the filter scan's packed 8-row gaps: half the smallest finite positive error bound of the
// launch's queries (0: none).  Every wave / workgroup recomputes it from the nq (<= 128) bounds.
__device__ __forceinline__ float rr_flt_gap_step(const float* __restrict__ eps, int nq) {
    const int lane = threadIdx.x & 63;
    float e = INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float v = lane + 64 * i < nq ? eps[lane + 64 * i] : INFINITY;
        e = fminf(e, (v > 0.f && v < 3.0e38f) ? v : INFINITY);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) e = fminf(e, __shfl_xor(e, m, 64));
    return e < 3.0e38f ? 0.5f * e : 0.f;
}

// The 4-bit gap code of an 8-row M-tile: how far its maximum e sits below the 32-row tile's maximum m (m >= e), in units
// of `step` (inv_step = 0.9999 / step), ROUNDED DOWN on a two-slope scale: codes 0..7 = [0, 8) steps one step each,
// 8..15 = [8, 24) steps two steps each, 15 = "at least 22 steps" (an ordinary M-tile next to a top row sits ~60 steps
// below it and must not be opened with it).  Four instructions: the code is bits 20..23 of the float 8 + gap (clamped
// below 32): three mantissa bits and the low exponent bit.  inf - inf = NaN comes out as 15: the decoded bound
// m - 22 steps is then still +-inf, what the M-tile holds.  rr_flt_gap_steps() is the decoder (rr_select_mtiles).
__device__ __forceinline__ uint32_t rr_flt_gap_code(float m, float e, float inv_step) {
    float g8 = __builtin_fmaf(m - e, inv_step, 8.0f);
    asm("v_min_f32 %0, 0x41ffeb85, %0" : "+v"(g8));          // min(g8, 31.99): the other operand for a NaN
    return (__float_as_uint(g8) >> 20) & 15u;
}
__device__ __forceinline__ float rr_flt_gap_steps(uint32_t code) { return code < 8u ? (float)code : (float)(2u * code - 8u); }

// NaN scores and pad rows rank last (rows row0 .. row0+3 of one query)
__device__ __forceinline__ f32x4 rr_x3_canon(f32x4 v, int64_t row0, int64_t n_rows) {
    v.x = (row0 + 0 < n_rows && v.x == v.x) ? v.x : -INFINITY;
    v.y = (row0 + 1 < n_rows && v.y == v.y) ? v.y : -INFINITY;
    v.z = (row0 + 2 < n_rows && v.z == v.z) ? v.z : -INFINITY;
    v.w = (row0 + 3 < n_rows && v.w == v.w) ? v.w : -INFINITY;
    return v;
}
