// rr_ce_h2.hip -- K5, reference-precision mode (RR_CE_PRECISION_F32) on the fp16 matrix cores.
//
// The reference runs both encoders in fp32 torch (app/app_product_search.py:250-251, 277-278).  An fp32 product needs 24 x 24
// significand bits; fp16 carries 11, so x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048) holds 22-23 bits of x
// (the subtraction is exact; the scale keeps lo a normal fp16 down to |x - hi| = 3e-8, and subnormal fp16 values are
// multiplied as they are), and
//     x * y  =  hi_x hi_y  +  (hi_x lo_y + lo_x hi_y) / 2048  +  O(2^-22 x y)
// is three v_mfma_f32_16x16x32_f16 instead of the six bf16 products of ce_gemm_x3 (rr_ce.hip): the first into one fp32
// accumulator, the two cross terms into a second one that is scaled by 2^-11 once, in the epilogue.  What is dropped (lo lo,
// 2^-22) is a quarter of the rounding of ONE fp32 multiply-add chain's noise: on N(0,1) x N(0, 0.05^2) operands, K = 384 or
// 1536, the result sits 9.4e-8 (rms, relative) from the exact product against 4.4e-7 for an fp32 GEMM (numpy sgemm) --
// tests/test_gpu_k5.py holds the whole forward to the same 1e-5 / 5e-6 bars as before.  fp16's range is the price: a value
// beyond 65504 cannot be split; every producer ORs a flag when it meets one and the host reruns that forward on the bf16
// three-term kernels (rr_ce.hip: ce_gemm_x3 / ce_attention_x3, any fp32 range).
//
// Operands travel as "h2" images (rr_ce_h2.h): two fp16 planes in 16-byte units of eight k, chunk-major, written by the
// PRODUCER's epilogue (LayerNorm, GELU, attention) -- a value is split once, not once per column block that reads it, and an h2
// image is exactly as large as the fp32 matrix.  For ce_gemm_h2 that layout makes a 64-row x 8-k piece of a tile 1 KiB that
// is contiguous in HBM and in LDS: the tiles are staged by LDS-DMA (global_load_lds_dwordx4, no registers, no ds_write), three
// K steps deep, and an MFMA operand fragment is one conflict-free ds_read_b128.
#include "rr_ce_h2.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "rr_common.h"

typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_f16x4 __attribute__((ext_vector_type(4)));
typedef float h2_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int h2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int h2_u32x2 __attribute__((ext_vector_type(2)));
typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));

#define H2_H 384

struct h2_pair { _Float16 hi, lo; };
__device__ __forceinline__ h2_pair h2_split(float x) {
    h2_pair r;
    // nearest even.  No special case below 2^-14: the vector unit and the matrix core both take SUBNORMAL fp16 values as they
    // are (tools/probes/f16_flush_probe.hip: entries log-uniform in [1e-7, 1e-3], error 1.3e-7 of sum |a b|), so hi + lo / 2048
    // holds 22 bits there too; sending such a value to lo alone would keep eleven
    r.hi = (_Float16)x;
    r.lo = (_Float16)((x - (float)r.hi) * CE_H2_SCALE);         // (exact difference; |.| * 2048 <= |x|)
    return r;
}
__device__ __forceinline__ bool h2_out_of_range(float x) { return !(__builtin_fabsf(x) <= 65504.f); }    // (NaN too)
__device__ __forceinline__ void h2_raise(bool bad, unsigned* flag) {
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
__device__ __forceinline__ float h2_vmax(float a, float b) {      // (no canonicalising v_max x, x in front: the inputs are MFMA sums)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float h2_vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float h2_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------ fp32 rows -> h2 (weights, at load)
__global__ __launch_bounds__(256) void ce_h2_pack_kernel(const float* __restrict__ src, int R, int K, h2_u32x4* __restrict__ dst,
                                                         int64_t rs) {
    const int KC = K >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // unit (kc, row), rows fastest
    if (i >= (int64_t)R * KC) return;
    const int row = (int)(i % R), kc = (int)(i / R);
    const float* p = src + (int64_t)row * K + 8 * kc;
    h2_f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const h2_pair s = h2_split(p[e]);
        hi[e] = s.hi;
        lo[e] = s.lo;
    }
    dst[(int64_t)kc * rs + row] = __builtin_bit_cast(h2_u32x4, hi);
    dst[((int64_t)KC + kc) * rs + row] = __builtin_bit_cast(h2_u32x4, lo);
}

void ce_h2_pack(const float* d_src, int R, int K, void* d_dst, int64_t row_stride, hipStream_t st) {
    const int64_t n = (int64_t)R * (K / 8);
    hipLaunchKernelGGL(ce_h2_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_src, R, K, (h2_u32x4*)d_dst, row_stride);
}

// ------------------------------------------------------------------ LayerNorm producers
// Workgroup = 32 tokens, a wave takes eight of them one after the other (lane c < 48 = chunk c: 32 bytes of the fp32 row).
// The h2 units of the 32 tokens are gathered in LDS ([plane][chunk][token], token rows padded to 33 units) and leave as 512-byte
// runs of 32 consecutive tokens per (plane, chunk): written straight from the lanes they would be 16-byte pieces 16 * xs bytes
// apart (measured: 287 us per launch at 131 072 tokens against 100 for this form).
#define H2LN_TOK 32
#define H2LN_LD 33
template <bool EMBED>
__global__ __launch_bounds__(256) void ce_h2_ln_kernel(const int32_t* __restrict__ tok, const int32_t* __restrict__ typ,
                                                       const int32_t* __restrict__ pos, int T, int vocab, int n_pos, int n_typ,
                                                       const float* __restrict__ we, const float* __restrict__ pe,
                                                       const float* __restrict__ te, const float* __restrict__ y,
                                                       const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                       float* __restrict__ h32, h2_u32x4* __restrict__ hx, int64_t xs,
                                                       unsigned* __restrict__ flag) {
    __shared__ h2_u32x4 img[2 * 48 * H2LN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tb = blockIdx.x * H2LN_TOK;
    const bool on = lane < 48;
    const int c8 = 8 * (on ? lane : 0);
    h2_f32x4 gg[2], bb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gg[h] = *reinterpret_cast<const h2_f32x4*>(g + c8 + 4 * h);
        bb[h] = *reinterpret_cast<const h2_f32x4*>(b + c8 + 4 * h);
    }
    bool bad = false;
    for (int i8 = 0; i8 < 8; ++i8) {
        const int tl = 8 * wave + i8, t = tb + tl;
        if (t >= T) break;                                   // (wave-uniform)
        float x[8];
        if (EMBED) {
            int id = tok[t], ty = typ[t], po = pos[t];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // ids are validated on the host; stay in bounds anyway
            ty = ty < 0 ? 0 : (ty >= n_typ ? n_typ - 1 : ty);
            po = po < 0 ? 0 : (po >= n_pos ? n_pos - 1 : po);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const h2_f32x4 a = *reinterpret_cast<const h2_f32x4*>(we + (int64_t)id * H2_H + c8 + 4 * h);
                const h2_f32x4 d = *reinterpret_cast<const h2_f32x4*>(te + (int64_t)ty * H2_H + c8 + 4 * h);
                const h2_f32x4 e = *reinterpret_cast<const h2_f32x4*>(pe + (int64_t)po * H2_H + c8 + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[4 * h + i] = (a[i] + d[i]) + e[i];      // (the order of BertEmbeddings: word + type, + position)
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const h2_f32x4 a = *reinterpret_cast<const h2_f32x4*>(y + (int64_t)t * H2_H + c8 + 4 * h);
                const h2_f32x4 d = *reinterpret_cast<const h2_f32x4*>(h32 + (int64_t)t * H2_H + c8 + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[4 * h + i] = a[i] + d[i];
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += on ? x[i] : 0.f;
        const float mean = h2_wave_sum(s) * (1.f / H2_H);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = x[i] - mean; v += on ? d * d : 0.f; }
        const float var = h2_wave_sum(v) * (1.f / H2_H) + eps;
        const float rstd = EMBED ? rsqrtf(var) : 1.0f / sqrtf(var);     // (as ce_embed_ln / ce_add_ln_f32, rr_ce.hip)
        h2_f16x8 hi, lo;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float r = (x[i] - mean) * rstd * gg[i >> 2][i & 3] + bb[i >> 2][i & 3];
            o[i] = r;
            const h2_pair p = h2_split(r);
            hi[i] = p.hi;
            lo[i] = p.lo;
            bad |= h2_out_of_range(r);
        }
        if (on) {
            *reinterpret_cast<h2_f32x4*>(h32 + (int64_t)t * H2_H + c8) = h2_f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<h2_f32x4*>(h32 + (int64_t)t * H2_H + c8 + 4) = h2_f32x4{o[4], o[5], o[6], o[7]};
            img[lane * H2LN_LD + tl] = __builtin_bit_cast(h2_u32x4, hi);
            img[(48 + lane) * H2LN_LD + tl] = __builtin_bit_cast(h2_u32x4, lo);
        }
    }
    h2_raise(on && bad, flag);
    __syncthreads();
    for (int i = tid; i < 2 * 48 * H2LN_TOK; i += 256) {
        const int pc = i >> 5, tl = i & 31;
        if (tb + tl < T) hx[(int64_t)pc * xs + tb + tl] = img[pc * H2LN_LD + tl];
    }
}

void ce_h2_embed_ln(const int32_t* tok, const int32_t* typ, const int32_t* pos, int T, int vocab, int n_pos, int n_typ,
                    const float* we, const float* pe, const float* te, const float* g, const float* b, float eps, float* h32,
                    void* hx, int64_t xs, unsigned* flag, hipStream_t st) {
    hipLaunchKernelGGL((ce_h2_ln_kernel<true>), dim3((unsigned)((T + H2LN_TOK - 1) / H2LN_TOK)), dim3(256), 0, st, tok, typ, pos, T, vocab, n_pos,
                       n_typ, we, pe, te, (const float*)nullptr, g, b, eps, h32, (h2_u32x4*)hx, xs, flag);
}

void ce_h2_add_ln(const float* y, float* h32, int T, const float* g, const float* b, float eps, void* hx, int64_t xs,
                  unsigned* flag, hipStream_t st) {
    hipLaunchKernelGGL((ce_h2_ln_kernel<false>), dim3((unsigned)((T + H2LN_TOK - 1) / H2LN_TOK)), dim3(256), 0, st, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, T, 0, 0, 0, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, y, g, b, eps, h32, (h2_u32x4*)hx, xs, flag);
}

// ------------------------------------------------------------------ GEMM
// Workgroup = 8 waves, tile = 128 features x 256 tokens, wave (wf = w & 1, wt = w >> 1) = 64 features x 64 tokens = 4 x 4
// MFMA tiles, two accumulators (128 registers); two waves per SIMD, one workgroup per CU.  The MFMA computes the TRANSPOSED
// tile, D[feature][token] = W X^T (A operand = the weights): a lane then holds four consecutive FEATURES of one token --
// 16 bytes of an fp32 row, or the 8-byte half of an h2 unit.
// K step = 32 = one MFMA depth; per step a stage of LDS holds W [plane][chunk][128 rows] (16 KB) and X [plane][chunk][256 rows]
// (32 KB) in units: 48 LDS-DMA pieces of 64 rows, six per wave.  Three stages: the pieces of step s + 2 are issued right
// after the barrier of step s (every wave has read stage s - 1 by then), `s_waitcnt vmcnt(6)` + the barrier make stage s
// visible.  Fragment of lane (r = l & 15, kq = l >> 4): unit [plane][kq][tile row r]: a ds_read_b128 is served in four groups
// of 16 lanes that hold 16 different rows r each (MI355X_MICROARCH.md, LDS) = 16 different 16-byte bank groups: no conflict.
// Grid: 1-D, XCD-aware -- workgroup id % 8 is the XCD; the feature blocks of one token block run on ONE XCD back to back, so
// the token tile is fetched from HBM once and re-read from that XCD's L2.
#define H2G_BT 256
#define H2G_BF 128
#define H2G_STAGE_UNITS 3072
#define H2G_LDS (3 * H2G_STAGE_UNITS * 16)

#define H2_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

// gelu(x) = x Phi(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2), erfc(t) = 2^(t P(t)) with P of degree 7 (tools/fit_gelu_f32.py:
// weighted minimax fit on [0, 4.25], the argument clamped there -- erfc(4.25) = 1.8e-9): no branch, one v_exp_f32, 13 vector
// operations against ~40 for the library's erff, which made the FFN's first GEMM vector-bound.  In fp32 with this operation
// order: |error| <= 2.5e-7 (8.9e-8 of max(1, |x|)) against the float64 value, where torch's own fp32 gelu sits at 1.2e-6.
// Two values at a time: the Horner steps, the products and the last step on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32).
__device__ __forceinline__ h2_f32x2 h2_gelu2(h2_f32x2 x) {
    const h2_f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
    const h2_f32x2 t = {__builtin_fminf(a[0] * 0.70710678118654752440f, 4.25f), __builtin_fminf(a[1] * 0.70710678118654752440f, 4.25f)};
#define H2_C2(c) (h2_f32x2{c, c})
    h2_f32x2 p = H2_C2(-3.043266588e-05f);
    p = __builtin_elementwise_fma(p, t, H2_C2(3.174242738e-04f));
    p = __builtin_elementwise_fma(p, t, H2_C2(-1.051770989e-03f));
    p = __builtin_elementwise_fma(p, t, H2_C2(-1.539122313e-03f));
    p = __builtin_elementwise_fma(p, t, H2_C2(2.898171730e-02f));
    p = __builtin_elementwise_fma(p, t, H2_C2(-1.488533765e-01f));
    p = __builtin_elementwise_fma(p, t, H2_C2(-9.183242917e-01f));
    p = __builtin_elementwise_fma(p, t, H2_C2(-1.627916813e+00f));
    const h2_f32x2 q = p * t;
    const h2_f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const h2_f32x2 m = {__builtin_fmaxf(x[0], 0.f), __builtin_fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(a * H2_C2(-0.5f), e, m);
#undef H2_C2
}
// two values -> (hi, lo) pairs, packed conversions (v_cvt_pk_f16_f32)
__device__ __forceinline__ void h2_split2(h2_f32x2 x, h2_f16x2& hi, h2_f16x2& lo) {
    hi = __builtin_convertvector(x, h2_f16x2);
    lo = __builtin_convertvector((x - __builtin_convertvector(hi, h2_f32x2)) * h2_f32x2{CE_H2_SCALE, CE_H2_SCALE}, h2_f16x2);
}

#ifdef RR_DEBUG_HARNESS
// tools/k5_h2_stamps.py: phase clocks of waves 0 and 4 of workgroup 2048 of the LAST ce_gemm_h2 launch with K = 384 and
// N = 1536: [prologue, wait + barrier, data movement, reads until the first MFMA may start, MFMA issue, epilogue, steps,
// wall clock (100 MHz)] in shader cycles
__device__ unsigned long long h2_dbg[2][8];
#define H2_STAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#define H2_DBG(...) __VA_ARGS__
#else
#define H2_STAMP(x)
#define H2_DBG(...)
#endif
// Staging: LDS-DMA (global_load_lds_dwordx4), three LDS stages, no registers.  Two other forms were built and measured
// (profiles/r04_k5_h2_gemm_stamps.txt, tools/k5_h2_stamps.py): global_load_dwordx4 one step ahead + ds_write_b128 with two LDS
// stages -- 4 - 10 % slower (QKV 445 vs 405 us at 131 072 tokens) --, and 128-token tiles with two workgroups per CU -- 5 - 8 %
// slower.
// A step: `s_waitcnt vmcnt(6)` + the barrier make stage s visible (the pieces of stage s + 1 may still fly); every wave issues
// its sixteen fragment reads; the SIMD's first wave (w < 4) issues its six pieces of stage s + 2 under their latency, the
// second one behind its MFMAs (RR_CE_H2_NO_STAGGER=1: both up front); 16 MFMAs hi x hi (acc1), 16 hi x lo + 16 lo x hi (acc2).
// In-kernel clocks (FFN1 shape, 1.5 GHz shader clock): the SIMD's two waves share one matrix pipe, 2 x 48 MFMAs = 1 536 cycles,
// and a step takes ~2 330 -- 300 - 500 until the first MFMA (eight waves read 128 KB of LDS at once), the pieces' issue
// 190 - 300, ~150 at the barrier.  The epilogue (GELU + split + stores: ~11 000 cycles of vector work for the SIMD's two
// waves) runs with the matrix pipe idle: 27 % of an FFN1 tile.  Also built and measured slower: the next stage's high-plane
// fragments read one step ahead (every stage must then land within ONE step: 13.8 vs 13.1 ms per forward); and a small-tile
// form -- four waves, 128 x 128, 32x32x16 MFMAs, K step 16, four 16 KB stages, TWO independent workgroups per CU so that one
// multiplies while the other reads, waits at its barrier or runs its epilogue: 13.6 ms (FFN-1 626 vs 613 us, QKV 435 vs 405).
// That every re-cut lands within a few per cent says the phases are not what limits the kernel: the shader clock under it is
// 1.5 GHz (stamps) of 2.4, i.e. the matrix peak at the clock it holds is 1.57 PF and FFN-2 runs at 1.1 PF, QKV at 0.85, FFN-1
// (with its GELU) at 0.75 -- the power-bound regime rr_scan_fltq sits in, where cycles taken out of one
// phase come back as a lower clock.  Less ENERGY per product (fewer LDS bytes per MFMA: wider wave tiles) is what would move it.
template <int EPI>
__global__ __launch_bounds__(512, 1) void ce_gemm_h2(const h2_u32x4* __restrict__ W2, int N, const h2_u32x4* __restrict__ X2, int64_t xs, int M,
                                                     int K, const float* __restrict__ bias, float* __restrict__ out32,
                                                     h2_u32x2* __restrict__ out2, int64_t os, unsigned* __restrict__ flag, float qscale,
                                                     int qcols, int stagger) {
    extern __shared__ __attribute__((aligned(16))) h2_u32x4 h2g_lds[];
    constexpr int TB = H2G_BT;
    constexpr int PW = 6;                                  // 1 KB pieces per wave and stage: 48 / 8
    constexpr int STAGE_UNITS = H2G_STAGE_UNITS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nf = N / H2G_BF;
    const int id = blockIdx.x, slot = id >> 3;
    const int tb = (slot / nf) * 8 + (id & 7), fb = slot % nf;
    const int m0 = tb * TB, n0 = fb * H2G_BF;
    if (m0 >= M) return;                                   // (whole workgroup: the grid is padded to eight token blocks)
    const int KC = K >> 3, KS = K >> 5;

    // ---- the pieces of this wave: g = 6 w + j; g < 16: W piece (plane-chunk g >> 1, row half g & 1), else X piece
    const h2_u32x4* src[PW];
    int64_t step[PW];
    int dst[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int g = PW * w + j;
        if (g < 16) {
            const int pc = g >> 1, half = g & 1;
            src[j] = W2 + ((int64_t)((pc >> 2) * KC + (pc & 3)) * N + n0 + 64 * half + lane);
            step[j] = 4 * (int64_t)N;
            dst[j] = pc * 128 + 64 * half;
        } else {
            const int pc = (g - 16) >> 2, q = (g - 16) & 3;
            src[j] = X2 + ((int64_t)((pc >> 2) * KC + (pc & 3)) * xs + m0 + 64 * q + lane);
            step[j] = 4 * xs;
            dst[j] = 1024 + pc * TB + 64 * q;
        }
    }
    auto issue = [&](int s, int buf) {                     // stage s -> LDS buffer buf
#pragma unroll
        for (int j = 0; j < PW; ++j)
            __builtin_amdgcn_global_load_lds(src[j] + s * step[j],
                                             (__attribute__((address_space(3))) void*)(h2g_lds + buf * STAGE_UNITS + dst[j]), 16, 0, 0);
    };
    const int r = lane & 15, kq = lane >> 4, wf = w & 1, wt = w >> 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)h2g_lds;
    const uint32_t aW = lds0 + 16u * (uint32_t)(kq * 128 + 64 * wf + r);
    const uint32_t aX = lds0 + 16u * (uint32_t)(1024 + kq * TB + 64 * wt + r);
    constexpr int XLO = 64 * TB;                           // byte offset of the lo plane inside the X part: 4 chunks x TB rows x 16

    h2_f32x4 acc1[4][4], acc2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc1[i][j] = h2_f32x4{0.f, 0.f, 0.f, 0.f};
            acc2[i][j] = h2_f32x4{0.f, 0.f, 0.f, 0.f};
        }

    H2_DBG(unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long wall0 = __builtin_amdgcn_s_memrealtime();)
    H2_STAMP(tp0);
    // the SIMD's second wave (w >= 4) issues its pieces behind its MFMAs, the first one under the latency of its reads
    const bool late = stagger && w >= 4;
    issue(0, 0);
    if (KS > 1) issue(1, 1);
    int buf = 0;
    H2_STAMP(tp1);
    H2_DBG(dbg[0] = tp1 - tp0;)
#define H2_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h2_f16x8, A), __builtin_bit_cast(h2_f16x8, B), C, 0, 0, 0)
    for (int s = 0; s < KS; ++s) {
        H2_STAMP(t0);
        // stage s has landed: the only pieces that may still fly are those of stage s + 1
        if (s + 1 < KS) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        H2_STAMP(t1);
        const uint32_t bW = aW + (uint32_t)buf * (STAGE_UNITS * 16), bX = aX + (uint32_t)buf * (STAGE_UNITS * 16);
        h2_u32x4 wh[4], wl[4], xh[4], xl[4];
        // (X hi and the first W hi fragment first: four MFMAs can start once five of the sixteen fragments are there)
        H2_READ(xh[0], bX, 0); H2_READ(xh[1], bX, 256); H2_READ(xh[2], bX, 512); H2_READ(xh[3], bX, 768);
        H2_READ(wh[0], bW, 0); H2_READ(wh[1], bW, 256); H2_READ(wh[2], bW, 512); H2_READ(wh[3], bW, 768);
        H2_READ(xl[0], bX, XLO); H2_READ(xl[1], bX, XLO + 256); H2_READ(xl[2], bX, XLO + 512); H2_READ(xl[3], bX, XLO + 768);
        H2_READ(wl[0], bW, 8192); H2_READ(wl[1], bW, 8192 + 256); H2_READ(wl[2], bW, 8192 + 512); H2_READ(wl[3], bW, 8192 + 768);
        __builtin_amdgcn_sched_barrier(0);
        H2_STAMP(t2);
        // the pieces of stage s + 2 go into the buffer of the stage multiplied in step s - 1 (every wave is past it)
        if (!late && s + 2 < KS) issue(s + 2, buf == 0 ? 2 : buf - 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(wh[0]), "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]) :: "memory");
        H2_DBG(__builtin_amdgcn_sched_barrier(0);)
        H2_STAMP(t3);
        H2_DBG(__builtin_amdgcn_sched_barrier(0);)
#pragma unroll
        for (int j = 0; j < 4; ++j) H2_MFMA(wh[0], xh[j], acc1[0][j]);
        asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(wh[1]) :: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) H2_MFMA(wh[1], xh[j], acc1[1][j]);
        asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(wh[2]) :: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) H2_MFMA(wh[2], xh[j], acc1[2][j]);
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(wh[3]) :: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) H2_MFMA(wh[3], xh[j], acc1[3][j]);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]) :: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) H2_MFMA(wh[i], xl[j], acc2[i][j]);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wl[0]), "+v"(wl[1]), "+v"(wl[2]), "+v"(wl[3]) :: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) H2_MFMA(wl[i], xh[j], acc2[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        H2_STAMP(t4);
        if (late && s + 2 < KS) issue(s + 2, buf == 0 ? 2 : buf - 1);
        H2_STAMP(t5);
        H2_DBG(dbg[1] += t1 - t0; dbg[2] += t5 - t4; dbg[3] += t3 - t1; dbg[4] += t4 - t3; dbg[6] += 1;)
        buf = buf == 2 ? 0 : buf + 1;
    }
    H2_STAMP(te0);

    // ---- epilogue (pairs of values on the packed fp32 pipe; 32-bit byte offsets from the output's base)
    float amax = 0.f;                                      // max |value split to fp16|: one range test per lane at the end
    {
        // acc[i][j][e]: feature n0 + 64 wf + 16 i + 4 kq + e, token m0 + 64 wt + 16 j + r
        const uint32_t os32 = (uint32_t)os, lo_plane = (uint32_t)(N >> 3) * os32 * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = n0 + 64 * wf + 16 * i + 4 * kq;
            const h2_f32x4 bv = *reinterpret_cast<const h2_f32x4*>(bias + f);
            const h2_f32x2 b01 = {bv[0], bv[1]}, b23 = {bv[2], bv[3]}, c2 = {CE_H2_INV_SCALE, CE_H2_INV_SCALE};
            const bool scaled = EPI == CE_H2_EPI_H2 && f < qcols;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = m0 + 64 * wt + 16 * j + r;
                h2_f32x2 v01 = __builtin_elementwise_fma(h2_f32x2{acc2[i][j][0], acc2[i][j][1]}, c2, h2_f32x2{acc1[i][j][0], acc1[i][j][1]}) + b01;
                h2_f32x2 v23 = __builtin_elementwise_fma(h2_f32x2{acc2[i][j][2], acc2[i][j][3]}, c2, h2_f32x2{acc1[i][j][2], acc1[i][j][3]}) + b23;
                if (EPI == CE_H2_EPI_GELU_H2) {
                    v01 = h2_gelu2(v01);
                    v23 = h2_gelu2(v23);
                }
                if (scaled) {
                    v01 *= h2_f32x2{qscale, qscale};
                    v23 *= h2_f32x2{qscale, qscale};
                }
                if (t >= M) continue;
                if (EPI == CE_H2_EPI_F32) {
                    *reinterpret_cast<h2_f32x4*>(out32 + (int64_t)t * N + f) = h2_f32x4{v01[0], v01[1], v23[0], v23[1]};
                } else {
                    h2_f16x2 h01, l01, h23, l23;
                    h2_split2(v01, h01, l01);
                    h2_split2(v23, h23, l23);
                    amax = h2_vmax3(amax, __builtin_fabsf(v01[0]), __builtin_fabsf(v01[1]));
                    amax = h2_vmax3(amax, __builtin_fabsf(v23[0]), __builtin_fabsf(v23[1]));
                    // unit (chunk f / 8, token), half kq & 1
                    const uint32_t off = ((uint32_t)(f >> 3) * os32 + (uint32_t)t) * 16u + 8u * (uint32_t)(kq & 1);
                    *reinterpret_cast<h2_u32x2*>((char*)out2 + off) = h2_u32x2{__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23)};
                    *reinterpret_cast<h2_u32x2*>((char*)out2 + lo_plane + off) = h2_u32x2{__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23)};
                }
            }
        }
    }
    const bool bad = !(amax <= 65504.f);                   // (a NaN never raises amax: it comes out of an overflow flagged before it)
    if (EPI != CE_H2_EPI_F32) h2_raise(bad, flag);
#ifdef RR_DEBUG_HARNESS
    if (blockIdx.x == 2048 && K == 384 && N == 1536 && (w == 0 || w == 4) && lane == 0) {
        dbg[5] = __builtin_amdgcn_s_memtime() - te0;
        dbg[7] = __builtin_amdgcn_s_memrealtime() - wall0;
        for (int i = 0; i < 8; ++i) h2_dbg[w >> 2][i] = dbg[i];
    }
#endif
}

void ce_h2_gemm(int epi, const void* W2, int N, const void* X2, int64_t xs, int M, int K, const float* bias, float* out32,
                void* out2, int64_t os, unsigned* flag, hipStream_t st, float qscale, int qcols) {
    static const int stagger = getenv("RR_CE_H2_NO_STAGGER") == nullptr;      // (A/B)
    const int tbs = (M + H2G_BT - 1) / H2G_BT;
    const dim3 grid((unsigned)(((tbs + 7) / 8) * 8 * (N / H2G_BF)));
#define H2_LAUNCH(E) \
    hipLaunchKernelGGL((ce_gemm_h2<E>), grid, dim3(512), H2G_LDS, st, (const h2_u32x4*)W2, N, (const h2_u32x4*)X2, xs, M, K, bias, out32, \
                       (h2_u32x2*)out2, os, flag, qscale, qcols, stagger)
    if (epi == CE_H2_EPI_F32) H2_LAUNCH(CE_H2_EPI_F32);
    else if (epi == CE_H2_EPI_H2) H2_LAUNCH(CE_H2_EPI_H2);
    else H2_LAUNCH(CE_H2_EPI_GELU_H2);
#undef H2_LAUNCH
}

// ------------------------------------------------------------------ attention
// One workgroup (eight waves) per (sequence, head), the sequence's keys in groups of 32.  Q (scaled by log2 e / sqrt 32 in the
// projection's epilogue), K and V arrive in ONE h2 image of [T][1152] (chunks 4 head .. + 3 of the Q / K / V thirds): per
// (plane, chunk) a token is one 16-byte unit.  K and V of every group go to LDS by LDS-DMA at once (8 KB per group, <= 16
// groups = 128 KB, one workgroup per CU) into the SAME image form, [plane][key][4 slots] with the chunk of key k in slot
// chunk ^ 2 ((k >> 2) & 1) -- the DMA lays lanes out linearly in LDS, so the swizzle is applied to the global address a lane
// fetches.  That one image serves both products (cdna_hip_programming.md T10, "one image for row reads and transposed reads"):
//   S^T[key][query] = K Q^T on v_mfma_f32_16x16x32_f16 (K = 32 = the head dimension: one MFMA per term and 16 x 16 tile): the A
//     operand of lane (key r, chunk kq) is one ds_read_b128, conflict-free under the swizzle;
//     s = s1 + s2 / 2048, s1 = Khi Qhi, s2 = Khi Qlo + Klo Qhi;
//   online softmax in base 2; a lane holds the 8 scores of ITS query against keys 16 a + 4 (l >> 4) + e (a = 0, 1), the row
//     maximum is two v_permlane swaps away; p = 2^(s - m + 15) -- the 2^15 keeps a probability down to 2e-9 a normal fp16 and
//     one down to 2e-12 a subnormal one with its rest in lo (fp32's own rounding of the row is 6e-8); it cancels in o / l;
//   O^T[dim][query] += V^T P^T: the lane's 8 probabilities ARE the B operand's k slots 8 (l >> 4) + 4 a + e, so P never leaves
//     its registers, and the A operand -- dim (l & 15) against exactly those keys -- is two ds_read_b64_tr_b16 of the row-major
//     V image (the hardware transpose: 4 keys x 16 dims per 16 lanes), conflict-free under the same swizzle;
//     o1 += Vhi Phi, o2 += Vhi Plo + Vlo Phi.
// The context leaves as an h2 image: a lane holds four consecutive dims of its query = one 8-byte half unit.  Nothing depends on
// where the sequence sits in the pack: a sequence's scores are bit for bit the same alone or in any batch.
#define H2A_GROUP_UNITS 512
#define H2A_MAX_GROUPS 16
#define H2A_LDS (H2A_MAX_GROUPS * H2A_GROUP_UNITS * 16)
#define H2A_CHUNKS 144                                          // 16-byte units per token and plane: 1152 / 8
typedef __fp16 h2_tr4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// NT query tiles of one wave (tiles qt0, qt0 + TSTEP, ...) against every key group, their chains in one instruction stream (650 us
// per layer at 131 072 tokens with one tile per pass, 590 with four; the K / V fragments are read once for the NT tiles).
// The kernel is bound by the vector unit's ISSUE: ~93 vector instructions per 16 x 32 score tile (PMC: 143 k per SIMD and
// launch, ~4.4 cycles each with exp2 at 8) plus 8 of every MFMA's 16 cycles, in which the SIMD issues no vector instruction
// (MI355X_MICROARCH.md) -- 630 k + 145 k of the 885 k cycles a launch takes.  A form that issues the next key group's K Q^T
// MFMAs inside the exp2 / split stretch of the current one (software-pipelined by one group, two tiles per pass) was built
// and ran at the same speed, and so did the split of P by v_fma_mixlo / mixhi_f16 (four vector instructions per pair instead
// of six, the same bits): the instruction count of one phase is not what sets this kernel's time either.
template <int NT, int TSTEP = 8>
__device__ __forceinline__ void h2a_tiles(const h2_u32x4* __restrict__ qkv, int64_t xs, int t0, int S, int ng, int head, int qt0, int n_tiles,
                                          const h2_u32x4* kimg, const char* vimg, int vdi, int r, int kq, h2_u32x2* __restrict__ ctx2,
                                          int64_t os, int64_t cls_row, bool& bad) {
    h2_f16x8 bqh[NT], bql[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int q = 16 * (qt0 + TSTEP * t) + r;
        q = q < S ? q : S - 1;                            // (lanes / tiles past the sequence: a valid row, never stored)
        const h2_u32x4* p = qkv + ((int64_t)(head * 4 + kq) * xs + t0 + q);
        bqh[t] = __builtin_bit_cast(h2_f16x8, p[0]);
        bql[t] = __builtin_bit_cast(h2_f16x8, p[H2A_CHUNKS * xs]);
    }
    h2_f32x4 o1[NT][2], o2[NT][2];
    float mm[NT];                                          // running maximum - 15
    h2_f32x2 lsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            o1[t][i] = h2_f32x4{0.f, 0.f, 0.f, 0.f};
            o2[t][i] = h2_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        mm[t] = -INFINITY;
        lsum[t] = h2_f32x2{0.f, 0.f};
    }
    // one key group; MASKED = the sequence's last, partial group (its own copy of the body: no mask work, and no branch between
    // the four tiles' chains, in the groups before it)
    auto group = [&](int gi, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
        const h2_u32x4* kb = kimg + gi * H2A_GROUP_UNITS;
        h2_f16x8 kh[2], kl[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            kh[a] = __builtin_bit_cast(h2_f16x8, kb[64 * a]);
            kl[a] = __builtin_bit_cast(h2_f16x8, kb[128 + 64 * a]);
        }
        h2_f32x2 s[NT][4];                                 // [a][e pair]: keys 16 a + 4 kq + 2 j, + 1
        float cm[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const h2_f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const h2_f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[a], bqh[t], z, 0, 0, 0);
                h2_f32x4 s2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[a], bql[t], z, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[a], bqh[t], s2, 0, 0, 0);
                const h2_f32x2 c = {CE_H2_INV_SCALE, CE_H2_INV_SCALE};
                s[t][2 * a] = __builtin_elementwise_fma(h2_f32x2{s2[0], s2[1]}, c, h2_f32x2{s1[0], s1[1]});
                s[t][2 * a + 1] = __builtin_elementwise_fma(h2_f32x2{s2[2], s2[3]}, c, h2_f32x2{s1[2], s1[3]});
            }
            if (MASKED) {
                const int key0 = 32 * gi + 4 * kq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = key0 + 16 * (j >> 1) + 2 * (j & 1);
                    s[t][j][0] = key < S ? s[t][j][0] : -INFINITY;
                    s[t][j][1] = key + 1 < S ? s[t][j][1] : -INFINITY;
                }
            }
            float m8 = h2_vmax3(h2_vmax3(s[t][0][0], s[t][0][1], s[t][1][0]), h2_vmax3(s[t][1][1], s[t][2][0], s[t][2][1]), h2_vmax(s[t][3][0], s[t][3][1]));
            const auto x16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(m8), __float_as_uint(m8), false, false);
            m8 = h2_vmax(__uint_as_float(x16[0]), __uint_as_float(x16[1]));
            const auto x32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m8), __float_as_uint(m8), false, false);
            cm[t] = h2_vmax(__uint_as_float(x32[0]), __uint_as_float(x32[1])) - 15.f;
        }
        bool grow = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) grow |= cm[t] > mm[t];
        if (__any(grow)) {                                 // (wave-uniform; after the first groups a new maximum is rare)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float nm = h2_vmax(mm[t], cm[t]);
                const float f = __builtin_amdgcn_exp2f(mm[t] - nm);      // first group: 2^-inf = 0; no change: 1
                lsum[t] *= f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    o1[t][i] *= f;
                    o2[t][i] *= f;
                }
                mm[t] = nm;
            }
        }
        h2_f16x8 ph[NT], pl[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h2_f32x2 m2 = {mm[t], mm[t]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2_f32x2 d = s[t][j] - m2;
                const h2_f32x2 pe = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};      // masked keys: 2^-inf = 0
                lsum[t] += pe;
                h2_f16x2 h2v, l2v;
                h2_split2(pe, h2v, l2v);
                ph[t][2 * j] = h2v[0];
                ph[t][2 * j + 1] = h2v[1];
                pl[t][2 * j] = l2v[0];
                pl[t][2 * j + 1] = l2v[1];
            }
        }
        const char* vb = vimg + gi * (H2A_GROUP_UNITS * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* vi = vb + i * vdi;
            h2_tr4 tr[2][2];
#pragma unroll
            for (int pln = 0; pln < 2; ++pln)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    tr[pln][a] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h2_tr4*)(vi + pln * 2048 + a * 1024));
            h2_f16x8 vh, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vh[e] = (_Float16)tr[0][0][e]; vh[4 + e] = (_Float16)tr[0][1][e];
                vl[e] = (_Float16)tr[1][0][e]; vl[4 + e] = (_Float16)tr[1][1][e];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                o1[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[t], o1[t][i], 0, 0, 0);
                o2[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[t], o2[t][i], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) o2[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[t], o2[t][i], 0, 0, 0);
        }
    };
    const int ng_full = S >> 5;
    for (int gi = 0; gi < ng_full; ++gi) group(gi, std::false_type{});
    if (ng_full < ng) group(ng_full, std::true_type{});
    // ---- the row sum over the query's four lanes, normalise, split, store: o[i][e] = dim 16 i + 4 kq + e of query 16 qt + r
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float l = lsum[t][0] + lsum[t][1];
        const auto x16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l), __float_as_uint(l), false, false);
        l = __uint_as_float(x16[0]) + __uint_as_float(x16[1]);
        const auto x32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
        l = __uint_as_float(x32[0]) + __uint_as_float(x32[1]);
        const int qt = qt0 + TSTEP * t, q = 16 * qt + r;
        if (qt < n_tiles && q < S && (cls_row < 0 || q == 0)) {
            const float inv = 1.0f / l;
            const int64_t row = cls_row >= 0 ? cls_row : t0 + q;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                h2_f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = __builtin_fmaf(o2[t][i][e], CE_H2_INV_SCALE, o1[t][i][e]) * inv;
                    const h2_pair pr = h2_split(v);
                    hi[e] = pr.hi;
                    lo[e] = pr.lo;
                    bad |= h2_out_of_range(v);
                }
                const int64_t u = (int64_t)(head * 4 + 2 * i + (kq >> 1)) * os + row;
                ctx2[2 * u + (kq & 1)] = __builtin_bit_cast(h2_u32x2, hi);
                ctx2[2 * ((int64_t)(H2_H / 8) * os + u) + (kq & 1)] = __builtin_bit_cast(h2_u32x2, lo);
            }
        }
    }
}

__global__ __launch_bounds__(512, 1) void ce_attention_h2(const h2_u32x4* __restrict__ qkv, int64_t xs, const int32_t* __restrict__ cu,
                                                          h2_u32x2* __restrict__ ctx2, int64_t os, unsigned* __restrict__ flag,
                                                          int cls_only) {
    extern __shared__ __attribute__((aligned(16))) h2_u32x4 h2a_lds[];
    const int seq = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = cu[seq], S = cu[seq + 1] - t0;
    const int ng = (S + 31) >> 5;
    // ---- staging: wave w = (K | V: w >> 2, plane (w >> 1) & 1, key half w & 1) of every group; lane = (key, slot)
    {
        const int which = w >> 2, p = (w >> 1) & 1, key = 16 * (w & 1) + (lane >> 2), slot = lane & 3;
        const int chunk = slot ^ (2 * ((key >> 2) & 1));
        const h2_u32x4* plane = qkv + (int64_t)(p * H2A_CHUNKS + 48 * (1 + which) + head * 4 + chunk) * xs;
        const int dst = which * 256 + p * 128 + 64 * (w & 1);
        for (int gi = 0; gi < ng; ++gi) {
            int k = 32 * gi + key;
            k = k < S ? k : S - 1;                         // (keys past the sequence: a valid row; masked / probability 0)
            __builtin_amdgcn_global_load_lds(plane + t0 + k, (__attribute__((address_space(3))) void*)(h2a_lds + gi * H2A_GROUP_UNITS + dst), 16, 0, 0);
        }
    }
    const int r = lane & 15, kq = lane >> 4;
    const int n_tiles = cls_only ? 1 : (S + 15) >> 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // K row reads: lane (key 16 a + r, chunk kq): unit plane * 128 + key * 4 + (kq ^ 2 ((r >> 2) & 1))
    const h2_u32x4* kimg = h2a_lds + r * 4 + (kq ^ (2 * ((r >> 2) & 1)));
    // V transposed reads: lane (q = (l >> 2) & 3, pp = l & 3) of group kq addresses key 16 a + 4 kq + q, dims 16 i + 4 pp .. + 3:
    // chunk 2 i + (pp >> 1) in slot chunk ^ 2 (kq & 1), byte half pp & 1
    const int vq = (lane >> 2) & 3, vpp = lane & 3;
    const char* vimg = (const char*)(h2a_lds + 256 + (4 * kq + vq) * 4 + ((vpp >> 1) ^ (2 * (kq & 1)))) + 8 * (vpp & 1);
    const int vdi = ((vpp >> 1) ^ (2 * (kq & 1))) & 2 ? -32 : 32;      // dims 16 .. 31: chunk + 2 = slot ^ 2 = 32 bytes up or down
    bool bad = false;
    const int64_t cls_row = cls_only ? seq : -1;
    if (w < n_tiles) {                                     // (wave-uniform; every lane of the wave runs the transposed reads)
        if (n_tiles > 16) h2a_tiles<4>(qkv, xs, t0, S, ng, head, w, n_tiles, kimg, vimg, vdi, r, kq, ctx2, os, cls_row, bad);
        else if (n_tiles > 8) h2a_tiles<2>(qkv, xs, t0, S, ng, head, w, n_tiles, kimg, vimg, vdi, r, kq, ctx2, os, cls_row, bad);
        else h2a_tiles<1>(qkv, xs, t0, S, ng, head, w, n_tiles, kimg, vimg, vdi, r, kq, ctx2, os, cls_row, bad);
    }
    if (bad) atomicOr(flag, 1u);
}

// Short sequences (<= 64 tokens: the query encoder's 16-token queries): a whole workgroup per (sequence, head) is mostly
// launch and barrier -- 24 576 workgroups of eight waves with one tile of work among them took 201 us per layer at 2048 x 16
// tokens.  Here a WAVE owns a (sequence, head): it stages the one or two key groups into its own 16 KB of LDS (the same image,
// no barrier: only this wave reads what its own LDS-DMA wrote, after s_waitcnt vmcnt(0)) and runs all of its <= 4 query tiles.
#define H2A_SMALL_GROUPS 2
#define H2A_SMALL_LDS (8 * H2A_SMALL_GROUPS * H2A_GROUP_UNITS * 16)
__global__ __launch_bounds__(512) void ce_attention_h2_small(const h2_u32x4* __restrict__ qkv, int64_t xs, const int32_t* __restrict__ cu,
                                                                int n_pairs, h2_u32x2* __restrict__ ctx2, int64_t os,
                                                                unsigned* __restrict__ flag, int cls_only, int wave_groups) {
    extern __shared__ __attribute__((aligned(16))) h2_u32x4 h2a_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x * 8 + w;
    if (pair >= n_pairs) return;                           // (whole wave; no barrier in this kernel)
    const int seq = pair / 12, head = pair % 12;
    const int t0 = cu[seq], S = cu[seq + 1] - t0;
    const int ng = (S + 31) >> 5;                          // <= H2A_SMALL_GROUPS (the launcher checked max_len)
    h2_u32x4* const mine = h2a_lds + w * (wave_groups * H2A_GROUP_UNITS);      // (one or two groups per wave: 64 or 128 KB per workgroup)
    // staging: the wave's eight pieces per group, piece j = (K | V: j >> 2, plane (j >> 1) & 1, key half j & 1)
    for (int gi = 0; gi < ng; ++gi)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int which = j >> 2, p = (j >> 1) & 1, key = 16 * (j & 1) + (lane >> 2), slot = lane & 3;
            const int chunk = slot ^ (2 * ((key >> 2) & 1));
            int k = 32 * gi + key;
            k = k < S ? k : S - 1;
            const h2_u32x4* src = qkv + (int64_t)(p * H2A_CHUNKS + 48 * (1 + which) + head * 4 + chunk) * xs + t0 + k;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(mine + gi * H2A_GROUP_UNITS + which * 256 + p * 128 + 64 * (j & 1)),
                                             16, 0, 0);
        }
    const int r = lane & 15, kq = lane >> 4;
    const int n_tiles = cls_only ? 1 : (S + 15) >> 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const h2_u32x4* kimg = mine + r * 4 + (kq ^ (2 * ((r >> 2) & 1)));
    const int vq = (lane >> 2) & 3, vpp = lane & 3;
    const char* vimg = (const char*)(mine + 256 + (4 * kq + vq) * 4 + ((vpp >> 1) ^ (2 * (kq & 1)))) + 8 * (vpp & 1);
    const int vdi = ((vpp >> 1) ^ (2 * (kq & 1))) & 2 ? -32 : 32;
    bool bad = false;
    const int64_t cls_row = cls_only ? seq : -1;
    if (n_tiles > 2) h2a_tiles<4, 1>(qkv, xs, t0, S, ng, head, 0, n_tiles, kimg, vimg, vdi, r, kq, ctx2, os, cls_row, bad);
    else if (n_tiles > 1) h2a_tiles<2, 1>(qkv, xs, t0, S, ng, head, 0, n_tiles, kimg, vimg, vdi, r, kq, ctx2, os, cls_row, bad);
    else h2a_tiles<1, 1>(qkv, xs, t0, S, ng, head, 0, n_tiles, kimg, vimg, vdi, r, kq, ctx2, os, cls_row, bad);
    if (bad) atomicOr(flag, 1u);
}

void ce_h2_attention(const void* qkv, int64_t xs, const int32_t* cu, int n_seqs, int max_len, void* ctx2, int64_t os, unsigned* flag,
                     int cls_only, hipStream_t st) {
    const int groups = (max_len + 31) / 32;
    static const bool no_small = getenv("RR_CE_H2_ATT_NO_SMALL") != nullptr;      // (A/B)
    if (groups <= H2A_SMALL_GROUPS && !no_small) {
        const int n_pairs = n_seqs * 12;
        hipLaunchKernelGGL(ce_attention_h2_small, dim3((unsigned)((n_pairs + 7) / 8)), dim3(512), (size_t)8 * groups * H2A_GROUP_UNITS * 16, st,
                           (const h2_u32x4*)qkv, xs, cu, n_pairs, (h2_u32x2*)ctx2, os, flag, cls_only, groups);
        return;
    }
    const size_t lds = (size_t)(groups < H2A_MAX_GROUPS ? groups : H2A_MAX_GROUPS) * H2A_GROUP_UNITS * 16;
    hipLaunchKernelGGL(ce_attention_h2, dim3((unsigned)n_seqs, 12), dim3(512), lds, st, (const h2_u32x4*)qkv, xs, cu, (h2_u32x2*)ctx2, os, flag,
                       cls_only);
}

int ce_h2_set_attributes() {
#define H2_ATTR(E) RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_gemm_h2<E>, hipFuncAttributeMaxDynamicSharedMemorySize, H2G_LDS))
    H2_ATTR(CE_H2_EPI_F32);
    H2_ATTR(CE_H2_EPI_H2);
    H2_ATTR(CE_H2_EPI_GELU_H2);
#undef H2_ATTR
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_attention_h2, hipFuncAttributeMaxDynamicSharedMemorySize, H2A_LDS));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_attention_h2_small, hipFuncAttributeMaxDynamicSharedMemorySize, H2A_SMALL_LDS));
    return RR_OK;
}

#ifdef RR_DEBUG_HARNESS
// h2 image -> fp32 rows (the inverse of ce_h2_pack: hi + lo / 2048), for the harness's GEMM check
__global__ __launch_bounds__(256) void ce_h2_unpack_kernel(const h2_u32x4* __restrict__ src, int R, int K, int64_t rs, float* __restrict__ dst) {
    const int KC = K >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)R * KC) return;
    const int row = (int)(i % R), kc = (int)(i / R);
    const h2_f16x8 hi = __builtin_bit_cast(h2_f16x8, src[(int64_t)kc * rs + row]), lo = __builtin_bit_cast(h2_f16x8, src[((int64_t)KC + kc) * rs + row]);
    for (int e = 0; e < 8; ++e) dst[(int64_t)row * K + 8 * kc + e] = (float)hi[e] + (float)lo[e] * CE_H2_INV_SCALE;
}

// tests/test_gpu_k5.py (child process on librr_hip_dbg.so): out[M][N] = x[M][K] w[N][K]^T + bias through ce_h2_pack +
// ce_gemm_h2, device pointers, fp32 row-major; epi = CE_H2_EPI_F32, or CE_H2_EPI_H2 / CE_H2_EPI_GELU_H2 with the h2 result
// unpacked to fp32 (qscale 0.5 on the first `qcols` features for CE_H2_EPI_H2).  *flag_out = the range flag.
extern "C" int rr_debug_ce_h2_gemm(int32_t epi, int32_t M, int32_t N, int32_t K, const float* d_x, const float* d_w, const float* d_bias,
                                   int32_t qcols, float* d_out, int32_t* flag_out) {
    RR_REQUIRE(M >= 1 && N % 128 == 0 && K % 32 == 0 && d_x && d_w && d_bias && d_out && flag_out, "rr_debug_ce_h2_gemm: bad arguments");
    if (int rc = ce_h2_set_attributes()) return rc;
    const int64_t xs = rr_round_up(M, 256);
    void *x2 = nullptr, *w2 = nullptr, *o2 = nullptr;
    unsigned* flag = nullptr;
    RR_HIP_TRY(hipMalloc(&x2, (size_t)xs * K * 4));
    RR_HIP_TRY(hipMalloc(&w2, (size_t)N * K * 4));
    RR_HIP_TRY(hipMalloc(&o2, (size_t)xs * N * 4));
    RR_HIP_TRY(hipMalloc((void**)&flag, 4));
    RR_HIP_TRY(hipMemset(x2, 0, (size_t)xs * K * 4));
    RR_HIP_TRY(hipMemset(flag, 0, 4));
    ce_h2_pack(d_x, M, K, x2, xs, nullptr);
    ce_h2_pack(d_w, N, K, w2, N, nullptr);
    ce_h2_gemm(epi, w2, N, x2, xs, M, K, d_bias, d_out, o2, xs, flag, nullptr, 0.5f, qcols);
    if (epi != CE_H2_EPI_F32) {
        const int64_t n = (int64_t)M * (N / 8);
        hipLaunchKernelGGL(ce_h2_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const h2_u32x4*)o2, M, N, xs, d_out);
    }
    RR_HIP_TRY(hipDeviceSynchronize());
    unsigned f = 0;
    RR_HIP_TRY(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
    *flag_out = (int32_t)f;
    hipFree(x2); hipFree(w2); hipFree(o2); hipFree(flag);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

extern "C" int rr_debug_ce_h2_stamps(unsigned long long* out16) {
    RR_HIP_TRY(hipDeviceSynchronize());
    RR_HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(h2_dbg), sizeof(unsigned long long) * 16));
    return RR_OK;
}
#endif
