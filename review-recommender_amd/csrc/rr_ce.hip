// rr_ce.hip -- K5: BERT-family encoder forward on the gfx950 matrix cores (cross-encoder rerank + query encoder).
//
// Replaces `CrossEncoder.predict(pairs, batch_size=64, show_progress_bar=False)` as the reference calls it
// (app/app_product_search.py:271-282, app/test.py:217-225): a BertForSequenceClassification forward
// (ms-marco-MiniLM-L-6-v2: 6 layers, hidden 384, 12 heads x 32, FFN 1536, <= 512 tokens, 1 label) over
// (query, text[:2000]) pairs -- and, with mode RR_CE_OUT_CLS, the SentenceTransformer query encoder of
// app/app_product_search.py:250-251 (bge-small: the same block shape, 12 layers, CLS pooling).
//
// Layout: sequences are PACKED (no padding tokens are ever computed): T = sum of lengths, cu_seqlens[P+1].
//   residual stream  h32 [T][384] fp32  +  hb [T][384] bf16 (the next GEMM's A operand)
//   qkv [T][1152] bf16, ctx [T][384] bf16, inter [T][1536] bf16
// Arithmetic: bf16 MFMA operands (weights rounded once at load, activations at each producer's epilogue),
// fp32 accumulation, fp32 residual / LayerNorm / softmax / GELU(erf) / pooler / classifier.
//
// Kernels (per layer: 4 GEMMs + 1 attention; MFMA-bound, ~4.3 MFLOP per token and layer at 512 tokens):
//   ce_embed_ln      word + position + type embedding, LayerNorm               (HBM-bound, one wave per token)
//   ce_gemm<128x384> eight waves (2 x 4, 64 x 96 per wave), one workgroup per CU at two waves per SIMD; epilogues:
//                    out = A W^T + b [, GELU] -> bf16                          (QKV 384->1152, FFN1 384->1536)
//                    out = LayerNorm(A W^T + b + residual) -> fp32 + bf16      (attention output, FFN2 1536->384):
//                    a workgroup owns whole rows (BN = hidden), so the normalisation is fused into the epilogue
//   ce_attention     one workgroup (eight waves) per (sequence, head): K and V^T of the head in LDS, S^T = K Q^T on
//                    v_mfma_f32_16x16x32_bf16 (K = head dim = 32: one MFMA per 16x16 score tile), base-2 online
//                    softmax over 128-key chunks in registers, and the probability tile is fed straight back as
//                    the A operand of P V (the accumulator-as-operand idiom: no lane movement, no LDS round trip)
//   ce_head          pooler (tanh) + classifier on the [CLS] rows, fp32
// GEMM tiles: 32x32x16 bf16 MFMA, K-step 64 through LDS rows padded to 144 B (conflict-free ds_read_b128),
// next K tile prefetched global -> registers under the MFMAs of the current one.
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "rr_common.h"
#include "rr_ce_h2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#define CE_H 384
#define CE_HEADS 12
#define CE_HD 32
#define CE_FFN 1536
#define CE_LDK 72          // LDS row of a K tile: 64 bf16 + 8 pad = 144 B

__device__ __forceinline__ unsigned short ce_bf16_bits(float x) {
    const __bf16 b = (__bf16)x;                      // round to nearest even
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float ce_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------ embeddings + LayerNorm
__global__ __launch_bounds__(256) void ce_embed_ln(const int32_t* __restrict__ tok, const int32_t* __restrict__ typ,
                                                   const int32_t* __restrict__ pos, int T, int vocab, int n_pos, int n_typ,
                                                   const float* __restrict__ we, const float* __restrict__ pe,
                                                   const float* __restrict__ te, const float* __restrict__ g,
                                                   const float* __restrict__ b, float eps, float* __restrict__ h32,
                                                   unsigned short* __restrict__ hb) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    int id = tok[t], ty = typ[t], po = pos[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // ids are validated on the host; stay in bounds anyway
    ty = ty < 0 ? 0 : (ty >= n_typ ? n_typ - 1 : ty);
    po = po < 0 ? 0 : (po >= n_pos ? n_pos - 1 : po);
    float x[6];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        x[i] = (we[(int64_t)id * CE_H + c] + te[(int64_t)ty * CE_H + c]) + pe[(int64_t)po * CE_H + c];
        s += x[i];
    }
    const float mean = ce_wave_sum(s) * (1.f / CE_H);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) { const float d = x[i] - mean; v += d * d; }
    const float rstd = rsqrtf(ce_wave_sum(v) * (1.f / CE_H) + eps);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        const float y = (x[i] - mean) * rstd * g[c] + b[c];
        h32[(int64_t)t * CE_H + c] = y;
        hb[(int64_t)t * CE_H + c] = ce_bf16_bits(y);
    }
}

// ------------------------------------------------------------------ GEMM  out[M][N] = A[M][K] W[N][K]^T (+ epilogue)
#define CE_EPI_BIAS 0        // + bias                      -> bf16
#define CE_EPI_GELU 1        // gelu_erf(+ bias)            -> bf16
#define CE_EPI_RES_LN 2      // LayerNorm(+ bias + res32)   -> fp32 (in place over res32) and bf16; needs BN == N

// gelu(x) = x/2 (1 + erf(x / sqrt 2)), erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16
// rounding of the result): 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2; one v_rcp, one v_exp.
// Two values at a time on the packed fp32 pipe (v_pk_mul / v_pk_fma / v_pk_add: two lanes' worth per instruction):
// the same formula; gelu(x) = x/2 * (x >= 0 ? 2 - q : q) with q = poly(t) t exp(-z^2) = 1 - erf(|x| / sqrt 2).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CE_GELU_X 4.5f
__device__ constexpr float CE_GELU_P[11] = {4.877777100e-01f, 7.139661163e-02f, -1.809151024e-01f, 2.453869879e-01f, -1.558733135e-01f,
                                            -3.181198984e-02f, 1.249853820e-01f, -5.963810533e-02f, -2.130715176e-02f, 2.488414198e-02f,
                                            -4.885168280e-03f};
// bf16 path: no transcendental (a v_rcp + a v_exp per value are 32 of the ~54 vector-pipe cycles the formula above costs per
// value, and in ce_ffn_fused that stretch was longer than the matrix work beside it).  Phi(x) - 1/2 is odd and flat beyond
// |x| = 4.5 (1/2 - 3.4e-6): a degree-10 polynomial in t = 2 min(|x|, 4.5) / 4.5 - 1 (weighted least-squares Chebyshev fit,
// tools/fit_gelu.py), sign restored; |gelu_poly - gelu| <= 1.6e-5 for every x (the output is rounded to bf16 next: 2^-9
// relative).  The fp32 mode keeps the formula above (ce_gelu).  In three pieces so that ce_ffn_fused can spread a value's
// work over several MFMA slots: t, the Horner steps, the finish.
__device__ __forceinline__ f32x2_t ce_gelu_t(f32x2_t x) {
    const f32x2_t a = {__builtin_fminf(__builtin_fabsf(x[0]), CE_GELU_X), __builtin_fminf(__builtin_fabsf(x[1]), CE_GELU_X)};
    return __builtin_elementwise_fma(a, (f32x2_t){2.f / CE_GELU_X, 2.f / CE_GELU_X}, (f32x2_t){-1.f, -1.f});
}
template <int K_HI, int K_LO>      // Horner steps with coefficients K_HI ... K_LO (h = P[10] before step 9)
__device__ __forceinline__ f32x2_t ce_gelu_horner(f32x2_t h, f32x2_t t) {
#pragma unroll
    for (int k = K_HI; k >= K_LO; --k) h = __builtin_elementwise_fma(h, t, (f32x2_t){CE_GELU_P[k], CE_GELU_P[k]});
    return h;
}
__device__ __forceinline__ f32x2_t ce_gelu_finish(f32x2_t h, f32x2_t x) {
    const f32x2_t hs = {__builtin_copysignf(h[0], x[0]), __builtin_copysignf(h[1], x[1])};
    return x * (hs + (f32x2_t){0.5f, 0.5f});
}
__device__ __forceinline__ f32x2_t ce_gelu2(f32x2_t x) {
    const f32x2_t t = ce_gelu_t(x);
    return ce_gelu_finish(ce_gelu_horner<9, 0>((f32x2_t){CE_GELU_P[10], CE_GELU_P[10]}, t), x);
}
__device__ __forceinline__ float ce_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erf_abs = 1.f - poly * t * e;
    const float erf_x = x < 0.f ? -erf_abs : erf_abs;
    return 0.5f * x * (1.f + erf_x);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N == 4 ? 2 : 1)) void ce_gemm(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W,
                                               const float* __restrict__ bias, int M, int N, int K,
                                               unsigned short* __restrict__ outb, float* __restrict__ res32,
                                               const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps) {
    constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N, MB = TM / 32, NB = TN / 32;
    constexpr int A_CHUNKS = BM * 8 / THREADS, W_CHUNKS = BN * 8 / THREADS;      // 16-B pieces per thread and K tile
    static_assert(BM * 8 % THREADS == 0 && BN * 8 % THREADS == 0, "tile rows must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char ce_smem[];
    unsigned short* As = reinterpret_cast<unsigned short*>(ce_smem);           // [BM][CE_LDK]
    unsigned short* Ws = As + BM * CE_LDK;                                      // [BN][CE_LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int r32 = lane & 31, hh = lane >> 5;
    // 1-D grid, XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so the
    // column tiles of one row tile are given consecutive ids ON THE SAME XCD: they run together and share the A
    // tile in that L2 (A is then read from HBM once instead of once per column tile; W is L2-resident everywhere).
    const int NT = N / BN;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int64_t mtile = (int64_t)(loc / NT) * 8 + xcd;
    const int64_t row0 = mtile * BM;
    const int col0 = (loc % NT) * BN;
    if (row0 >= M) return;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // K tiles travel global -> registers -> LDS with a prefetch distance of TWO tiles (two register sets): a
    // workgroup is alone on its CU (two waves per SIMD, all in step behind the barriers), so a tile's loads need
    // two compute phases, not one, to cover the L2 / HBM latency under load.
    u32x4 pa0[A_CHUNKS], pw0[W_CHUNKS], pa1[A_CHUNKS], pw1[W_CHUNKS];
    auto load_tile = [&](int kt, u32x4 (&pa)[A_CHUNKS], u32x4 (&pw)[W_CHUNKS]) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int c = tid + THREADS * i, r = c >> 3, p = c & 7;
            int64_t row = row0 + r;
            row = row < M ? row : M - 1;                                   // rows past M are never stored
            pa[i] = *reinterpret_cast<const u32x4*>(A + row * K + kt * 64 + p * 8);
        }
#pragma unroll
        for (int i = 0; i < W_CHUNKS; ++i) {
            const int c = tid + THREADS * i, r = c >> 3, p = c & 7;
            pw[i] = *reinterpret_cast<const u32x4*>(W + (int64_t)(col0 + r) * K + kt * 64 + p * 8);
        }
    };
    auto store_tile = [&](const u32x4 (&pa)[A_CHUNKS], const u32x4 (&pw)[W_CHUNKS]) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int c = tid + THREADS * i, r = c >> 3, p = c & 7;
            *reinterpret_cast<u32x4*>(As + r * CE_LDK + p * 8) = pa[i];
        }
#pragma unroll
        for (int i = 0; i < W_CHUNKS; ++i) {
            const int c = tid + THREADS * i, r = c >> 3, p = c & 7;
            *reinterpret_cast<u32x4*>(Ws + r * CE_LDK + p * 8) = pw[i];
        }
    };
    auto compute_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[MB], wf[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(As + (wm * TM + i * 32 + r32) * CE_LDK + ks * 16 + hh * 8);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                wf[j] = *reinterpret_cast<const bf16x8*>(Ws + (wn * TN + j * 32 + r32) * CE_LDK + ks * 16 + hh * 8);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
    };

    // Barriers of the K loop: raw s_barrier behind an explicit LDS wait.  __syncthreads() would also wait vmcnt(0)
    // (hipcc fences outstanding global loads at it), i.e. drain the prefetched tiles at every barrier.
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int KT = K / 64;                           // even (K is 384 or 1536)
    load_tile(0, pa0, pw0);
    load_tile(1, pa1, pw1);
    for (int kt = 0; kt < KT; kt += 2) {
        store_tile(pa0, pw0);
        lds_barrier();
        if (kt + 2 < KT) load_tile(kt + 2, pa0, pw0);
        compute_tile();
        lds_barrier();
        store_tile(pa1, pw1);
        lds_barrier();
        if (kt + 3 < KT) load_tile(kt + 3, pa1, pw1);
        compute_tile();
        lds_barrier();
    }
    __syncthreads();

    // C layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    if constexpr (EPI != CE_EPI_RES_LN) {
        // bias (+ GELU), then the wave's TM x TN sub-tile goes through its own LDS region so that it leaves as
        // 16-byte row pieces (TN * 2 / 16 per row) instead of 2-byte stores: neighbouring columns sit in
        // neighbouring lanes, so lane pairs first exchange one value and write packed bf16 pairs.
        constexpr int SLD = TN + 8;                                        // staging row: TN bf16 + 16 B pad
        unsigned short* stage = reinterpret_cast<unsigned short*>(ce_smem) + wave * (TM * SLD);
        const int odd = lane & 1;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float bv = bias[col0 + wn * TN + j * 32 + r32];
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    float v0 = acc[i][j][e] + bv, v1 = acc[i][j][e + 1] + bv;      // rows R(e), R(e) + 1, column r32
                    if (EPI == CE_EPI_GELU) { const f32x2_t gg = ce_gelu2((f32x2_t){v0, v1}); v0 = gg[0]; v1 = gg[1]; }
                    const float got = __shfl_xor(odd ? v0 : v1, 1, 64);
                    // even lane: row R(e), columns (c, c + 1);  odd lane: row R(e) + 1, columns (c - 1, c)
                    const bf16x2_t pk = odd ? bf16x2_t{(__bf16)got, (__bf16)v1} : bf16x2_t{(__bf16)v0, (__bf16)got};
                    const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh + odd;
                    *reinterpret_cast<unsigned int*>(stage + row * SLD + j * 32 + (r32 & ~1)) = __builtin_bit_cast(unsigned int, pk);
                }
            }
        __syncthreads();
        constexpr int PIECES = TN / 8;                                     // 16-byte pieces per row of the sub-tile
#pragma unroll
        for (int u = 0; u < TM * PIECES / 64; ++u) {
            const int idx = lane + 64 * u, row = idx / PIECES, pc = idx % PIECES;
            const int64_t grow = row0 + wm * TM + row;
            const u32x4 val = *reinterpret_cast<const u32x4*>(stage + row * SLD + pc * 8);
            if (grow < M) *reinterpret_cast<u32x4*>(outb + grow * N + col0 + wn * TN + pc * 8) = val;
        }
    } else {
        // whole rows live in this workgroup (BN == N, WAVES_M == 1): x = acc + bias + residual, two-pass LayerNorm.
        // Row sums: in-lane over the wave's NB column blocks, a reduce-scatter butterfly over the 32 lanes of a
        // half (31 shuffles for the lane's MB*16 rows), the four waves' partials through LDS.
        static_assert(MB * 16 == 32, "LayerNorm epilogue: 64 rows per wave");
        float* red = reinterpret_cast<float*>(ce_smem);                     // [WAVES_N][BM] partial row sums
        float* stat = red + WAVES_N * BM;                                    // [2][BM] mean, rstd
        float bv[NB], gv[NB], be[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = wn * TN + j * 32 + r32;
            bv[j] = bias[col]; gv[j] = ln_g[col]; be[j] = ln_b[col];
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = row0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int64_t rr = row < M ? row : M - 1;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j][e] = (acc[i][j][e] + bv[j]) + res32[rr * N + wn * TN + j * 32 + r32];
            }
        // pass 0: row means; pass 1: centred sums of squares.  The statistics stay in LDS (stat: mean, then rstd)
        // and are re-read where needed, so the epilogue holds no per-row register arrays beside the accumulators.
        float* mean_s = stat;                                                // [BM]
        float* rstd_s = stat + BM;                                           // [BM]
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float v[32];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float mu = pass == 0 ? 0.f : mean_s[wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh];
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const float d = acc[i][j][e] - mu;
                        s += pass == 0 ? d : d * d;
                    }
                    v[i * 16 + e] = s;
                }
#pragma unroll
            for (int si = 0; si < 5; ++si) {
                const int st = 16 >> si;
                const bool up = (lane & st) != 0;
#pragma unroll
                for (int q = 0; q < st; ++q) {
                    const float keep = up ? v[q + st] : v[q];
                    const float send = up ? v[q] : v[q + st];
                    v[q] = keep + __shfl_xor(send, st, 64);
                }
            }
            // lane (r32, hh) now holds the wave's sum of slot s = r32: block i = s >> 4, e = s & 15
            {
                const int s_ = r32, i = s_ >> 4, e = s_ & 15;
                red[wn * BM + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh] = v[0];
            }
            __syncthreads();
            if (tid < BM) {
                float tot = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < WAVES_N; ++w_) tot += red[w_ * BM + tid];
                if (pass == 0) mean_s[tid] = tot * (1.f / BN);
                else rstd_s[tid] = rsqrtf(tot * (1.f / BN) + ln_eps);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rt = wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int64_t row = row0 + rt;
                if (row < M) {
                    const float mu = mean_s[rt], rs = rstd_s[rt];
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int col = wn * TN + j * 32 + r32;
                        const float y = (acc[i][j][e] - mu) * rs * gv[j] + be[j];
                        res32[row * N + col] = y;
                        outb[row * N + col] = ce_bf16_bits(y);
                    }
                }
            }
    }
}

// ------------------------------------------------------------------ fused FFN: LayerNorm(h + W2 gelu(W1 h + b1) + b2)
// One kernel for intermediate.dense + GELU + output.dense + residual + LayerNorm, in TRANSPOSED form: the wave's 32
// tokens sit on the MFMA columns (lanes) for the whole kernel.
//   X^T[j][tok] = W1[j] . h[tok]      A operand = W1 rows of a 32-feature chunk (LDS), B operand = the tokens' bf16 rows,
//                                     held in registers for the whole kernel (24 K-steps x 4 VGPRs, loaded once)
//   out^T[n][tok] += W2[n][j] X^T[j][tok]   the 32 x 32 accumulator tile X^T (+ b1, GELU, -> bf16) IS the B operand of the
//                                     second product as it sits (token on the lane, features in registers): no LDS round
//                                     trip, no lane movement.  Its k order inside a 16-step is permuted (element jj of
//                                     lane half h = feature 16 s + 8 (jj >> 2) + 4 h + (jj & 3)); W2 is stored with its
//                                     columns permuted the same way once at load (rr_ce_create: w2p).
// The 6 KB-per-token intermediate never leaves the CU, the whole FFN is one launch, and the LayerNorm statistics of a
// token are sums over the registers of one lane pair (lane, lane ^ 32): no cross-wave reduction.
// Workgroup = 4 waves x 32 tokens, one wave per SIMD (hb fragments 96 + out^T accumulators 192 registers per lane); W1 / W2
// chunks (24 KB each) stream L2 -> registers -> LDS, double-buffered, one barrier per iteration; per iteration and wave 48
// MFMAs against 48 ds_read_b128.  Round 3: the iteration is one sequence of 48 MFMA slots (the second product of the
// previous chunk, then the first product of the next) with the GELU -- Phi by table interpolation --, the staging stores
// and loads dealt out over the slots; the residual row is the accumulators' start; the rows leave through LDS in whole
// lines; OPROJ puts the attention output projection + first LayerNorm in front.  
#define CE_FFN_TOK 128          // tokens per workgroup
#define CE_FFN_CH 32            // intermediate features per chunk
#define CE_W1_LD (CE_H + 8)     // LDS row of a W1 chunk: 384 bf16 + 16 B pad
#define CE_W2_LD (CE_FFN_CH + 8)    // LDS row of a W2 chunk: 32 bf16 + 16 B pad
#define CE_FFN_BUF (CE_FFN_CH * CE_W1_LD + CE_H * CE_W2_LD)     // bf16 elements per chunk buffer
#define CE_GELU_N 512           // intervals of the fused FFN's Phi table over [-CE_GELU_R, CE_GELU_R]
#define CE_GELU_R 4.5f
#define CE_FFN_LDS (2 * CE_FFN_BUF * 2 + CE_FFN * 4 + 6 * CE_H * 4 + CE_GELU_N * 8)   // two chunk buffers + b1 + (b2, ln_g, ln_b) + the Phi table + (bo, ln1_g, ln1_b)

// DEPTH: A fragments in flight.  STAGGER rotates the chunk order per workgroup (it spreads the L2 lines the CUs ask for at
// one time, but makes a token's rounding depend on where it sits in the batch): measured r02 at 256 x 512 tokens, whole
// forward: <4, false> 6.26 ms, <8, false> 6.38 ms, <4, true> 6.38 ms -- neither LDS depth nor L2 hot-spotting is what
// bounds the kernel (a single wave per SIMD issues its VALU, LDS and MFMA work in order); only <4, false> is built.
#ifdef RR_DEBUG_HARNESS
// in-kernel phase clocks of ce_ffn_fused (tools/k5_stamps.py): [wave 0 of workgroups 0 and 600][10] cycle sums
__device__ unsigned long long ce_dbg_ffn[2][10];
#define CE_STAMP(slot) do { const unsigned long long t_ = clock64(); dbg_t[slot] += t_ - dbg_last; dbg_last = t_; } while (0)
#else
#define CE_STAMP(slot) do { } while (0)
#endif

// OPROJ: the attention output projection + residual + first LayerNorm run in front, in the same transposed form (the rows of
// `ctx` are the B fragments, Wo streams through the W1 chunk buffers, out^T accumulates on top of the residual): the
// normalised rows stay in the accumulators as the second LayerNorm's residual and become the FFN's B fragments by one
// v_permlane32_swap per two dwords -- the [T][384] fp32 + bf16 round trip through HBM between the two kernels, the second
// kernel's row loads and the first one's row stores are gone.
// (NOSTORE: debug library only, RR_CE_FFN_NOSTORE=1 -- the staging loads without their LDS stores in the main loop; garbage results)
template <int DEPTH, bool STAGGER, bool OPROJ, bool NOSTORE = false>
__global__ __launch_bounds__(256, 1) void ce_ffn_fused(
    unsigned short* __restrict__ hb, float* __restrict__ h32, int M,
    const unsigned short* __restrict__ W1, const float* __restrict__ b1,       // [1536][384], [1536]
    const unsigned short* __restrict__ W2p, const float* __restrict__ b2,      // [384][1536] columns permuted, [384]
    const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    const float* __restrict__ gelu_tab,                                         // [CE_GELU_N][2]: ce_gelu_table
    const unsigned short* __restrict__ ctx, const unsigned short* __restrict__ Wo, const float* __restrict__ bo,
    const float* __restrict__ ln1_g, const float* __restrict__ ln1_b) {         // OPROJ only: [M][384], [384][384], 3 x [384]
    extern __shared__ __attribute__((aligned(16))) unsigned char ce_smem[];
    unsigned short* wbuf = reinterpret_cast<unsigned short*>(ce_smem);          // [2][CE_FFN_BUF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    const int64_t tok0 = (int64_t)blockIdx.x * CE_FFN_TOK + wave * 32;
    int64_t tok = tok0 + c;
    tok = tok < M ? tok : M - 1;                 // (lanes past M work on row M - 1 again and write the same bytes to it)

    // the tokens' rows as B fragments: lane (c, hh), K-step s holds hb[tok][16 s + 8 hh .. + 7]
    bf16x8 hf[24];
#pragma unroll
    for (int s = 0; s < 24; ++s) hf[s] = *reinterpret_cast<const bf16x8*>((OPROJ ? ctx : hb) + tok * CE_H + 16 * s + 8 * hh);

    // out^T accumulates ON TOP of the token's residual row: its 48 loads are in flight under the staging of the first chunks
    // instead of standing, all workgroups at once, between the last MFMA and the LayerNorm (r03 in-kernel clocks: the
    // epilogue was 42 .. 66 thousand cycles of a workgroup's 250 thousand)
    f32x16 acc[12];
#pragma unroll
    for (int nb = 0; nb < 12; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(h32 + tok * CE_H + nb * 32 + 8 * g + 4 * hh);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nb][4 * g + r] = rv[r];
        }

    // chunk staging: a W1 chunk (rows j0 .. j0+31: 32 x 768 B) and a W2p chunk (columns j0 .. j0+31 of all 384 rows:
    // 384 x 64 B) are 1536 pieces of 16 B each = 6 per thread, each kind with its own six staging registers: both loads
    // are issued a whole chunk (>= 1536 matrix-pipe cycles) before their LDS stores, and NO other vector-memory load sits
    // between a load and its store's wait (vmcnt retires in order: the b1 loads this loop used to issue per chunk made
    // their first use wait for the weight prefetch issued just before them -- b1 now comes from LDS)
    u32x4 pw[6], pw2[6];
    auto load_w1 = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 256 * i;
            pw[i] = *reinterpret_cast<const u32x4*>(W1 + (int64_t)(ch * CE_FFN_CH + id / 48) * CE_H + (id % 48) * 8);
        }
    };
    auto load_w2 = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 256 * i;
            pw2[i] = *reinterpret_cast<const u32x4*>(W2p + (int64_t)(id >> 2) * CE_FFN + ch * CE_FFN_CH + (id & 3) * 8);
        }
    };
    auto w1_of = [&](int k) { return wbuf + (k & 1) * CE_FFN_BUF; };                          // LDS home of W1 chunk k
    auto w2_of = [&](int k) { return wbuf + (k & 1) * CE_FFN_BUF + CE_FFN_CH * CE_W1_LD; };    // ... of W2 chunk k
    auto store_w1 = [&](unsigned short* buf) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 256 * i;
            *reinterpret_cast<u32x4*>(buf + (id / 48) * CE_W1_LD + (id % 48) * 8) = pw[i];
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

#ifdef RR_DEBUG_HARNESS
    unsigned long long dbg_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dbg_last = clock64();
    const unsigned long long dbg_w0 = wall_clock64();
#endif
    constexpr int NCH = CE_FFN / CE_FFN_CH;          // 48 chunks
    // DEPTH A fragments are in flight ahead of their MFMA: one wave per SIMD, so the LDS latency (~100+ cycles against
    // 32 per MFMA) is hidden by depth
    const int rot = STAGGER ? (int)((blockIdx.x * 7u) % NCH) : 0;          // (experiment: spreads the L2 lines the CUs ask for)
    auto chunk_of = [&](int k) { return STAGGER ? (k + rot) % NCH : k; };
    float* b1s = reinterpret_cast<float*>(wbuf + 2 * CE_FFN_BUF);               // [1536] the first bias, read per chunk
    float* eps_ = b1s + CE_FFN;                                                 // [3][384] b2, ln_g, ln_b for the epilogue
    for (int i = tid; i < CE_FFN / 4; i += 256) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(b1)[i];
    float* gtab = eps_ + 3 * CE_H;                                              // [CE_GELU_N][2] Phi(x_i), Phi(x_i+1) - Phi(x_i)
    for (int i = tid; i < CE_GELU_N / 2; i += 256) reinterpret_cast<f32x4*>(gtab)[i] = reinterpret_cast<const f32x4*>(gelu_tab)[i];
    for (int i = tid; i < 3 * CE_H / 4; i += 256) {
        const float* src = i < CE_H / 4 ? b2 : i < 2 * CE_H / 4 ? ln_g : ln_b;
        reinterpret_cast<f32x4*>(eps_)[i] = reinterpret_cast<const f32x4*>(src)[i % (CE_H / 4)];
    }
    float* eps1_ = gtab + 2 * CE_GELU_N;                                        // [3][384] bo, ln1_g, ln1_b (OPROJ)
    if (OPROJ) {
        for (int i = tid; i < 3 * CE_H / 4; i += 256) {
            const float* src = i < CE_H / 4 ? bo : i < 2 * CE_H / 4 ? ln1_g : ln1_b;
            reinterpret_cast<f32x4*>(eps1_)[i] = reinterpret_cast<const f32x4*>(src)[i % (CE_H / 4)];
        }
        // ---- out^T = residual + Wo . ctx^T: twelve chunks of 32 output features, 24 MFMAs each into the chunk's accumulator
        auto load_wo = [&](int k, int i) {
            const int id = tid + 256 * i;
            pw[i] = *reinterpret_cast<const u32x4*>(Wo + (int64_t)(k * CE_FFN_CH + id / 48) * CE_H + (id % 48) * 8);
        };
        auto store_wo = [&](int k, int i) {
            const int id = tid + 256 * i;
            *reinterpret_cast<u32x4*>(w1_of(k) + (id / 48) * CE_W1_LD + (id % 48) * 8) = pw[i];
        };
#pragma unroll
        for (int i = 0; i < 6; ++i) load_wo(0, i);
#pragma unroll
        for (int i = 0; i < 6; ++i) store_wo(0, i);
#pragma unroll
        for (int i = 0; i < 6; ++i) load_wo(1, i);
        lds_barrier();
#pragma unroll
        for (int nb = 0; nb < 12; ++nb) {
            const unsigned short* ap = w1_of(nb) + c * CE_W1_LD + 8 * hh;
            bf16x8 af[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ap + 16 * i);
#pragma unroll
            for (int s = 0; s < 24; ++s) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % DEPTH], hf[s], acc[nb], 0, 0, 0);
                if (s + DEPTH < 24) af[s % DEPTH] = *reinterpret_cast<const bf16x8*>(ap + 16 * (s + DEPTH));
                if (s % 4 == 1 && nb + 1 < 12) store_wo(nb + 1, s >> 2);       // (its home held chunk nb - 1, read a barrier ago)
                if (s % 4 == 3 && nb + 2 < 12) load_wo(nb + 2, s >> 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier();
        }
        // ---- + bo, first LayerNorm (statistics over the lane pair, as at the end of the kernel)
        int h4 = 4 * hh;
        asm volatile("" : "+v"(h4));          // (its own copy: addresses shared with the kernel's last phase would live -- spilled -- across the whole FFN)
        float sum1 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(eps1_ + nb * 32 + 8 * g + h4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[nb][4 * g + r] += bv[r];
                    sum1 += acc[nb][4 * g + r];
                }
            }
        sum1 += __shfl_xor(sum1, 32, 64);
        const float mean1 = sum1 * (1.f / CE_H);
        float var1 = 0.f;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = acc[nb][e] - mean1; var1 += d * d; }
        var1 += __shfl_xor(var1, 32, 64);
        const float rstd1 = rsqrtf(var1 * (1.f / CE_H) + ln_eps);
        // the normalised row: fp32 in the accumulators (the residual the second product adds to) and, rounded to bf16, the
        // FFN's B fragments.  Lane half hh holds features 8 g + 4 hh .. + 3 of each 32; fragment s wants 16 s + 8 hh .. + 7:
        // half 0 keeps its group 2 (s & 1) and takes half 1's, half 1 keeps its group 2 (s & 1) + 1 and takes half 0's --
        // v_permlane32_swap exchanges exactly those (first operand's upper 32 lanes <-> second operand's lower 32)
#pragma unroll
        for (int nb = 0; nb < 12; ++nb) {
            unsigned int pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb * 32 + 8 * g + h4;
                const f32x4 gv = *reinterpret_cast<const f32x4*>(eps1_ + CE_H + n);
                const f32x4 be = *reinterpret_cast<const f32x4*>(eps1_ + 2 * CE_H + n);
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[r] = (acc[nb][4 * g + r] - mean1) * rstd1 * gv[r] + be[r];
                    acc[nb][4 * g + r] = y[r];
                }
                pk[g][0] = __builtin_bit_cast(unsigned int, (bf16x2_t){(__bf16)y[0], (__bf16)y[1]});
                pk[g][1] = __builtin_bit_cast(unsigned int, (bf16x2_t){(__bf16)y[2], (__bf16)y[3]});
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                u32x4 f;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto r = __builtin_amdgcn_permlane32_swap(pk[2 * half][d], pk[2 * half + 1][d], false, false);
                    f[d] = r[0];
                    f[2 + d] = r[1];
                }
                hf[2 * nb + half] = __builtin_bit_cast(bf16x8, f);
            }
        }
    }
    load_w1(chunk_of(0)); store_w1(w1_of(0));
    load_w1(chunk_of(1)); store_w1(w1_of(1));
    lds_barrier();
    // register e of lane half hh is feature (e & 3) + 8 (e >> 2) + 4 hh of the chunk: the accumulator of a first product
    // STARTS from the chunk's bias (four 16-byte LDS reads straight into the C operand: no vector add per value)
    auto bias_of = [&](int k) {
        f32x16 v;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(b1s + chunk_of(k) * CE_FFN_CH + 8 * g + 4 * hh);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * g + r] = q[r];
        }
        return v;
    };
    f32x16 xc = bias_of(0);                          // X^T + b1 of the current chunk (before the GELU)
    {
        const unsigned short* ap = w1_of(0) + c * CE_W1_LD + 8 * hh;
        bf16x8 af[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ap + 16 * i);
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            xc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % DEPTH], hf[s], xc, 0, 0, 0);
            if (s + DEPTH < 24) af[s % DEPTH] = *reinterpret_cast<const bf16x8*>(ap + 16 * (s + DEPTH));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();                                                         // every wave is done with W1 chunk 0: its home is reused below
    load_w1(chunk_of(2));                                                  // stored by iteration 0
    load_w2(chunk_of(0));                                                  // stored by iteration 0
    CE_STAMP(0);                                                           // prologue

    // ---- One iteration = 48 MFMA slots: out^T += W2 chunk (ch - 1) . GELU(X^T(ch - 1)) (24), then X^T(ch + 1) = W1 chunk
    // (ch + 1) . h^T (24), with EVERYTHING else of the iteration dealt out over those slots, a few instructions each: the
    // GELU of chunk ch (its output feeds the next iteration's second product), the LDS stores of the staged weights, the
    // global loads of the next ones.  Why: a wave issues in order and one wave per SIMD has nobody to overlap with -- the
    // probe (tools/probes/mfma_valu_probe.hip) and the in-kernel clocks say a 32x32x16 MFMA hides about five other
    // instructions and every further one adds its own issue cycles, and anything bunched between two products (r03
    // before this: iteration top 550 cycles, staging 2 x 450, GELU beside ONE product 24 x 62) runs with the matrix pipe
    // idle.  The empty asm statements pin each piece to its slot (the compiler otherwise sinks it next to its use).
    u32x4 xbw[2], xbp[2];                            // GELU(X^T) of chunk ch (being made) and of chunk ch - 1 (second product's B)
    float gx[2][2], gf[2][2];
    f32x2_t gt[2][2];
    auto gelu_lookup = [&](int e) {                  // value e of xc: interval of the Phi table, fraction, gather
        const int k = e >> 1, i = e & 1;
        float x = xc[e];
        float t = __builtin_amdgcn_fmed3f(__builtin_fmaf(x, CE_GELU_N / (2.f * CE_GELU_R), CE_GELU_N / 2.f), 0.f, 511.99997f);
        const unsigned int idx = (unsigned int)t;
        float f = __builtin_amdgcn_fractf(t);
        asm volatile("" : "+v"(x), "+v"(f));
        gx[k & 1][i] = x;
        gf[k & 1][i] = f;
        gt[k & 1][i] = reinterpret_cast<const f32x2_t*>(gtab)[idx];
    };
    auto gelu_finish = [&](int k) {                  // pair k -> xbw
        // (one value at a time, each pinned: paired up by the vectoriser they become v_pk_fma / v_pk_mul -- two plain
        //  instructions' worth each on this chip -- plus the moves that line their operands up)
        float p0 = __builtin_fmaf(gf[k & 1][0], gt[k & 1][0][1], gt[k & 1][0][0]);
        asm volatile("" : "+v"(p0));
        float p1 = __builtin_fmaf(gf[k & 1][1], gt[k & 1][1][1], gt[k & 1][1][0]);
        asm volatile("" : "+v"(p1));
        float y0 = gx[k & 1][0] * p0;
        asm volatile("" : "+v"(y0));
        float y1 = gx[k & 1][1] * p1;
        asm volatile("" : "+v"(y1));
        unsigned int pk = __builtin_bit_cast(unsigned int, (bf16x2_t){(__bf16)y0, (__bf16)y1});
        asm volatile("" : "+v"(pk));
        xbw[k >> 2][k & 3] = pk;
    };
    // HAS_A: second product of chunk ch - 1; HAS_G: GELU of chunk ch; HAS_B: first product of chunk ch + 1 (+ the staging)
    auto iteration = [&](int ch, auto has_a, auto has_g, auto has_b) {
        constexpr bool HAS_A = decltype(has_a)::value, HAS_G = decltype(has_g)::value, HAS_B = decltype(has_b)::value;
        const unsigned short* ap2 = w2_of(ch - 1) + c * CE_W2_LD + 8 * hh;
        const unsigned short* ap1 = w1_of(ch + 1) + c * CE_W1_LD + 8 * hh;
        auto frag = [&](int j) {                     // A fragment of slot j
            return j < 24 ? *reinterpret_cast<const bf16x8*>(ap2 + (j >> 1) * 32 * CE_W2_LD + 16 * (j & 1))
                          : *reinterpret_cast<const bf16x8*>(ap1 + 16 * (j - 24));
        };
        constexpr int J0 = HAS_A ? 0 : 24, J1 = HAS_B ? 48 : 24;
        f32x16 xn;
        if (HAS_B) xn = bias_of(ch + 1);             // X^T + b1 of chunk ch + 1 starts from the bias
        bf16x8 af[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            if (J0 + i < J1) af[i] = frag(J0 + i);
        // the staging work of an iteration, one piece per free slot: W1 chunk ch + 2 registers -> LDS and chunk ch + 3 global ->
        // registers, piece by piece; then W2 chunk ch and ch + 1 the same way.  (Chunk numbers past the end are clamped: a store
        // nobody reads.)  Slot j carries a GELU piece when j % 3 == 0 (look-up of value j / 3), j == 6 k + 7 (finish of pair
        // k, its gathers four slots old) and j == 47; the other 24 slots carry staging piece 0 .. 23.
        const int cw1 = chunk_of(ch + 3 < NCH ? ch + 3 : NCH - 1), cw2 = chunk_of(ch + 1 < NCH ? ch + 1 : NCH - 1);
        // (global addresses = a wave-uniform base per piece + ONE per-lane 32-bit offset: a W1 chunk is 24 KB of contiguous
        //  rows, piece id at byte 16 id; W2p piece id is row (id >> 2) = (tid >> 2) + 64 i, 16-byte column id & 3)
        const unsigned char* g1 = reinterpret_cast<const unsigned char*>(W1) + (size_t)cw1 * (CE_FFN_CH * CE_H * 2);
        const unsigned char* g2 = reinterpret_cast<const unsigned char*>(W2p) + (size_t)cw2 * (CE_FFN_CH * 2);
        const unsigned int l1 = 16u * tid, l2 = (unsigned int)(tid >> 2) * (CE_FFN * 2) + 16u * (tid & 3);
        auto staging = [&](int n) {
            const int i = (n % 12) >> 1, id = tid + 256 * i;
            if (n < 12) {
                if ((n & 1) == 0) {
                    if (NOSTORE) asm volatile("" :: "v"(pw[i]));
                    else *reinterpret_cast<u32x4*>(w1_of(ch + 2) + (id / 48) * CE_W1_LD + (id % 48) * 8) = pw[i];
                } else pw[i] = *reinterpret_cast<const u32x4*>(g1 + (size_t)i * 4096 + l1);
            } else {
                if ((n & 1) == 0) {
                    if (NOSTORE) asm volatile("" :: "v"(pw2[i]));
                    else *reinterpret_cast<u32x4*>(w2_of(ch) + (id >> 2) * CE_W2_LD + (id & 3) * 8) = pw2[i];
                } else pw2[i] = *reinterpret_cast<const u32x4*>(g2 + (size_t)i * (64 * CE_FFN * 2) + l2);
            }
        };
        CE_STAMP(1);                                 // top of the iteration
        int n_free = 0;
#pragma unroll
        for (int j = 0; j < 48; ++j) {
            if (j >= J0 && j < J1) {
                if (j < 24) acc[j >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(j - J0) % DEPTH], __builtin_bit_cast(bf16x8, xbp[j & 1]), acc[j >> 1], 0, 0, 0);
                else xn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(j - J0) % DEPTH], hf[j - 24], xn, 0, 0, 0);
                if (j + DEPTH < J1) af[(j - J0) % DEPTH] = frag(j + DEPTH);
            }
            const bool gelu_piece = j % 3 == 0 || (j % 6 == 1 && j >= 7) || j == 47;
            if (gelu_piece) {
                if (HAS_G) {
                    if (j % 3 == 0) gelu_lookup(j / 3);
                    else if (j == 47) gelu_finish(7);
                    else gelu_finish((j - 7) / 6);
                }
            } else {
                if (HAS_B || (HAS_G && n_free >= 12)) staging(n_free);      // (the last chunk's W2 rows are still needed; W1 rows are not)
                ++n_free;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j == 23) CE_STAMP(2);                // slots 0 .. 23
        }
        CE_STAMP(3);                                 // slots 24 .. 47
        lds_barrier();
        CE_STAMP(4);                                 // barrier
        if (HAS_B) xc = xn;
        xbp[0] = xbw[0];
        xbp[1] = xbw[1];
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    iteration(0, no(), yes(), yes());
    for (int ch = 1; ch + 1 < NCH; ++ch) iteration(ch, yes(), yes(), yes());
    iteration(NCH - 1, yes(), yes(), no());
    iteration(NCH, yes(), no(), no());

    // ---- + b2, LayerNorm over the token's 384 features (this lane: 192 of them, lane ^ 32 the others)
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < 12; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nb * 32 + 8 * g + 4 * hh;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(eps_ + n);            // (LDS: two addresses per wave-instruction)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[nb][4 * g + r] + bv[r];                     // (the residual is in the accumulator already)
                acc[nb][4 * g + r] = v;
                sum += v;
            }
        }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / CE_H);
    float var = 0.f;
#pragma unroll
    for (int nb = 0; nb < 12; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = acc[nb][e] - mean; var += d * d; }
    var += __shfl_xor(var, 32, 64);
    const float rstd = rsqrtf(var * (1.f / CE_H) + ln_eps);
    // The normalised rows leave through LDS, 32 features at a time: the accumulator layout (lane = token, 16-byte pieces 32
    // bytes apart) would store 32 partial lines per instruction; read back with eight lanes per token every store
    // instruction writes eight whole 128-byte lines (fp32 row) / 64-byte runs (bf16 row).  The wave's own 32 x 36-float
    // patch of the (now idle) chunk buffers, no barrier: a wave's LDS operations execute in order.  Lanes past M hold row
    // M - 1 again and write the same bytes to it.
    {
        float* tb = reinterpret_cast<float*>(ce_smem) + wave * (32 * 36);
        const int64_t tlast = (int64_t)M - 1;
#pragma unroll
        for (int nb = 0; nb < 12; ++nb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb * 32 + 8 * g + 4 * hh;
                const f32x4 gv = *reinterpret_cast<const f32x4*>(eps_ + CE_H + n);
                const f32x4 be = *reinterpret_cast<const f32x4*>(eps_ + 2 * CE_H + n);
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (acc[nb][4 * g + r] - mean) * rstd * gv[r] + be[r];
                *reinterpret_cast<f32x4*>(tb + c * 36 + 8 * g + 4 * hh) = y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = (lane >> 3) + 8 * j, pc = lane & 7;
                const f32x4 y = *reinterpret_cast<const f32x4*>(tb + t * 36 + 4 * pc);
                int64_t row = tok0 + t;
                row = row < tlast ? row : tlast;
                bf16x4 yb;
#pragma unroll
                for (int r = 0; r < 4; ++r) yb[r] = (__bf16)y[r];
                *reinterpret_cast<f32x4*>(h32 + row * CE_H + nb * 32 + 4 * pc) = y;
                *reinterpret_cast<bf16x4*>(hb + row * CE_H + nb * 32 + 4 * pc) = yb;
            }
        }
    }
#ifdef RR_DEBUG_HARNESS
    CE_STAMP(5);                                                           // bias + residual + LayerNorm + stores
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 600)) {
        dbg_t[9] = wall_clock64() - dbg_w0;
        for (int i = 0; i < 10; ++i) ce_dbg_ffn[blockIdx.x ? 1 : 0][i] = dbg_t[i];
    }
#endif
}

// ------------------------------------------------------------------ QKV projection, token-stationary: out = bf16(h W^T + b)
// The first product of ce_ffn_fused without anything behind it: a wave's 32 tokens sit on the MFMA columns for the whole
// kernel (their bf16 rows = 24 B fragments in registers, loaded once), the weight rows stream through LDS in chunks of 32
// output features, X^T[j][tok] = W[j] . h[tok] + b[j] leaves the accumulator as bf16x4 pieces (features 8 g + 4 hh .. + 3
// of the token's row: lane halves hh = 0, 1 of a token write 16 adjacent bytes).  EIGHT waves per workgroup, two per SIMD
// (96 + 2 x 16 + staging registers: under 256): what one wave cannot overlap with its own MFMAs -- a wave issues in order,
// tools/probes/mfma_valu_probe.hip -- the other wave's MFMAs cover.  (r02 built this with one wave per SIMD: 262 us per
// layer against the tiled GEMM's 222.)  256 tokens per workgroup; a chunk's store pieces and the staging of the next
// chunks are dealt out over the 24 MFMA slots of the chunk after it.
#define CE_QKV_TOK 256
#define CE_QKV_LDS(N) (2 * CE_FFN_CH * CE_W1_LD * 2 + (N) * 4 + 8 * 32 * 80)
// EXP (debug library only, RR_CE_PROJ_EXP): bit 0 = no output, bit 1 = no staging of the next chunks, bit 2 = no barrier
// per chunk (timing only: the results are garbage); bit 3 = output rows stored with sc1 (correct results); bit 4 = the staging
// loads without their LDS stores, bit 5 = the LDS stores without the loads (garbage).
template <int DEPTH, int EXP = 0>
__global__ __launch_bounds__(512) void ce_proj_ts(const unsigned short* __restrict__ hb, int M, const unsigned short* __restrict__ W,
                                                   const float* __restrict__ bias, int N, unsigned short* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ce_smem[];
    unsigned short* wbuf = reinterpret_cast<unsigned short*>(ce_smem);          // [2][32][CE_W1_LD]
    float* bs = reinterpret_cast<float*>(wbuf + 2 * CE_FFN_CH * CE_W1_LD);      // [N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, hh = lane >> 5;
    int64_t tok = (int64_t)blockIdx.x * CE_QKV_TOK + wave * 32 + c;
    tok = tok < M ? tok : M - 1;
    bf16x8 hf[24];
#pragma unroll
    for (int s = 0; s < 24; ++s) hf[s] = *reinterpret_cast<const bf16x8*>(hb + tok * CE_H + 16 * s + 8 * hh);
    for (int i = tid; i < N / 4; i += 512) reinterpret_cast<f32x4*>(bs)[i] = reinterpret_cast<const f32x4*>(bias)[i];
    // a chunk = 32 rows x 768 B = 1536 pieces of 16 B, three per thread
    u32x4 pw[3];
    const int nch = N / CE_FFN_CH;
    // Workgroup w takes the chunks in the order rot, rot + 1, ... (mod nch): the chunks are independent output columns, so
    // the order changes no bit of the result, and at any moment the CUs ask L2 for 36 different 24 KB pieces instead of all
    // for the same one (r03 ablation: the staging of the next chunks was ~100 of the launch's 136 us)
    const int rot = (int)((blockIdx.x * 7u) % (unsigned)nch);
    auto chunk_of = [&](int k) { const int kk = k + rot; return kk >= nch ? kk - nch : kk; };
    auto load_w = [&](int k, int i) {
        const int id = tid + 512 * i;
        pw[i] = *reinterpret_cast<const u32x4*>(W + (int64_t)(chunk_of(k) * CE_FFN_CH + id / 48) * CE_H + (id % 48) * 8);
    };
    auto store_w = [&](int k, int i) {
        const int id = tid + 512 * i;
        *reinterpret_cast<u32x4*>(wbuf + (k & 1) * CE_FFN_CH * CE_W1_LD + (id / 48) * CE_W1_LD + (id % 48) * 8) = pw[i];
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) load_w(0, i);
#pragma unroll
    for (int i = 0; i < 3; ++i) store_w(0, i);
#pragma unroll
    for (int i = 0; i < 3; ++i) load_w(nch > 1 ? 1 : 0, i);
    lds_barrier();
    auto bias_of = [&](int k) {
        f32x16 v;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(bs + chunk_of(k) * CE_FFN_CH + 8 * g + 4 * hh);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * g + r] = q[r];
        }
        return v;
    };
    // a chunk's 32 tokens x 32 features leave through the wave's own LDS patch ([token][64 B + 16 pad]): written as the
    // accumulator holds them (8-byte pieces), read back with four lanes per token, so that a store instruction writes 16
    // rows x 64 contiguous bytes instead of 32 rows x 16 (the scattered form cost ~2 000 of a chunk's ~5 000 cycles in
    // the address path).  No barrier: a wave's LDS operations execute in order.
    unsigned char* tp = reinterpret_cast<unsigned char*>(bs + N) + wave * (32 * 80);
    const int64_t row0 = (int64_t)blockIdx.x * CE_QKV_TOK + wave * 32, rlast = (int64_t)M - 1;
    f32x16 prev;
    auto put_lds = [&](int g) {                      // features 8 g + 4 hh .. + 3 of the previous chunk, from `prev`
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (__bf16)prev[4 * g + r];
        *reinterpret_cast<bf16x4*>(tp + c * 80 + 16 * g + 8 * hh) = v;
    };
    auto put_rows = [&](int k, int j) {              // tokens 16 j .. 16 j + 15 of chunk k: lane = (token, 16-byte piece)
        const int t = 16 * j + (lane >> 2), pc = lane & 3;
        const u32x4 v = *reinterpret_cast<const u32x4*>(tp + t * 80 + 16 * pc);
        int64_t row = row0 + t;
        row = row < rlast ? row : rlast;             // (lanes past M hold row M - 1 again: the same bytes)
        u32x4* dst = reinterpret_cast<u32x4*>(out + row * N + chunk_of(k) * CE_FFN_CH + 8 * pc);
        if (EXP & 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");   // (A/B: rows that leave L2 at once)
        else *dst = v;
    };
    for (int k = 0; k < nch; ++k) {
        const unsigned short* ap = wbuf + (k & 1) * CE_FFN_CH * CE_W1_LD + c * CE_W1_LD + 8 * hh;
        f32x16 acc = bias_of(k);
        bf16x8 af[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ap + 16 * i);
        const int kn = k + 2 < nch ? k + 2 : nch - 1;
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % DEPTH], hf[s], acc, 0, 0, 0);
            if (s + DEPTH < 24) af[s % DEPTH] = *reinterpret_cast<const bf16x8*>(ap + 16 * (s + DEPTH));
            // slots 1, 5, 9: chunk k + 1 registers -> LDS (its home held chunk k - 1, read before the last barrier);
            // slots 3, 7, 11: chunk k + 2 global -> registers; slots 13 .. 16 and 19, 21: the previous chunk's output
            if (!(EXP & 2) && (s == 1 || s == 5 || s == 9)) {
                if (EXP & 16) asm volatile("" :: "v"(pw[(s - 1) >> 2]));          // (ablation: the loads stay, the LDS stores go)
                else store_w(k + 1, (s - 1) >> 2);
            }
            if (!(EXP & 2) && !(EXP & 32) && (s == 3 || s == 7 || s == 11)) load_w(kn, (s - 3) >> 2);
            if (!(EXP & 1) && k > 0 && s >= 13 && s <= 16) put_lds(s - 13);
            if (!(EXP & 1) && k > 0 && (s == 19 || s == 21)) put_rows(k - 1, (s - 19) >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(EXP & 4)) lds_barrier();
        prev = acc;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) put_lds(g);
    put_rows(nch - 1, 0);
    put_rows(nch - 1, 1);
}

// ------------------------------------------------------------------ attention: one workgroup per (sequence, head)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define CE_KS_LD 40         // K rows in LDS: 32 bf16 + 8 pad = 80 B (conflict-free ds_read_b128 over 16 rows)

#define CE_ATT_THREADS 512     // eight waves share one (sequence, head): K and V^T are staged once for all of them
#define CE_ATT_CH 8            // 16-key tiles per online-softmax chunk (128 keys)

__global__ __launch_bounds__(CE_ATT_THREADS, 2) void ce_attention(const unsigned short* __restrict__ qkv,
                                                                  const int32_t* __restrict__ cu,
                                                                  unsigned short* __restrict__ ctx, float scale, int smax_pad,
                                                                  unsigned short* __restrict__ ctx_cls) {
    // ctx_cls != null: LAST layer of a [CLS]-pooled output -- only query row 0 of every sequence is needed downstream
    // (pooler / CLS embedding), so only q-block 0 is computed and its row 0 lands in the compact ctx_cls[sequence]
    const int VT_LD = smax_pad + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char ce_smem[];
    unsigned short* Ks = reinterpret_cast<unsigned short*>(ce_smem);          // [smax_pad][CE_KS_LD]
    unsigned short* Vt = Ks + smax_pad * CE_KS_LD;                              // [32][VT_LD]   V transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seq = blockIdx.x, head = blockIdx.y;
    const int t0 = cu[seq], S = cu[seq + 1] - t0;
    const int Spad = (S + 31) & ~31;
    const int c16 = lane & 15, g = lane >> 4;

    // stage K (row major) and V^T of this head; rows past S are zero
    for (int c = tid; c < Spad * 4; c += CE_ATT_THREADS) {
        const int r = c >> 2, p = c & 3;
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (r < S) {
            const unsigned short* base = qkv + (int64_t)(t0 + r) * (3 * CE_H) + head * CE_HD + p * 8;
            kv = *reinterpret_cast<const u32x4*>(base + CE_H);
            vv = *reinterpret_cast<const u32x4*>(base + 2 * CE_H);
        }
        *reinterpret_cast<u32x4*>(Ks + r * CE_KS_LD + p * 8) = kv;
        const unsigned short* ve = reinterpret_cast<const unsigned short*>(&vv);
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(p * 8 + j) * VT_LD + r] = ve[j];
    }
    __syncthreads();

    const int n_tiles = Spad >> 4;                       // even
    // softmax in base 2 with the 1/sqrt(d) scale folded in: p = exp2(c1 * s - c1 * max), streamed over chunks of
    // 128 keys with a running maximum (raw scores; c1 > 0) and running sums, the accumulators rescaled per chunk
    const float c1 = scale * 1.4426950408889634f;
    const int q_limit = ctx_cls ? 1 : S;
    for (int qb = wave; qb * 16 < q_limit; qb += CE_ATT_THREADS / 64) {
        // B operand of S^T = K Q^T: lane (q = c16, g) holds Q[q][8g .. 8g+7]
        int qrow = qb * 16 + c16;
        qrow = qrow < S ? qrow : S - 1;
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qkv + (int64_t)(t0 + qrow) * (3 * CE_H) + head * CE_HD + g * 8);
        float m_run = -INFINITY, l_run = 0.f;             // of query column c16 (l: this lane's keys only until the end)
        f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < n_tiles; kc += CE_ATT_CH) {
            f32x4_t st[CE_ATT_CH];
            float cm = -INFINITY;
#pragma unroll
            for (int i = 0; i < CE_ATT_CH; ++i) {
                const int kt = kc + i;
                if (kt < n_tiles) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 16 + c16) * CE_KS_LD + g * 8);
                    f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, z, 0, 0, 0);   // z[r] = S^T[key 16kt + 4g + r][q = c16]
                    if (kt * 16 + 16 > S) {                                            // only the last tiles hold padding keys
#pragma unroll
                        for (int r = 0; r < 4; ++r) z[r] = (kt * 16 + 4 * g + r) < S ? z[r] : -INFINITY;
                    }
                    cm = fmaxf(fmaxf(cm, fmaxf(z[0], z[1])), fmaxf(z[2], z[3]));
                    st[i] = z;
                }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float m_new = fmaxf(m_run, cm);              // finite: key 0 of chunk 0 is never masked
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c1);   // exp2(-inf) = 0 on the first chunk
            const float nmx = -m_new * c1;
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                      // accumulator row q = 4g + r: the factor of that query
                const float a = __shfl(alpha, 4 * g + r, 64);
                o0[r] *= a;
                o1[r] *= a;
            }
            // out[q][d] += sum_k P[q][k] V[k][d]: the probability tiles are the A operand as they sit (lane = q column,
            // registers = keys); two 16-key tiles form one K = 32 step whose key order is
            //   element j of lane group g:  j < 4 -> key 32u + 4g + j,   j >= 4 -> key 32u + 16 + 4g + (j - 4)
            // and V^T is read in that same order.
#pragma unroll
            for (int i = 0; i < CE_ATT_CH; i += 2) {
                const int kt = kc + i;
                if (kt < n_tiles) {
                    bf16x8 pf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i][r], c1, nmx));
                        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i + 1][r], c1, nmx));
                        l_run += p0 + p1;
                        pf[r] = (__bf16)p0;
                        pf[4 + r] = (__bf16)p1;
                    }
                    const unsigned short* vb = Vt + c16 * VT_LD + 16 * kt + 4 * g;
                    const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(vb);
                    const bf16x4 a1 = *reinterpret_cast<const bf16x4*>(vb + 16);
                    const bf16x4 b0 = *reinterpret_cast<const bf16x4*>(vb + 16 * VT_LD);
                    const bf16x4 b1 = *reinterpret_cast<const bf16x4*>(vb + 16 * VT_LD + 16);
                    bf16x8 v0, v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] = a0[r]; v0[4 + r] = a1[r]; v1[r] = b0[r]; v1[4 + r] = b1[r]; }
                    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, v0, o0, 0, 0, 0);    // o[r] = out[q = 4g + r][d = c16]
                    o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, v1, o1, 0, 0, 0);    //                     [d = 16 + c16]
                }
            }
        }
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qb * 16 + 4 * g + r;
            const float inv = 1.f / __shfl(l_run, 4 * g + r, 64);          // lane 4g + r holds the sum of query column 4g + r
            if (q < q_limit) {
                unsigned short* dst = (ctx_cls ? ctx_cls + (int64_t)seq * CE_H : ctx + (int64_t)(t0 + q) * CE_H) + head * CE_HD;
                dst[c16] = ce_bf16_bits(o0[r] * inv);
                dst[16 + c16] = ce_bf16_bits(o1[r] * inv);
            }
        }
    }
}

// ------------------------------------------------------------------ pooler + classifier on the [CLS] rows (fp32)
__global__ __launch_bounds__(256) void ce_head(const float* __restrict__ h32, const int32_t* __restrict__ cu,
                                               const float* __restrict__ wp, const float* __restrict__ bp,
                                               const float* __restrict__ wc, const float* __restrict__ bc, int n_labels,
                                               int mode, float* __restrict__ out, const unsigned* __restrict__ range_flag = nullptr) {
    __shared__ float x[CE_H], pooled[CE_H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, seq = blockIdx.x;
    if (range_flag && *range_flag) {                       // (rr_ce_h2.hip: a value left the fp16 range: no finite answer)
        const int n = mode == 1 ? CE_H : n_labels;
        for (int c = tid; c < n; c += 256) out[(int64_t)seq * n + c] = __builtin_nanf("");
        return;
    }
    const float* src = h32 + (int64_t)(cu ? cu[seq] : seq) * CE_H;       // cu == null: h32 is compact, one row per sequence
    for (int c = tid; c < CE_H; c += 256) x[c] = src[c];
    __syncthreads();
    if (mode == 1) {                                       // RR_CE_OUT_CLS: last_hidden_state[:, 0]
        for (int c = tid; c < CE_H; c += 256) out[(int64_t)seq * CE_H + c] = x[c];
        return;
    }
    // pooled[j] = tanh(Wp[j] . x + bp[j]); a wave takes rows j = wave, wave + 4, ...; eight rows' loads in flight at a time
    // (one row at a time the kernel was a chain of dependent L2 round trips: 75 us for 256 sequences), each row's own sum in
    // the order it always had
    for (int j0 = wave; j0 < CE_H; j0 += 32) {
        float s8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + 4 * u;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) s = __builtin_fmaf(wp[(int64_t)j * CE_H + lane + 64 * i], x[lane + 64 * i], s);
            s8[u] = s;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float s = ce_wave_sum(s8[u]);
            if (lane == 0) pooled[j0 + 4 * u] = tanhf(s + bp[j0 + 4 * u]);
        }
    }
    __syncthreads();
    for (int l = wave; l < n_labels; l += 4) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) s = __builtin_fmaf(wc[(int64_t)l * CE_H + lane + 64 * i], pooled[lane + 64 * i], s);
        s = ce_wave_sum(s);
        if (lane == 0) out[(int64_t)seq * n_labels + l] = s + bc[l];
    }
}

// compact residual rows for the [CLS]-only tail of the last layer: dst[seq] = h32[cu[seq]]
__global__ __launch_bounds__(128) void ce_gather_cls(const float* __restrict__ h32, const int32_t* __restrict__ cu,
                                                     float* __restrict__ dst) {
    const int seq = blockIdx.x;
    const float* src = h32 + (int64_t)cu[seq] * CE_H;
    for (int c = threadIdx.x; c < CE_H; c += 128) dst[(int64_t)seq * CE_H + c] = src[c];
}

__global__ void ce_to_bf16(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ce_bf16_bits(src[i]);
}


// ------------------------------------------------------------------ reference-precision mode (RR_CE_PRECISION_F32)
// The reference runs both encoders in fp32 torch (app/app_product_search.py:250-251, 277-278).  The kernels above multiply
// in bf16 (2.5e-2 on logits of O(1)): the fast path.  This mode keeps every operand fp32 end to end -- fp32 weights, fp32
// activations -- so that logits and embeddings agree with the `transformers` fixtures to fp32 rounding (1e-5,
// tests/test_gpu_k5.py).  Three generations of its products live in the library:
//   * rr_ce_h2.hip (default since round 4): operands as two fp16 numbers, three MFMA products per fp32 product -- 13 ms per
//     256 x 512 tokens; fp16's range guarded by a device flag (ce_forward_h2 below);
//   * this file, ce_gemm_x3 / ce_attention_x3: operands as three bf16 terms, six products, ANY fp32 range -- 23.9 ms; what
//     rr_ce_set_wide_range switches a handle to when the flag came up, and RR_CE_F32_SPLIT=bf16x3 for the A/B;
//   * this file, ce_gemm_f32 / ce_attention_f32: the fp32-input matrix instruction v_mfma_f32_32x32x2_f32 (the chip's
//     vector-rate matrix path, 157 TF/s) -- 35 ms; RR_CE_F32_MFMA=1 / RR_CE_F32_ATT_MFMA32=1 (A/B only).
typedef float f32x16r __attribute__((ext_vector_type(16)));

// out[M][N] = A[M][K] W[N][K]^T + bias (+ exact GELU); A, W, out fp32 row-major; N % 128 == 0, K % 16 == 0.
// Workgroup = 4 waves (2 x 2), 128 x 128 outputs, wave = 64 x 64 = 2 x 2 MFMA tiles; K tiles of 16 through LDS, stored
// k-major ([k][row], rows padded to 132 floats: conflict-free for the ds_write_b32 of the staging and the ds_read_b32 of
// the operands).  K % 16 == 0.  MFMA operands: A lane l = row (l & 31), k = l >> 5; B lane l = column (l & 31), k = l >> 5; C register
// 4 g + i = row 8 g + 4 (l >> 5) + i, column l & 31.
#define CE_F32_LD 132
// fp32 GEMM on the bf16 matrix cores by operand splitting (the arithmetic of the exact K1a' scans, csrc/rr_x3.h): every fp32
// operand x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (the subtractions are exact, the
// three terms carry 24 bits), and a product keeps the six term pairs down to 2^-16 of it: hi*hi, hi*mid, mid*hi, hi*lo,
// mid*mid, lo*hi -- what is dropped is below 2^-23 of the product, the rounding of one fp32 multiply; accumulation in fp32 in
// the MFMA.  Six v_mfma_f32_32x32x16_bf16 per 16 k instead of eight v_mfma_f32_32x32x2_f32: 2.6 x the matrix rate at peak.
// Same tile as ce_gemm_f32 (128 x 128 per workgroup, four waves of 2 x 2 MFMA blocks); a thread's eight consecutive k of one
// row are exactly one lane's fragment: split in registers, one 16-byte LDS store per term, conflict-free 16-byte reads.
template <bool GELU>
__global__ __launch_bounds__(256) void ce_gemm_x3(const float* __restrict__ A, const float* __restrict__ W,
                                                  const float* __restrict__ bias, int M, int N, int K,
                                                  float* __restrict__ out) {
    // (K tiles of 32 -- two k-steps per LDS fill -- measured slower: 172 registers, one wave per SIMD fewer)
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * 2 * 128 * 8], Ws[3 * 2 * 128 * 8];   // [term][k half][row][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int srow = tid >> 1, skq = tid & 1;                 // staging: row of the tile, k half (8 consecutive k)
    int arow = m0 + srow;
    arow = arow < M ? arow : M - 1;                           // (rows past the end: clamped loads, masked stores)
    const float* ap = A + (int64_t)arow * K + 8 * skq;
    const float* wp = W + (int64_t)(n0 + srow) * K + 8 * skq;
    f32x16r acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r = lane & 31, kh = lane >> 5;
    f32x4 av[2], wv[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        av[h2] = *reinterpret_cast<const f32x4*>(ap + 4 * h2);
        wv[h2] = *reinterpret_cast<const f32x4*>(wp + 4 * h2);
    }
    auto split_store = [&](const f32x4 (&v)[2], unsigned short* dst) {       // eight fp32 -> three bf16x8 terms
        bf16x8 t0, t1, t2;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[h2][e];
                const __bf16 hi = (__bf16)x;
                const float r1 = x - (float)hi;
                const __bf16 mid = (__bf16)r1;
                const float r2 = r1 - (float)mid;
                t0[4 * h2 + e] = hi; t1[4 * h2 + e] = mid; t2[4 * h2 + e] = (__bf16)r2;
            }
        *reinterpret_cast<bf16x8*>(dst + ((0 * 2 + skq) * 128 + srow) * 8) = t0;
        *reinterpret_cast<bf16x8*>(dst + ((1 * 2 + skq) * 128 + srow) * 8) = t1;
        *reinterpret_cast<bf16x8*>(dst + ((2 * 2 + skq) * 128 + srow) * 8) = t2;
    };
    for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();                                      // the previous tile has been read
        split_store(av, As);
        split_store(wv, Ws);
        __syncthreads();
        if (k0 + 16 < K) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                av[h2] = *reinterpret_cast<const f32x4*>(ap + k0 + 16 + 4 * h2);
                wv[h2] = *reinterpret_cast<const f32x4*>(wp + k0 + 16 + 4 * h2);
            }
        }
        bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[t][i] = *reinterpret_cast<const bf16x8*>(As + ((t * 2 + kh) * 128 + wm + 32 * i + r) * 8);
                b[t][i] = *reinterpret_cast<const bf16x8*>(Ws + ((t * 2 + kh) * 128 + wn + 32 * i + r) * 8);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
            }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + r;
            const float bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + 8 * (e >> 2) + 4 * kh + (e & 3);
                if (row < M) {
                    float x = acc[i][j][e] + bv;
                    if (GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));      // gelu(x) = x Phi(x), erf form (HF "gelu")
                    out[(int64_t)row * N + col] = x;
                }
            }
        }
}

template <bool GELU>
__global__ __launch_bounds__(256) void ce_gemm_f32(const float* __restrict__ A, const float* __restrict__ W,
                                                   const float* __restrict__ bias, int M, int N, int K,
                                                   float* __restrict__ out) {
    // K tiles of 16, k-major in LDS; the next tile's global loads are issued before the current tile's MFMAs (their latency
    // runs under 32 MFMAs per wave)
    __shared__ float As[16 * CE_F32_LD], Ws[16 * CE_F32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int srow = tid >> 1, skq = tid & 1;                 // staging: row of the tile, float4s 4 skq and 4 skq + 8 of its 16 k
    int arow = m0 + srow;
    arow = arow < M ? arow : M - 1;                           // (rows past the end: clamped loads, masked stores)
    const float* ap = A + (int64_t)arow * K + 4 * skq;
    const float* wp = W + (int64_t)(n0 + srow) * K + 4 * skq;
    f32x16r acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r = lane & 31, kh = lane >> 5;
    f32x4 av[2], wv[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        av[h2] = *reinterpret_cast<const f32x4*>(ap + 8 * h2);
        wv[h2] = *reinterpret_cast<const f32x4*>(wp + 8 * h2);
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        __syncthreads();                                      // the previous tile has been read
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int kb = (4 * skq + 8 * h2) * CE_F32_LD + srow;
            As[kb] = av[h2].x; As[kb + CE_F32_LD] = av[h2].y; As[kb + 2 * CE_F32_LD] = av[h2].z; As[kb + 3 * CE_F32_LD] = av[h2].w;
            Ws[kb] = wv[h2].x; Ws[kb + CE_F32_LD] = wv[h2].y; Ws[kb + 2 * CE_F32_LD] = wv[h2].z; Ws[kb + 3 * CE_F32_LD] = wv[h2].w;
        }
        __syncthreads();
        if (k0 + 16 < K) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                av[h2] = *reinterpret_cast<const f32x4*>(ap + k0 + 16 + 8 * h2);
                wv[h2] = *reinterpret_cast<const f32x4*>(wp + k0 + 16 + 8 * h2);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {                      // k pairs: lanes < 32 hold k = 2 kk, lanes >= 32 k = 2 kk + 1
            const int kr = (2 * kk + kh) * CE_F32_LD;
            const float a0 = As[kr + wm + r], a1 = As[kr + wm + 32 + r];
            const float b0 = Ws[kr + wn + r], b1 = Ws[kr + wn + 32 + r];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + r;
            const float bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + 8 * (e >> 2) + 4 * kh + (e & 3);
                if (row < M) {
                    float x = acc[i][j][e] + bv;
                    if (GELU) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));      // gelu(x) = x Phi(x), erf form (HF "gelu")
                    out[(int64_t)row * N + col] = x;
                }
            }
        }
}

// h32[t] = LayerNorm(y[t] + h32[t]) * g + b  (BertSelfOutput / BertOutput: dense -> + residual -> LayerNorm), one wave per token
__global__ __launch_bounds__(256) void ce_add_ln_f32(const float* __restrict__ y, float* __restrict__ h32, int T,
                                                     const float* __restrict__ g, const float* __restrict__ b, float eps) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float x[6];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        x[i] = y[(int64_t)t * CE_H + c] + h32[(int64_t)t * CE_H + c];
        s += x[i];
    }
    const float mean = ce_wave_sum(s) * (1.f / CE_H);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) { const float d = x[i] - mean; v += d * d; }
    const float rstd = 1.0f / sqrtf(ce_wave_sum(v) * (1.f / CE_H) + eps);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = lane + 64 * i;
        h32[(int64_t)t * CE_H + c] = (x[i] - mean) * rstd * g[c] + b[c];
    }
}

// softmax(Q K^T / sqrt(32)) V of one (sequence, head) in fp32 on the fp32-input matrix instruction.  K (rows padded to 33
// floats: the A operand below reads 32 consecutive keys at one dim) and V of the head sit in LDS as fp32; a wave owns tiles of
// 32 queries.  Per 32-key tile:  S^T = K Q^T  (A = K: lane l = key l & 31, dim 2 s + (l >> 5); B = Q^T held in 16 registers
// per lane for the whole query tile), so a lane holds ONE query's scores against 16 keys (C layout: register 4 g + i = key
// 8 g + 4 h + i, h = l >> 5): the running maximum / sum of the online softmax are per-lane values plus one exchange with
// lane ^ 32, and the probabilities are the B operand of  O^T += V^T P^T  as they sit (A = V^T: lane l = dim l & 31, key of
// register t in its half) -- no lane movement, no LDS round trip.  exp is expf: fp32 rounding apart, the two-pass softmax.
#define CE_F32_KLD 33
__global__ __launch_bounds__(256) void ce_attention_f32(const float* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                        float* __restrict__ ctx, float scale) {
    extern __shared__ float kv[];                             // K [Sp][33], V [Sp][32], Sp = S rounded up to 32
    const int seq = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = cu[seq], S = cu[seq + 1] - t0;
    const int Sp = (S + 31) & ~31;
    float* Ks = kv;
    float* Vs = kv + (size_t)Sp * CE_F32_KLD;
    for (int i = tid; i < Sp * (CE_HD / 4); i += 256) {
        const int j = i / (CE_HD / 4), c4 = i % (CE_HD / 4);
        f32x4 k4 = {0.f, 0.f, 0.f, 0.f}, v4 = {0.f, 0.f, 0.f, 0.f};       // rows past the sequence: zeros (their scores are masked)
        if (j < S) {
            const float* row = qkv + (int64_t)(t0 + j) * (3 * CE_H) + head * CE_HD + 4 * c4;
            k4 = *reinterpret_cast<const f32x4*>(row + CE_H);
            v4 = *reinterpret_cast<const f32x4*>(row + 2 * CE_H);
        }
        float* kd = Ks + j * CE_F32_KLD + 4 * c4;
        kd[0] = k4.x; kd[1] = k4.y; kd[2] = k4.z; kd[3] = k4.w;
        *reinterpret_cast<f32x4*>(Vs + j * CE_HD + 4 * c4) = v4;
    }
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const float scale2 = scale * 1.4426950408889634f;        // the softmax in base 2: no range reduction per probability
    for (int q0 = 32 * wave; q0 < S; q0 += 128) {
        int qrow = q0 + c;
        qrow = qrow < S ? qrow : S - 1;                       // (lanes past the sequence: a valid row, never stored)
        const float* qp = qkv + (int64_t)(t0 + qrow) * (3 * CE_H) + head * CE_HD + h;
        float qreg[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) qreg[st] = qp[2 * st];
        f32x16r o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
        float mx = -INFINITY, l = 0.f;
        for (int j0 = 0; j0 < Sp; j0 += 32) {
            f32x16r sT;
#pragma unroll
            for (int e = 0; e < 16; ++e) sT[e] = 0.f;
            const float* kp = Ks + (j0 + c) * CE_F32_KLD + h;
#pragma unroll
            for (int st = 0; st < 16; ++st) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * st], qreg[st], sT, 0, 0, 0);
            float cm = -INFINITY;
            if (j0 + 32 > S) {                                // only the last tile holds padding keys
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = j0 + 8 * (e >> 2) + 4 * h + (e & 3);
                    sT[e] = key < S ? sT[e] * scale2 : -INFINITY;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) sT[e] *= scale2;             // scores in units of log2 e: p = exp2(v - max)
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) cm = fmaxf(cm, sT[e]);
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));           // the query's other 16 keys of this tile
            if (cm > mx) {                                    // (per lane: a query's two lanes decide alike)
                const float r = __builtin_amdgcn_exp2f(mx - cm);     // first tile: exp2(-inf) = 0 on zero sums
                l *= r;
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] *= r;
                mx = cm;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f(sT[e] - mx);      // masked keys: exp2(-inf) = 0 (v_exp_f32: 1 ulp)
                sT[e] = pe;
                l += pe;
            }
            const float* vp = Vs + (j0 + 4 * h) * CE_HD + c;
#pragma unroll
            for (int e = 0; e < 16; ++e)                      // key of register e in this lane half: 8 (e >> 2) + 4 h + (e & 3)
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[(8 * (e >> 2) + (e & 3)) * CE_HD], sT[e], o, 0, 0, 0);
        }
        l += __shfl_xor(l, 32, 64);
        if (q0 + c < S) {
            const float inv = 1.0f / l;
            float* op = ctx + (int64_t)(t0 + q0 + c) * CE_H + head * CE_HD + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)                       // register 4 g + i = dim 8 g + 4 h + i: 16 contiguous bytes per g
                *reinterpret_cast<f32x4*>(op + 8 * g) = f32x4{o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
        }
    }
}

// The same attention on the bf16 matrix cores by operand splitting (what ce_gemm_x3 does for the four GEMMs): every fp32
// operand -- K, V, Q and the probabilities P -- is x = hi + mid + lo in three bf16 terms (24 bits), a product keeps the six
// term pairs down to 2^-16 of it, fp32 accumulation in v_mfma_f32_32x32x16_bf16: 12 MFMAs of 8 passes per 32 x 32 x 32 block
// instead of 16 v_mfma_f32_32x32x2_f32 of 16 passes (2.7x the matrix rate), and two waves per SIMD so that one wave's softmax
// (scale, max, exp2, the split of P: ~170 vector instructions per 32 x 32 tile) runs under the other's MFMAs.
//   * workgroup = one (sequence, head), 8 waves; wave w owns query tiles w and w + 8 (32 queries each: sequences up to 512);
//   * keys go through LDS in chunks of 256: K as three bf16 terms [term][key][32 dims] (rows padded to 40: the A-operand
//     ds_read_b128 of 32 keys x 8 dims is conflict-free), V transposed as three terms [term][dim][256 key slots] (rows padded
//     to 264); the online softmax state of a wave's two query tiles lives in registers across the chunks;
//   * S^T = K Q^T: A = K terms from LDS, B = Q terms (split in registers once per query tile and chunk); C layout as in
//     ce_attention_f32: a lane holds one query's scores against 16 keys, register 4 g + i = key 8 g + 4 h + i (h = l >> 5);
//   * O^T += V^T P^T: the B operand of k-step s wants, in lane (query, h), the probabilities of 8 keys -- registers 8 s .. 8 s
//     + 7 AS THEY SIT, i.e. keys 16 s + 8 a + 4 h + i (a = 0, 1); the product sums over keys, so any order does as long as V^T
//     uses the same one: key 16 s + 8 a + 4 h + i of a 32-key tile is stored at slot 16 s + 8 h + 4 a + i (bits 2 and 3 of
//     the key swapped), and the A operand of lane (dim, h) is one 16-byte read.  No lane movement, no LDS round trip for P.
// Scores, maxima, exp2 and sums stay fp32 vector arithmetic, as in ce_attention_f32: same 1e-5 bars (tests/test_gpu_k5.py).
// RR_CE_F32_ATT_MFMA32=1 keeps ce_attention_f32 (A/B).
#define CE_X3A_KC 256                                          // keys per LDS chunk
#define CE_X3A_KLD 40                                          // bf16 per K row (32 + 8 pad)
#define CE_X3A_VLD (CE_X3A_KC + 8)                             // bf16 per V^T row
#define CE_X3A_LDS ((3 * CE_X3A_KC * CE_X3A_KLD + 3 * CE_HD * CE_X3A_VLD) * 2)
struct ce_bf16x3 { __bf16 hi, mid, lo; };
__device__ __forceinline__ ce_bf16x3 ce_split3(float x) {
    ce_bf16x3 r;
    r.hi = (__bf16)x;
    const float r1 = x - (float)r.hi;
    r.mid = (__bf16)r1;
    r.lo = (__bf16)(r1 - (float)r.mid);
    return r;
}
// H2OUT: the context leaves as an h2 image (rr_ce_h2.h; chunk = head * 4 + g, half h of the unit) for ce_gemm_h2.
template <bool H2OUT>
__global__ __launch_bounds__(512, 2) void ce_attention_x3(const float* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                          float* __restrict__ ctx, float scale, u32x2* __restrict__ ctx2 = nullptr,
                                                          int64_t os = 0, unsigned* __restrict__ flag = nullptr, int cls_only = 0) {
    // cls_only (the last layer of a [CLS]-pooled output): only query 0 of the sequence is wanted -- one query tile, by wave 0,
    // and its context row goes to row `seq` of a compact image (one row per sequence)
    extern __shared__ __attribute__((aligned(16))) unsigned short x3a_lds[];
    unsigned short* Kt = x3a_lds;                                          // [3][KC][KLD]
    unsigned short* Vt = x3a_lds + 3 * CE_X3A_KC * CE_X3A_KLD;             // [3][32][VLD]
    const int seq = blockIdx.x, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = cu[seq], S = cu[seq + 1] - t0;
    const int c = lane & 31, h = lane >> 5;
    const float scale2 = scale * 1.4426950408889634f;        // the softmax in base 2
    constexpr int NT = 2;                                     // query tiles per wave: w and w + 8
    f32x16r o[NT];
    float mx[NT], l[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        mx[t] = -INFINITY;
        l[t] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    }
    for (int k0 = 0; k0 < S; k0 += CE_X3A_KC) {
        const int kn = S - k0 < CE_X3A_KC ? S - k0 : CE_X3A_KC;           // keys of this chunk
        const int knp = (kn + 31) & ~31;
        __syncthreads();                                      // the previous chunk has been read
        // ---- K: thread -> (key, 4 dims): three 8-byte stores
        for (int i = tid; i < knp * 8; i += 512) {
            const int j = i >> 3, c4 = i & 7;
            f32x4 k4 = {0.f, 0.f, 0.f, 0.f};
            if (j < kn) k4 = *reinterpret_cast<const f32x4*>(qkv + (int64_t)(t0 + k0 + j) * (3 * CE_H) + CE_H + head * CE_HD + 4 * c4);
            __bf16 t3[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const ce_bf16x3 x3 = ce_split3(k4[e]);
                t3[0][e] = x3.hi; t3[1][e] = x3.mid; t3[2][e] = x3.lo;
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<bf16x4*>(Kt + (t * CE_X3A_KC + j) * CE_X3A_KLD + 4 * c4) = bf16x4{t3[t][0], t3[t][1], t3[t][2], t3[t][3]};
            }
        }
        // ---- V^T: thread -> (key pair 2 m, 2 m + 1; 4 dims): per dim and term one 4-byte store at the pair's slot
        for (int i = tid; i < (knp >> 1) * 8; i += 512) {
            const int m = i >> 3, c4 = i & 7;
            const int j = 2 * m;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (j < kn) va = *reinterpret_cast<const f32x4*>(qkv + (int64_t)(t0 + k0 + j) * (3 * CE_H) + 2 * CE_H + head * CE_HD + 4 * c4);
            if (j + 1 < kn) vb = *reinterpret_cast<const f32x4*>(qkv + (int64_t)(t0 + k0 + j + 1) * (3 * CE_H) + 2 * CE_H + head * CE_HD + 4 * c4);
            // slot of key j inside its 32-key tile: bits 2 and 3 swapped
            const int in = j & 31;
            const int slot = (j & ~31) + (in & 0x13) + ((in & 4) << 1) + ((in & 8) >> 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const ce_bf16x3 xa = ce_split3(va[e]), xb = ce_split3(vb[e]);
                const __bf16 a3[3] = {xa.hi, xa.mid, xa.lo}, b3[3] = {xb.hi, xb.mid, xb.lo};
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<bf16x2v*>(Vt + (t * CE_HD + 4 * c4 + e) * CE_X3A_VLD + slot) = bf16x2v{a3[t], b3[t]};
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int q0 = 32 * (wave + 8 * t);
            if (q0 >= S || (cls_only && q0 != 0)) continue;   // (wave-uniform)
            int qrow = q0 + c;
            qrow = qrow < S ? qrow : S - 1;                   // (lanes past the sequence: a valid row, never stored)
            // ---- Q terms of this tile: lane (query c, h), k-step s: dims 16 s + 8 h .. + 7
            bf16x8 qb[3][2];
            {
                const float* qp = qkv + (int64_t)(t0 + qrow) * (3 * CE_H) + head * CE_HD + 8 * h;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f32x4 q0v = *reinterpret_cast<const f32x4*>(qp + 16 * ks), q1v = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const ce_bf16x3 x0 = ce_split3(q0v[e]), x1 = ce_split3(q1v[e]);
                        qb[0][ks][e] = x0.hi; qb[1][ks][e] = x0.mid; qb[2][ks][e] = x0.lo;
                        qb[0][ks][4 + e] = x1.hi; qb[1][ks][4 + e] = x1.mid; qb[2][ks][4 + e] = x1.lo;
                    }
                }
            }
            for (int j0 = 0; j0 < knp; j0 += 32) {
                // ---- S^T = K Q^T (smallest terms first)
                bf16x8 ka[3][2];
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        ka[tt][ks] = *reinterpret_cast<const bf16x8*>(Kt + (tt * CE_X3A_KC + j0 + c) * CE_X3A_KLD + 16 * ks + 8 * h);
                f32x16r sT;
#pragma unroll
                for (int e = 0; e < 16; ++e) sT[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[2][ks], qb[0][ks], sT, 0, 0, 0);
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0][ks], qb[2][ks], sT, 0, 0, 0);
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[1][ks], qb[1][ks], sT, 0, 0, 0);
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[1][ks], qb[0][ks], sT, 0, 0, 0);
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0][ks], qb[1][ks], sT, 0, 0, 0);
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[0][ks], qb[0][ks], sT, 0, 0, 0);
                }
                float cm = -INFINITY;
                if (k0 + j0 + 32 > S) {                       // only the sequence's last tile holds padding keys
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = k0 + j0 + 8 * (e >> 2) + 4 * h + (e & 3);
                        sT[e] = key < S ? sT[e] * scale2 : -INFINITY;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) sT[e] *= scale2;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) cm = fmaxf(cm, sT[e]);
                cm = fmaxf(cm, __shfl_xor(cm, 32, 64));       // the query's other 16 keys of this tile
                if (cm > mx[t]) {                             // (per lane: a query's two lanes decide alike)
                    const float r = __builtin_amdgcn_exp2f(mx[t] - cm);
                    l[t] *= r;
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[t][e] *= r;
                    mx[t] = cm;
                }
                // ---- probabilities, split as they sit: registers 8 s .. 8 s + 7 = the B operand of k-step s
                bf16x8 pb[3][2];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(sT[e] - mx[t]);      // masked keys: exp2(-inf) = 0
                    l[t] += pe;
                    const ce_bf16x3 p3 = ce_split3(pe);
                    pb[0][e >> 3][e & 7] = p3.hi; pb[1][e >> 3][e & 7] = p3.mid; pb[2][e >> 3][e & 7] = p3.lo;
                }
                // ---- O^T += V^T P^T: A = V^T terms, lane (dim c, h), k-step s: slots 16 s + 8 h .. + 7 of this tile
                bf16x8 va[3][2];
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        va[tt][ks] = *reinterpret_cast<const bf16x8*>(Vt + (tt * CE_HD + c) * CE_X3A_VLD + j0 + 16 * ks + 8 * h);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[2][ks], pb[0][ks], o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[0][ks], pb[2][ks], o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[1][ks], pb[1][ks], o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[1][ks], pb[0][ks], o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[0][ks], pb[1][ks], o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[0][ks], pb[0][ks], o[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q0 = 32 * (wave + 8 * t);
        const float lt = l[t] + __shfl_xor(l[t], 32, 64);
        if (q0 + c < S && (!cls_only || q0 + c == 0)) {
            const float inv = 1.0f / lt;
            if (H2OUT) {
                bool bad = false;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    f16x4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = o[t][4 * g + i] * inv;
                        hi[i] = (_Float16)v;                                                                 // (h2_split, rr_ce_h2.hip)
                        lo[i] = (_Float16)((v - (float)hi[i]) * CE_H2_SCALE);
                        bad |= !(__builtin_fabsf(v) <= 65504.f);
                    }
                    const int64_t u = (int64_t)(head * 4 + g) * os + (cls_only ? seq : t0 + q0 + c);
                    ctx2[2 * u + h] = __builtin_bit_cast(u32x2, hi);
                    ctx2[2 * ((int64_t)(CE_H / 8) * os + u) + h] = __builtin_bit_cast(u32x2, lo);
                }
                if (bad) atomicOr(flag, 1u);
            } else {
                float* op = ctx + (int64_t)(t0 + q0 + c) * CE_H + head * CE_HD + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g)                       // register 4 g + i = dim 8 g + 4 h + i: 16 contiguous bytes per g
                    *reinterpret_cast<f32x4*>(op + 8 * g) = f32x4{o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv};
            }
        }
    }
}

// ------------------------------------------------------------------ host side
struct rr_ce_layer {
    unsigned short *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;      // bf16 [N][K]
    unsigned short* w2p = nullptr;   // w2 with the columns of every 16-group in the fused FFN's accumulator-operand order
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    float *wqkv32 = nullptr, *wo32 = nullptr, *w1_32 = nullptr, *w2_32 = nullptr;     // RR_CE_PRECISION_F32: the Linear weights as given
    void *wqkv_h2 = nullptr, *wo_h2 = nullptr, *w1_h2 = nullptr, *w2_h2 = nullptr;      // ... and as h2 images (rr_ce_h2.h)
};

struct rr_ce {
    int device = 0;
    rr_ce_config cfg;
    float *word = nullptr, *pos = nullptr, *type = nullptr, *eln_g = nullptr, *eln_b = nullptr;
    rr_ce_layer* layers = nullptr;
    float *wp = nullptr, *bp = nullptr, *wc = nullptr, *bc = nullptr;
    float* gelu_tab = nullptr;       // ce_gelu_table: the fused FFN's Phi table
    // activation scratch for `cap` tokens
    int64_t cap = 0;
    float* h32 = nullptr;
    unsigned short *hb = nullptr, *qkv = nullptr, *ctx = nullptr, *inter = nullptr;
    // compact [sequence] buffers of the last layer's [CLS]-only tail
    int64_t cap_seqs = 0;
    float* h32c = nullptr;
    unsigned short *hbc = nullptr, *ctxc = nullptr, *interc = nullptr;
    // RR_CE_PRECISION_F32: fp32 activations
    int64_t cap32 = 0;
    float *qkv32 = nullptr, *y32 = nullptr, *inter32 = nullptr;
    int64_t cap32_seqs = 0;                    // ... and of the last layer's [CLS]-only tail: one row per sequence
    float *h32c32 = nullptr, *y32c = nullptr;
    void *hxc = nullptr, *ctxhc = nullptr, *interhc = nullptr;
    void *hx = nullptr, *ctxh = nullptr;      // h2 images of the residual stream / the attention context (inter32 doubles as the FFN's)
    unsigned* d_flag = nullptr;                // OR-ed with 1 by a producer that met a value outside fp16's range
    bool wide_range = false;                   // rr_ce_set_wide_range: the bf16 three-term kernels (any fp32 range)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::mutex mu;
};

// Phi(x) = (1 + erf(x / sqrt 2)) / 2 at the CE_GELU_N + 1 knots of [-CE_GELU_R, CE_GELU_R], as (value, step to the next knot)
// pairs: ce_ffn_fused interpolates linearly (max error h^2 / 8 max|Phi''| = 9.4e-6) and clamps outside (Phi(-4.5) = 3.4e-6)
static void ce_gelu_table(float* tab) {
    const double h = 2.0 * CE_GELU_R / CE_GELU_N;
    for (int i = 0; i < CE_GELU_N; ++i) {
        const double a = 0.5 * (1.0 + std::erf((-CE_GELU_R + i * h) * 0.70710678118654752440));
        const double b = 0.5 * (1.0 + std::erf((-CE_GELU_R + (i + 1) * h) * 0.70710678118654752440));
        tab[2 * i] = (float)a;
        tab[2 * i + 1] = (float)(b - a);
    }
}

static int ce_upload_f32(float** dst, const float* src, size_t n) {
    *dst = nullptr;
    RR_HIP_TRY(hipMalloc((void**)dst, sizeof(float) * (n ? n : 1)));
    RR_HIP_TRY(hipMemcpy(*dst, src, sizeof(float) * n, hipMemcpyHostToDevice));
    return RR_OK;
}

// fp32 host rows -> bf16 device rows (rounded once, to nearest even, on the device)
static int ce_upload_bf16(unsigned short* dst, const float* src, size_t n) {
    float* tmp = nullptr;
    RR_HIP_TRY(hipMalloc((void**)&tmp, sizeof(float) * n));
    hipError_t e = hipMemcpy(tmp, src, sizeof(float) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ce_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, tmp, dst, (int64_t)n);
        e = hipDeviceSynchronize();
    }
    hipFree(tmp);
    if (e != hipSuccess) { rr_set_error("rr_ce_create: %s", hipGetErrorString(e)); return RR_E_HIP; }
    return RR_OK;
}

extern "C" int rr_ce_destroy(rr_ce* ce) {
    if (!ce) return RR_OK;
    hipSetDevice(ce->device);
    hipDeviceSynchronize();
    hipFree(ce->word); hipFree(ce->pos); hipFree(ce->type); hipFree(ce->eln_g); hipFree(ce->eln_b);
    if (ce->layers)
        for (int l = 0; l < ce->cfg.n_layers; ++l) {
            rr_ce_layer& L = ce->layers[l];
            hipFree(L.wqkv); hipFree(L.wo); hipFree(L.w1); hipFree(L.w2); hipFree(L.w2p);
            hipFree(L.bqkv); hipFree(L.bo); hipFree(L.b1); hipFree(L.b2);
            hipFree(L.ln1_g); hipFree(L.ln1_b); hipFree(L.ln2_g); hipFree(L.ln2_b);
            hipFree(L.wqkv32); hipFree(L.wo32); hipFree(L.w1_32); hipFree(L.w2_32);
            hipFree(L.wqkv_h2); hipFree(L.wo_h2); hipFree(L.w1_h2); hipFree(L.w2_h2);
        }
    delete[] ce->layers;
    hipFree(ce->wp); hipFree(ce->bp); hipFree(ce->wc); hipFree(ce->bc); hipFree(ce->gelu_tab);
    hipFree(ce->h32); hipFree(ce->hb); hipFree(ce->qkv); hipFree(ce->ctx); hipFree(ce->inter);
    hipFree(ce->h32c); hipFree(ce->hbc); hipFree(ce->ctxc); hipFree(ce->interc);
    hipFree(ce->qkv32); hipFree(ce->y32); hipFree(ce->inter32);
    hipFree(ce->hx); hipFree(ce->ctxh); hipFree(ce->d_flag);
    hipFree(ce->h32c32); hipFree(ce->y32c); hipFree(ce->hxc); hipFree(ce->ctxhc); hipFree(ce->interhc);
    if (ce->ev0) hipEventDestroy(ce->ev0);
    if (ce->ev1) hipEventDestroy(ce->ev1);
    delete ce;
    return RR_OK;
}

extern "C" int rr_ce_create(int32_t device, const rr_ce_config* cfg, const float* const* t, int32_t n_tensors,
                            rr_ce** out) {
    RR_REQUIRE(out, "rr_ce_create: NULL out");
    *out = nullptr;
    RR_REQUIRE(cfg && t, "rr_ce_create: NULL argument");
    static_assert(CE_H % 128 == 0 && CE_FFN % 128 == 0, "the GEMM loop takes K tiles of 64 in pairs");
    RR_REQUIRE(cfg->hidden == CE_H && cfg->n_heads == CE_HEADS && cfg->ffn == CE_FFN,
               "rr_ce_create: the kernels are built for hidden 384 / 12 heads x 32 / FFN 1536 (MiniLM-L6, bge-small); "
               "got hidden %d heads %d ffn %d", cfg->hidden, cfg->n_heads, cfg->ffn);
    RR_REQUIRE(cfg->n_layers >= 1 && cfg->n_layers <= 48 && cfg->vocab >= 1 && cfg->max_pos >= 1 && cfg->max_pos <= 512 &&
                   cfg->type_vocab >= 1 && cfg->n_labels >= 0 && cfg->n_labels <= 64 && cfg->ln_eps > 0.f,
               "rr_ce_create: bad configuration (layers %d vocab %d max_pos %d types %d labels %d)", cfg->n_layers,
               cfg->vocab, cfg->max_pos, cfg->type_vocab, cfg->n_labels);
    const int want = 5 + 16 * cfg->n_layers + (cfg->n_labels > 0 ? 4 : 0);
    RR_REQUIRE(n_tensors == want, "rr_ce_create: %d tensors given, %d expected (include/rr_hip.h lists the order)",
               n_tensors, want);
    RR_REQUIRE(cfg->precision == RR_CE_PRECISION_BF16 || cfg->precision == RR_CE_PRECISION_F32,
               "rr_ce_create: unknown precision %d", cfg->precision);
    for (int i = 0; i < n_tensors; ++i) RR_REQUIRE(t[i], "rr_ce_create: tensor %d is NULL", i);
    RR_HIP_TRY(hipSetDevice(device));
    rr_ce* ce = new rr_ce();
    ce->device = device;
    ce->cfg = *cfg;
    ce->layers = new rr_ce_layer[cfg->n_layers];
    const size_t H = CE_H, F = CE_FFN;
    int rc = RR_OK;
    auto f32 = [&](float** d, const float* s, size_t n) { if (!rc) rc = ce_upload_f32(d, s, n); };
    auto b16 = [&](unsigned short* d, const float* s, size_t n) { if (!rc) rc = ce_upload_bf16(d, s, n); };
    auto alloc16 = [&](unsigned short** d, size_t n) {
        if (!rc && hipMalloc((void**)d, 2 * n) != hipSuccess) { rr_set_error("rr_ce_create: out of device memory"); rc = RR_E_NOMEM; }
    };
    f32(&ce->word, t[0], (size_t)cfg->vocab * H);
    f32(&ce->pos, t[1], (size_t)cfg->max_pos * H);
    f32(&ce->type, t[2], (size_t)cfg->type_vocab * H);
    f32(&ce->eln_g, t[3], H);
    f32(&ce->eln_b, t[4], H);
    for (int l = 0; l < cfg->n_layers && !rc; ++l) {
        const float* const* p = t + 5 + 16 * l;      // q_w q_b k_w k_b v_w v_b o_w o_b ln1_g ln1_b f1_w f1_b f2_w f2_b ln2_g ln2_b
        rr_ce_layer& L = ce->layers[l];
        alloc16(&L.wqkv, 3 * H * H);
        if (!rc) { b16(L.wqkv, p[0], H * H); b16(L.wqkv + H * H, p[2], H * H); b16(L.wqkv + 2 * H * H, p[4], H * H); }
        if (!rc && hipMalloc((void**)&L.bqkv, sizeof(float) * 3 * H) != hipSuccess) rc = RR_E_NOMEM;
        if (!rc) {
            hipMemcpy(L.bqkv, p[1], sizeof(float) * H, hipMemcpyHostToDevice);
            hipMemcpy(L.bqkv + H, p[3], sizeof(float) * H, hipMemcpyHostToDevice);
            hipMemcpy(L.bqkv + 2 * H, p[5], sizeof(float) * H, hipMemcpyHostToDevice);
        }
        alloc16(&L.wo, H * H);   b16(L.wo, p[6], H * H);   f32(&L.bo, p[7], H);
        f32(&L.ln1_g, p[8], H);  f32(&L.ln1_b, p[9], H);
        alloc16(&L.w1, F * H);   b16(L.w1, p[10], F * H);  f32(&L.b1, p[11], F);
        alloc16(&L.w2, H * F);   b16(L.w2, p[12], H * F);  f32(&L.b2, p[13], H);
        alloc16(&L.w2p, H * F);
        if (!rc) {
            // position 8 h + jj of a 16-group holds column 8 (jj >> 2) + 4 h + (jj & 3): quads (0, 2, 1, 3)
            std::vector<float> perm(H * F);
            for (size_t n = 0; n < H; ++n)
                for (size_t j = 0; j < F; ++j) {
                    const size_t q = (j >> 2) & 3, src = (j & ~(size_t)15) + 4 * (q == 1 ? 2 : q == 2 ? 1 : q) + (j & 3);
                    perm[n * F + j] = p[12][n * F + src];
                }
            b16(L.w2p, perm.data(), H * F);
        }
        f32(&L.ln2_g, p[14], H); f32(&L.ln2_b, p[15], H);
        if (cfg->precision == RR_CE_PRECISION_F32 && !rc) {
            if (hipMalloc((void**)&L.wqkv32, sizeof(float) * 3 * H * H) != hipSuccess) rc = RR_E_NOMEM;
            if (!rc) {
                hipMemcpy(L.wqkv32, p[0], sizeof(float) * H * H, hipMemcpyHostToDevice);
                hipMemcpy(L.wqkv32 + H * H, p[2], sizeof(float) * H * H, hipMemcpyHostToDevice);
                hipMemcpy(L.wqkv32 + 2 * H * H, p[4], sizeof(float) * H * H, hipMemcpyHostToDevice);
            }
            f32(&L.wo32, p[6], H * H);
            f32(&L.w1_32, p[10], F * H);
            f32(&L.w2_32, p[12], H * F);
            // the same four matrices as h2 images (two fp16 planes, as large as the fp32 matrix): ce_gemm_h2's A operand
            auto h2 = [&](void** d, const float* src32, size_t n, size_t k) {
                if (rc) return;
                if (hipMalloc(d, n * k * 4) != hipSuccess) { rc = RR_E_NOMEM; return; }
                ce_h2_pack(src32, (int)n, (int)k, *d, (int64_t)n, nullptr);
            };
            h2(&L.wqkv_h2, L.wqkv32, 3 * H, H);
            h2(&L.wo_h2, L.wo32, H, H);
            h2(&L.w1_h2, L.w1_32, F, H);
            h2(&L.w2_h2, L.w2_32, H, F);
        }
    }
    if (!rc && cfg->n_labels > 0) {
        const float* const* p = t + 5 + 16 * cfg->n_layers;
        f32(&ce->wp, p[0], H * H); f32(&ce->bp, p[1], H);
        f32(&ce->wc, p[2], (size_t)cfg->n_labels * H); f32(&ce->bc, p[3], (size_t)cfg->n_labels);
    }
    if (!rc) {
        std::vector<float> tab(2 * CE_GELU_N);
        ce_gelu_table(tab.data());
        f32(&ce->gelu_tab, tab.data(), tab.size());
    }
    if (!rc && (hipEventCreate(&ce->ev0) != hipSuccess || hipEventCreate(&ce->ev1) != hipSuccess)) rc = RR_E_HIP;
    if (!rc && (hipMalloc((void**)&ce->d_flag, 4) != hipSuccess || hipMemset(ce->d_flag, 0, 4) != hipSuccess)) rc = RR_E_NOMEM;
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = RR_E_HIP;      // (the weight images are packed)
    if (rc) { rr_ce_destroy(ce); return rc; }
    *out = ce;
    return RR_OK;
}

static int ce_reserve(rr_ce* ce, int64_t tokens) {
    if (tokens <= ce->cap) return RR_OK;
    RR_HIP_TRY(hipDeviceSynchronize());
    hipFree(ce->h32); hipFree(ce->hb); hipFree(ce->qkv); hipFree(ce->ctx); hipFree(ce->inter);
    ce->h32 = nullptr; ce->hb = ce->qkv = ce->ctx = ce->inter = nullptr;
    ce->cap = 0;
    const size_t n = (size_t)rr_round_up(tokens, 4096);
    // the fp32 precision touches h32 / hb only (ce_embed_ln writes both); its own activations are ce_reserve_f32's: no
    // bf16 qkv / ctx / inter for it (5.4 KB per token, 0.7 GB at the 131 072-token call size)
    const bool bf16_path = ce->cfg.precision != RR_CE_PRECISION_F32;
    hipError_t e = hipMalloc((void**)&ce->h32, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&ce->hb, n * CE_H * 2);
    if (e == hipSuccess && bf16_path) e = hipMalloc((void**)&ce->qkv, n * 3 * CE_H * 2);
    if (e == hipSuccess && bf16_path) e = hipMalloc((void**)&ce->ctx, n * CE_H * 2);
    if (e == hipSuccess && bf16_path) e = hipMalloc((void**)&ce->inter, n * CE_FFN * 2);
    if (e != hipSuccess) { rr_set_error("rr_ce_forward: activation scratch for %lld tokens: %s", (long long)tokens, hipGetErrorString(e)); return RR_E_NOMEM; }
    ce->cap = (int64_t)n;
    return RR_OK;
}

// workgroups of the XCD-aware 1-D GEMM grid: row tiles rounded up to a multiple of 8 (one per XCD), times column tiles
static unsigned ce_grid(int M, int BM, int n_col_tiles) {
    const int mt = (M + BM - 1) / BM;
    return (unsigned)(((mt + 7) / 8) * 8 * n_col_tiles);
}

static int ce_reserve_seqs(rr_ce* ce, int64_t seqs) {
    if (seqs <= ce->cap_seqs) return RR_OK;
    RR_HIP_TRY(hipDeviceSynchronize());
    hipFree(ce->h32c); hipFree(ce->hbc); hipFree(ce->ctxc); hipFree(ce->interc);
    ce->h32c = nullptr; ce->hbc = ce->ctxc = ce->interc = nullptr;
    ce->cap_seqs = 0;
    const size_t n = (size_t)rr_round_up(seqs, 1024);
    hipError_t e = hipMalloc((void**)&ce->h32c, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&ce->hbc, n * CE_H * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&ce->ctxc, n * CE_H * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&ce->interc, n * CE_FFN * 2);
    if (e != hipSuccess) { rr_set_error("rr_ce_forward: compact scratch for %lld sequences: %s", (long long)seqs, hipGetErrorString(e)); return RR_E_NOMEM; }
    ce->cap_seqs = (int64_t)n;
    return RR_OK;
}

static size_t ce_attention_lds(int smax_pad) { return (size_t)smax_pad * CE_KS_LD * 2 + (size_t)32 * (smax_pad + 8) * 2; }

// per-device one-time opt-in to dynamic LDS above 64 KB (hipFuncSetAttribute applies to the current device)
static int ce_set_attributes(int device) {
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64 || done[device]) return RR_OK;
    const int ldsB = (128 + 384) * CE_LDK * 2, ldsP = 8 * 64 * (96 + 8) * 2;
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_gemm<128, 384, 2, 4, CE_EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsP));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_gemm<128, 384, 2, 4, CE_EPI_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsP));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_gemm<128, 384, 2, 4, CE_EPI_RES_LN>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsB));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_attention, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ce_attention_lds(512)));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_ffn_fused<4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_FFN_LDS));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_ffn_fused<4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_FFN_LDS));
#ifdef RR_DEBUG_HARNESS
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_ffn_fused<4, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_FFN_LDS));
#endif
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
#ifdef RR_DEBUG_HARNESS
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_proj_ts<4, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_QKV_LDS(3 * CE_H)));
#endif
#ifdef RR_DEBUG_HARNESS
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_ffn_fused<8, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_FFN_LDS));
#endif
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_attention_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * (CE_F32_KLD + CE_HD) * 4));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_attention_x3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_X3A_LDS));
    RR_HIP_TRY(hipFuncSetAttribute((const void*)ce_attention_x3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CE_X3A_LDS));
    if (int rc = ce_h2_set_attributes()) return rc;
    done[device] = true;
    return RR_OK;
}

static int ce_reserve_f32(rr_ce* ce, int64_t tokens) {
    if (tokens <= ce->cap32) return RR_OK;
    RR_HIP_TRY(hipDeviceSynchronize());
    hipFree(ce->qkv32); hipFree(ce->y32); hipFree(ce->inter32); hipFree(ce->hx); hipFree(ce->ctxh);
    ce->qkv32 = ce->y32 = ce->inter32 = nullptr;
    ce->hx = ce->ctxh = nullptr;
    ce->cap32 = 0;
    const size_t n = (size_t)rr_round_up(tokens, 4096);      // (= the row stride of the h2 images: a multiple of 256)
    hipError_t e = hipMalloc((void**)&ce->qkv32, n * 3 * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&ce->y32, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&ce->inter32, n * CE_FFN * 4);
    if (e == hipSuccess) e = hipMalloc(&ce->hx, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc(&ce->ctxh, n * CE_H * 4);
    // (rows past a call's last token are read by the last tile's LDS-DMA and never stored: defined bits, once)
    if (e == hipSuccess) e = hipMemset(ce->hx, 0, n * CE_H * 4);
    if (e == hipSuccess) e = hipMemset(ce->ctxh, 0, n * CE_H * 4);
    if (e == hipSuccess) e = hipMemset(ce->inter32, 0, n * CE_FFN * 4);
    if (e == hipSuccess) e = hipMemset(ce->qkv32, 0, n * 3 * CE_H * 4);
    if (e == hipSuccess) e = hipDeviceSynchronize();      // (the fills run on the NULL stream: a non-blocking caller stream would not wait for them)
    if (e != hipSuccess) { rr_set_error("rr_ce_forward: fp32 activation scratch for %lld tokens: %s", (long long)tokens, hipGetErrorString(e)); return RR_E_NOMEM; }
    ce->cap32 = (int64_t)n;
    return RR_OK;
}

// the forward pass of RR_CE_PRECISION_F32 (every operand fp32): embeddings -> per layer QKV, attention, output projection
// + residual + LayerNorm, FFN (exact GELU) + residual + LayerNorm -> head
static int ce_reserve_seqs_f32(rr_ce* ce, int64_t seqs) {
    if (seqs <= ce->cap32_seqs) return RR_OK;
    RR_HIP_TRY(hipDeviceSynchronize());
    hipFree(ce->h32c32); hipFree(ce->y32c); hipFree(ce->hxc); hipFree(ce->ctxhc); hipFree(ce->interhc);
    ce->h32c32 = ce->y32c = nullptr;
    ce->hxc = ce->ctxhc = ce->interhc = nullptr;
    ce->cap32_seqs = 0;
    const size_t n = (size_t)rr_round_up(seqs, 1024);
    hipError_t e = hipMalloc((void**)&ce->h32c32, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&ce->y32c, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc(&ce->hxc, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc(&ce->ctxhc, n * CE_H * 4);
    if (e == hipSuccess) e = hipMalloc(&ce->interhc, n * CE_FFN * 4);
    if (e == hipSuccess) e = hipMemset(ce->hxc, 0, n * CE_H * 4);
    if (e == hipSuccess) e = hipMemset(ce->ctxhc, 0, n * CE_H * 4);
    if (e == hipSuccess) e = hipMemset(ce->interhc, 0, n * CE_FFN * 4);
    if (e == hipSuccess) e = hipDeviceSynchronize();      // (the fills run on the NULL stream: a non-blocking caller stream would not wait for them)
    if (e != hipSuccess) { rr_set_error("rr_ce_forward: compact fp32 scratch for %lld sequences: %s", (long long)seqs, hipGetErrorString(e)); return RR_E_NOMEM; }
    ce->cap32_seqs = (int64_t)n;
    return RR_OK;
}

// ... the same forward on the fp16 matrix cores (rr_ce_h2.hip): every GEMM operand an h2 image written by its producer
static int ce_forward_h2(rr_ce* ce, const int32_t* d_token_ids, const int32_t* d_type_ids, const int32_t* d_pos_ids,
                         const int32_t* d_cu_seqlens, int n_seqs, int T, int max_len, int mode, float* d_out, hipStream_t st) {
    const int64_t xs = ce->cap32;
    if (mode != RR_CE_OUT_HIDDEN) {
        const int rc = ce_reserve_seqs_f32(ce, n_seqs);
        if (rc) return rc;
    }
    const int64_t xc = ce->cap32_seqs;
    ce_h2_embed_ln(d_token_ids, d_type_ids, d_pos_ids, T, ce->cfg.vocab, ce->cfg.max_pos, ce->cfg.type_vocab, ce->word, ce->pos, ce->type,
                   ce->eln_g, ce->eln_b, ce->cfg.ln_eps, ce->h32, ce->hx, xs, ce->d_flag, st);
    for (int l = 0; l < ce->cfg.n_layers; ++l) {
        const rr_ce_layer& L = ce->layers[l];
        // QKV as ONE h2 image of [T][1152] inside the qkv32 allocation (Q scaled by log2 e / sqrt 32);
        // RR_CE_H2_ATT_X3=1 (A/B): fp32 QKV + the bf16 three-term attention
        static const bool att_x3 = getenv("RR_CE_H2_ATT_X3") != nullptr;
        const bool cls_tail = l == ce->cfg.n_layers - 1 && mode != RR_CE_OUT_HIDDEN;
        if (att_x3) {
            ce_h2_gemm(CE_H2_EPI_F32, L.wqkv_h2, 3 * CE_H, ce->hx, xs, T, CE_H, L.bqkv, ce->qkv32, nullptr, 0, ce->d_flag, st);
        } else {
            ce_h2_gemm(CE_H2_EPI_H2, L.wqkv_h2, 3 * CE_H, ce->hx, xs, T, CE_H, L.bqkv, nullptr, ce->qkv32, xs, ce->d_flag, st,
                       0.17677669529663687f * 1.4426950408889634f, CE_H);
            ce_h2_attention(ce->qkv32, xs, d_cu_seqlens, n_seqs, max_len, cls_tail ? ce->ctxhc : ce->ctxh, cls_tail ? xc : xs, ce->d_flag,
                            cls_tail ? 1 : 0, st);
        }
        if (cls_tail) {
            // The last layer of a [CLS]-pooled output needs keys and values of every token but only the [CLS] query row (as
            // the bf16 path below): attention for that row alone, everything behind it on one compact row per sequence.
            if (att_x3)
                hipLaunchKernelGGL(ce_attention_x3<true>, dim3((unsigned)n_seqs, CE_HEADS), dim3(512), CE_X3A_LDS, st, ce->qkv32, d_cu_seqlens,
                                   (float*)nullptr, 0.17677669529663687f, (u32x2*)ce->ctxhc, xc, ce->d_flag, 1);
            hipLaunchKernelGGL(ce_gather_cls, dim3((unsigned)n_seqs), dim3(128), 0, st, ce->h32, d_cu_seqlens, ce->h32c32);
            ce_h2_gemm(CE_H2_EPI_F32, L.wo_h2, CE_H, ce->ctxhc, xc, n_seqs, CE_H, L.bo, ce->y32c, nullptr, 0, ce->d_flag, st);
            ce_h2_add_ln(ce->y32c, ce->h32c32, n_seqs, L.ln1_g, L.ln1_b, ce->cfg.ln_eps, ce->hxc, xc, ce->d_flag, st);
            ce_h2_gemm(CE_H2_EPI_GELU_H2, L.w1_h2, CE_FFN, ce->hxc, xc, n_seqs, CE_H, L.b1, nullptr, ce->interhc, xc, ce->d_flag, st);
            ce_h2_gemm(CE_H2_EPI_F32, L.w2_h2, CE_H, ce->interhc, xc, n_seqs, CE_FFN, L.b2, ce->y32c, nullptr, 0, ce->d_flag, st);
            ce_h2_add_ln(ce->y32c, ce->h32c32, n_seqs, L.ln2_g, L.ln2_b, ce->cfg.ln_eps, ce->hxc, xc, ce->d_flag, st);
            hipLaunchKernelGGL(ce_head, dim3((unsigned)n_seqs), dim3(256), 0, st, ce->h32c32, (const int32_t*)nullptr, ce->wp, ce->bp, ce->wc,
                               ce->bc, ce->cfg.n_labels, mode, d_out, (const unsigned*)ce->d_flag);
            return RR_OK;
        }
        if (att_x3)
            hipLaunchKernelGGL(ce_attention_x3<true>, dim3((unsigned)n_seqs, CE_HEADS), dim3(512), CE_X3A_LDS, st, ce->qkv32, d_cu_seqlens,
                               (float*)nullptr, 0.17677669529663687f /* 1 / sqrt(32) */, (u32x2*)ce->ctxh, xs, ce->d_flag, 0);
        ce_h2_gemm(CE_H2_EPI_F32, L.wo_h2, CE_H, ce->ctxh, xs, T, CE_H, L.bo, ce->y32, nullptr, 0, ce->d_flag, st);
        ce_h2_add_ln(ce->y32, ce->h32, T, L.ln1_g, L.ln1_b, ce->cfg.ln_eps, ce->hx, xs, ce->d_flag, st);
        ce_h2_gemm(CE_H2_EPI_GELU_H2, L.w1_h2, CE_FFN, ce->hx, xs, T, CE_H, L.b1, nullptr, ce->inter32, xs, ce->d_flag, st);
        ce_h2_gemm(CE_H2_EPI_F32, L.w2_h2, CE_H, ce->inter32, xs, T, CE_FFN, L.b2, ce->y32, nullptr, 0, ce->d_flag, st);
        ce_h2_add_ln(ce->y32, ce->h32, T, L.ln2_g, L.ln2_b, ce->cfg.ln_eps, ce->hx, xs, ce->d_flag, st);
    }
    // (mode != RR_CE_OUT_HIDDEN left through the last layer's [CLS] tail above)
    RR_HIP_TRY(hipMemcpyAsync(d_out, ce->h32, sizeof(float) * (size_t)T * CE_H, hipMemcpyDeviceToDevice, st));
    return RR_OK;
}

static int ce_forward_f32(rr_ce* ce, const int32_t* d_token_ids, const int32_t* d_type_ids, const int32_t* d_pos_ids,
                          const int32_t* d_cu_seqlens, int n_seqs, int T, int max_len, int mode, float* d_out, hipStream_t st) {
    int rc = ce_reserve_f32(ce, T);
    if (rc) return rc;
    // RR_CE_F32_SPLIT=bf16x3 (A/B) or rr_ce_set_wide_range: three bf16 terms per operand, six products -- any fp32 range
    RR_HIP_TRY(hipMemsetAsync(ce->d_flag, 0, 4, st));
    static const bool split_bf16 = getenv("RR_CE_F32_SPLIT") != nullptr && strcmp(getenv("RR_CE_F32_SPLIT"), "bf16x3") == 0;
    if (!split_bf16 && !ce->wide_range && getenv("RR_CE_F32_MFMA") == nullptr && getenv("RR_CE_F32_ATT_MFMA32") == nullptr)
        return ce_forward_h2(ce, d_token_ids, d_type_ids, d_pos_ids, d_cu_seqlens, n_seqs, T, max_len, mode, d_out, st);
    hipLaunchKernelGGL(ce_embed_ln, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, d_token_ids, d_type_ids, d_pos_ids, T,
                       ce->cfg.vocab, ce->cfg.max_pos, ce->cfg.type_vocab, ce->word, ce->pos, ce->type, ce->eln_g, ce->eln_b,
                       ce->cfg.ln_eps, ce->h32, ce->hb);
    const unsigned mt = (unsigned)((T + 127) / 128), ln_blocks = (unsigned)((T + 3) / 4);
    static const bool f32_x3 = getenv("RR_CE_F32_MFMA") == nullptr;      // (RR_CE_F32_MFMA=1, A/B: the four GEMMs on v_mfma_f32_32x32x2_f32 instead of by operand splitting)
    const size_t att_lds = (size_t)((max_len + 31) & ~31) * (CE_F32_KLD + CE_HD) * 4;
    for (int l = 0; l < ce->cfg.n_layers; ++l) {
        const rr_ce_layer& L = ce->layers[l];
        if (f32_x3) hipLaunchKernelGGL((ce_gemm_x3<false>), dim3(3 * CE_H / 128, mt), dim3(256), 0, st, ce->h32, L.wqkv32, L.bqkv, T, 3 * CE_H, CE_H, ce->qkv32);
        else hipLaunchKernelGGL((ce_gemm_f32<false>), dim3(3 * CE_H / 128, mt), dim3(256), 0, st, ce->h32, L.wqkv32, L.bqkv, T, 3 * CE_H, CE_H, ce->qkv32);
        static const bool att_mfma32 = getenv("RR_CE_F32_ATT_MFMA32") != nullptr;      // (A/B: the fp32-input matrix instruction)
        if (att_mfma32)
            hipLaunchKernelGGL(ce_attention_f32, dim3((unsigned)n_seqs, CE_HEADS), dim3(256), att_lds, st, ce->qkv32, d_cu_seqlens, ce->y32,
                               0.17677669529663687f /* 1 / sqrt(32) */);
        else
            hipLaunchKernelGGL(ce_attention_x3<false>, dim3((unsigned)n_seqs, CE_HEADS), dim3(512), CE_X3A_LDS, st, ce->qkv32, d_cu_seqlens, ce->y32,
                               0.17677669529663687f /* 1 / sqrt(32) */, (u32x2*)nullptr, (int64_t)0, (unsigned*)nullptr, 0);
        // (y32 holds the context; the projection's output goes to the first T x 384 floats of inter32)
        if (f32_x3) hipLaunchKernelGGL((ce_gemm_x3<false>), dim3(CE_H / 128, mt), dim3(256), 0, st, ce->y32, L.wo32, L.bo, T, CE_H, CE_H, ce->inter32);
        else hipLaunchKernelGGL((ce_gemm_f32<false>), dim3(CE_H / 128, mt), dim3(256), 0, st, ce->y32, L.wo32, L.bo, T, CE_H, CE_H, ce->inter32);
        hipLaunchKernelGGL(ce_add_ln_f32, dim3(ln_blocks), dim3(256), 0, st, ce->inter32, ce->h32, T, L.ln1_g, L.ln1_b, ce->cfg.ln_eps);
        if (f32_x3) {
            hipLaunchKernelGGL((ce_gemm_x3<true>), dim3(CE_FFN / 128, mt), dim3(256), 0, st, ce->h32, L.w1_32, L.b1, T, CE_FFN, CE_H, ce->inter32);
            hipLaunchKernelGGL((ce_gemm_x3<false>), dim3(CE_H / 128, mt), dim3(256), 0, st, ce->inter32, L.w2_32, L.b2, T, CE_H, CE_FFN, ce->y32);
        } else {
            hipLaunchKernelGGL((ce_gemm_f32<true>), dim3(CE_FFN / 128, mt), dim3(256), 0, st, ce->h32, L.w1_32, L.b1, T, CE_FFN, CE_H, ce->inter32);
            hipLaunchKernelGGL((ce_gemm_f32<false>), dim3(CE_H / 128, mt), dim3(256), 0, st, ce->inter32, L.w2_32, L.b2, T, CE_H, CE_FFN, ce->y32);
        }
        hipLaunchKernelGGL(ce_add_ln_f32, dim3(ln_blocks), dim3(256), 0, st, ce->y32, ce->h32, T, L.ln2_g, L.ln2_b, ce->cfg.ln_eps);
    }
    if (mode == RR_CE_OUT_HIDDEN)
        RR_HIP_TRY(hipMemcpyAsync(d_out, ce->h32, sizeof(float) * (size_t)T * CE_H, hipMemcpyDeviceToDevice, st));
    else
        hipLaunchKernelGGL(ce_head, dim3((unsigned)n_seqs), dim3(256), 0, st, ce->h32, d_cu_seqlens, ce->wp, ce->bp, ce->wc, ce->bc,
                           ce->cfg.n_labels, mode, d_out, (const unsigned*)nullptr);
    return RR_OK;
}

extern "C" int rr_ce_forward_dev(rr_ce* ce, const int32_t* d_token_ids, const int32_t* d_type_ids,
                                 const int32_t* d_pos_ids, const int32_t* d_cu_seqlens, int32_t n_seqs,
                                 int64_t n_tokens, int32_t max_len, int32_t mode, float* d_out, void* stream) {
    RR_REQUIRE(ce && d_token_ids && d_type_ids && d_pos_ids && d_cu_seqlens && d_out, "rr_ce_forward_dev: NULL argument");
    RR_REQUIRE(n_seqs >= 1 && n_tokens >= n_seqs && n_tokens < (1ll << 31), "rr_ce_forward_dev: %d sequences / %lld tokens",
               n_seqs, (long long)n_tokens);
    RR_REQUIRE(max_len >= 1 && max_len <= ce->cfg.max_pos, "rr_ce_forward_dev: max_len %d outside [1, %d]", max_len,
               ce->cfg.max_pos);
    RR_REQUIRE(mode == RR_CE_OUT_LOGITS || mode == RR_CE_OUT_CLS || mode == RR_CE_OUT_HIDDEN, "rr_ce_forward_dev: unknown mode %d", mode);
    RR_REQUIRE(mode != RR_CE_OUT_LOGITS || ce->cfg.n_labels > 0, "rr_ce_forward_dev: the model was created without a classifier head");
    std::lock_guard<std::mutex> lk(ce->mu);
    RR_HIP_TRY(hipSetDevice(ce->device));
    int rc = ce_reserve(ce, n_tokens);
    if (rc) return rc;
    if (ce->cfg.precision != RR_CE_PRECISION_F32) {      // (the CLS-tail buffers of the bf16 path's last layer)
        rc = ce_reserve_seqs(ce, n_seqs);
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    const int T = (int)n_tokens;
    rc = ce_set_attributes(ce->device);
    if (rc) return rc;
    if (ce->cfg.precision == RR_CE_PRECISION_F32) {
        hipEventRecord(ce->ev0, st);
        rc = ce_forward_f32(ce, d_token_ids, d_type_ids, d_pos_ids, d_cu_seqlens, n_seqs, T, max_len, mode, d_out, st);
        if (rc) return rc;
        hipEventRecord(ce->ev1, st);
        ce->timed = true;
        RR_HIP_TRY(hipGetLastError());
        return RR_OK;
    }
    const int smax_pad = (max_len + 31) & ~31;
    hipEventRecord(ce->ev0, st);
    hipLaunchKernelGGL(ce_embed_ln, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, d_token_ids, d_type_ids, d_pos_ids, T,
                       ce->cfg.vocab, ce->cfg.max_pos, ce->cfg.type_vocab, ce->word, ce->pos, ce->type, ce->eln_g, ce->eln_b,
                       ce->cfg.ln_eps, ce->h32, ce->hb);
    const size_t ldsB = (size_t)(128 + 384) * CE_LDK * 2;
    const size_t ldsP = 8 * 64 * (96 + 8) * 2;          // plain epilogues stage eight 64 x 96 sub-tiles (> the K-tile buffers)
    const size_t ldsF = (size_t)CE_FFN_LDS;     // fused FFN: two chunk buffers + biases / LayerNorm rows + the Phi table
    static const bool unfused = getenv("RR_CE_UNFUSED") != nullptr;   // A/B: FFN as two GEMM launches
    static const bool qkv_tiled = getenv("RR_CE_QKV_TILED") != nullptr;   // A/B: the QKV projection as the tiled GEMM
    static const bool oproj_apart = getenv("RR_CE_OPROJ_APART") != nullptr;   // A/B: attention output + LayerNorm as its own launch
    for (int l = 0; l < ce->cfg.n_layers; ++l) {
        const rr_ce_layer& L = ce->layers[l];
        // The last layer of a [CLS]-pooled output (logits, CLS embedding) needs keys and values of every token but
        // only the [CLS] query row: attention runs for that row alone and everything behind it -- output projection,
        // both LayerNorms, the FFN -- on one compact row per sequence instead of one per token.
        const bool cls_tail = (l == ce->cfg.n_layers - 1) && mode != RR_CE_OUT_HIDDEN;
        const int Mr = cls_tail ? n_seqs : T;                              // rows behind the attention
        float* r32 = cls_tail ? ce->h32c : ce->h32;
        unsigned short* rb = cls_tail ? ce->hbc : ce->hb;
        unsigned short* rctx = cls_tail ? ce->ctxc : ce->ctx;
        unsigned short* rint = cls_tail ? ce->interc : ce->inter;
        // (the transposed, token-persistent form of the fused FFN was also built for these two projections and measured
        //  slower than the tiled GEMM: QKV 262 vs 222 us, attention output 154 vs 124 us per layer at 131 072 tokens -- with
        //  nothing to fuse, one wave per SIMD loses to eight waves per tile)
        if (qkv_tiled)
            hipLaunchKernelGGL((ce_gemm<128, 384, 2, 4, CE_EPI_BIAS>), dim3(ce_grid(T, 128, 3 * CE_H / 384)), dim3(512), ldsP, st, ce->hb, L.wqkv,
                               L.bqkv, T, 3 * CE_H, CE_H, ce->qkv, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f);
        else
        {
#ifdef RR_DEBUG_HARNESS
            static const int pe = getenv("RR_CE_PROJ_EXP") ? atoi(getenv("RR_CE_PROJ_EXP")) : 0;
            const dim3 pg((unsigned)((T + CE_QKV_TOK - 1) / CE_QKV_TOK));
            if (pe == 1) hipLaunchKernelGGL((ce_proj_ts<4, 1>), pg, dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T, L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
            else if (pe == 2) hipLaunchKernelGGL((ce_proj_ts<4, 2>), pg, dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T, L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
            else if (pe == 3) hipLaunchKernelGGL((ce_proj_ts<4, 3>), pg, dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T, L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
            else if (pe == 7) hipLaunchKernelGGL((ce_proj_ts<4, 7>), pg, dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T, L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
            else if (pe == 8) hipLaunchKernelGGL((ce_proj_ts<4, 8>), pg, dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T, L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
            else if (pe == 16) hipLaunchKernelGGL((ce_proj_ts<4, 16>), pg, dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T, L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
            else if (pe == 32) hipLaunchKernelGGL((ce_proj_ts<4, 32>), pg, dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T, L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
            else
#endif
            hipLaunchKernelGGL((ce_proj_ts<4>), dim3((unsigned)((T + CE_QKV_TOK - 1) / CE_QKV_TOK)), dim3(512), CE_QKV_LDS(3 * CE_H), st, ce->hb, (int)T,
                               L.wqkv, L.bqkv, 3 * CE_H, ce->qkv);
        }
        hipLaunchKernelGGL(ce_attention, dim3((unsigned)n_seqs, CE_HEADS), dim3(CE_ATT_THREADS), ce_attention_lds(smax_pad), st,
                           ce->qkv, d_cu_seqlens, ce->ctx, 0.17677669529663687f /* 1 / sqrt(32) */, smax_pad,
                           cls_tail ? ce->ctxc : (unsigned short*)nullptr);
        if (cls_tail) hipLaunchKernelGGL(ce_gather_cls, dim3((unsigned)n_seqs), dim3(128), 0, st, ce->h32, d_cu_seqlens, ce->h32c);
        const bool oproj_fused = !unfused && !oproj_apart;
        if (!oproj_fused)
            hipLaunchKernelGGL((ce_gemm<128, 384, 2, 4, CE_EPI_RES_LN>), dim3(ce_grid(Mr, 128, 1)), dim3(512), ldsB, st, rctx, L.wo, L.bo, Mr, CE_H,
                               CE_H, rb, r32, L.ln1_g, L.ln1_b, ce->cfg.ln_eps);
        if (oproj_fused) {
            // attention output projection + residual + LayerNorm + FFN + residual + LayerNorm: one launch
            const dim3 fg((unsigned)((Mr + CE_FFN_TOK - 1) / CE_FFN_TOK));
#ifdef RR_DEBUG_HARNESS
            static const bool nostore = getenv("RR_CE_FFN_NOSTORE") != nullptr;
            if (nostore) hipLaunchKernelGGL((ce_ffn_fused<4, false, true, true>), fg, dim3(256), ldsF, st, rb, r32, Mr, L.w1, L.b1, L.w2p, L.b2, L.ln2_g,
                               L.ln2_b, ce->cfg.ln_eps, ce->gelu_tab, (const unsigned short*)rctx, (const unsigned short*)L.wo, (const float*)L.bo,
                               (const float*)L.ln1_g, (const float*)L.ln1_b);
            else
#endif
            hipLaunchKernelGGL((ce_ffn_fused<4, false, true>), fg, dim3(256), ldsF, st, rb, r32, Mr, L.w1, L.b1, L.w2p, L.b2, L.ln2_g,
                               L.ln2_b, ce->cfg.ln_eps, ce->gelu_tab, (const unsigned short*)rctx, (const unsigned short*)L.wo, (const float*)L.bo,
                               (const float*)L.ln1_g, (const float*)L.ln1_b);
        } else if (!unfused) {
            const dim3 fg((unsigned)((Mr + CE_FFN_TOK - 1) / CE_FFN_TOK));
#ifdef RR_DEBUG_HARNESS
            static const bool depth8 = getenv("RR_CE_FFN_DEPTH8") != nullptr;      // (tools/k5_stamps.py: A fragments 8 slots ahead)
            if (depth8) hipLaunchKernelGGL((ce_ffn_fused<8, false, false>), fg, dim3(256), ldsF, st, rb, r32, Mr, L.w1, L.b1, L.w2p, L.b2, L.ln2_g,
                                           L.ln2_b, ce->cfg.ln_eps, ce->gelu_tab, (const unsigned short*)nullptr, (const unsigned short*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr);
            else
#endif
            hipLaunchKernelGGL((ce_ffn_fused<4, false, false>), fg, dim3(256), ldsF, st, rb, r32, Mr, L.w1, L.b1, L.w2p, L.b2, L.ln2_g,
                               L.ln2_b, ce->cfg.ln_eps, ce->gelu_tab, (const unsigned short*)nullptr, (const unsigned short*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, (const float*)nullptr);
        } else {
            hipLaunchKernelGGL((ce_gemm<128, 384, 2, 4, CE_EPI_GELU>), dim3(ce_grid(Mr, 128, CE_FFN / 384)), dim3(512), ldsP, st, rb, L.w1, L.b1, Mr,
                               CE_FFN, CE_H, rint, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f);
            hipLaunchKernelGGL((ce_gemm<128, 384, 2, 4, CE_EPI_RES_LN>), dim3(ce_grid(Mr, 128, 1)), dim3(512), ldsB, st, rint, L.w2, L.b2, Mr, CE_H,
                               CE_FFN, rb, r32, L.ln2_g, L.ln2_b, ce->cfg.ln_eps);
        }
    }
    if (mode == RR_CE_OUT_HIDDEN)
        RR_HIP_TRY(hipMemcpyAsync(d_out, ce->h32, sizeof(float) * (size_t)T * CE_H, hipMemcpyDeviceToDevice, st));
    else
        hipLaunchKernelGGL(ce_head, dim3((unsigned)n_seqs), dim3(256), 0, st, ce->h32c, (const int32_t*)nullptr, ce->wp, ce->bp, ce->wc, ce->bc,
                           ce->cfg.n_labels, mode, d_out, (const unsigned*)nullptr);
    hipEventRecord(ce->ev1, st);
    ce->timed = true;
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

extern "C" int rr_ce_last_forward_ms(rr_ce* ce, float* out_ms) {
    RR_REQUIRE(ce && out_ms, "rr_ce_last_forward_ms: NULL argument");
    std::lock_guard<std::mutex> lk(ce->mu);
    RR_REQUIRE(ce->timed, "rr_ce_last_forward_ms: no forward pass has run yet");
    RR_HIP_TRY(hipSetDevice(ce->device));
    RR_HIP_TRY(hipEventSynchronize(ce->ev1));
    RR_HIP_TRY(hipEventElapsedTime(out_ms, ce->ev0, ce->ev1));
    return RR_OK;
}

extern "C" int rr_ce_range_status(rr_ce* ce, int32_t* out_of_range) {
    RR_REQUIRE(ce && out_of_range, "rr_ce_range_status: NULL argument");
    std::lock_guard<std::mutex> lk(ce->mu);
    *out_of_range = 0;
    if (!ce->timed) return RR_OK;
    RR_HIP_TRY(hipSetDevice(ce->device));
    RR_HIP_TRY(hipEventSynchronize(ce->ev1));
    unsigned f = 0;
    RR_HIP_TRY(hipMemcpy(&f, ce->d_flag, 4, hipMemcpyDeviceToHost));
    *out_of_range = f ? 1 : 0;
    return RR_OK;
}

extern "C" int rr_ce_set_wide_range(rr_ce* ce, int32_t on) {
    RR_REQUIRE(ce, "rr_ce_set_wide_range: NULL handle");
    std::lock_guard<std::mutex> lk(ce->mu);
    ce->wide_range = on != 0;
    return RR_OK;
}

#ifdef RR_DEBUG_HARNESS
// tools/k5_stamps.py: phase clocks of the LAST ce_ffn_fused launch (wave 0 of workgroups 0 and 600): per workgroup
// [prologue, iteration top, slots 0-23, slots 24-47, barrier, epilogue, -, -, -] shader cycles + wall_clock64 ticks (100 MHz)
extern "C" int rr_debug_ce_ffn_stamps(unsigned long long* out20) {
    RR_HIP_TRY(hipDeviceSynchronize());
    RR_HIP_TRY(hipMemcpyFromSymbol(out20, HIP_SYMBOL(ce_dbg_ffn), sizeof(unsigned long long) * 20));
    return RR_OK;
}
#endif
