// rr_dense_bf16.hip -- K1 over a bf16-stored matrix (BASELINE configs 4-5).
//
// The reference has no bf16 path.  The semantics here are the ones SURVEY 8d fixes: the matrix
// is rounded to bf16 ONCE (round-to-nearest-even, after the fp32 l2 normalisation), queries stay
// fp32, every product is an exact fp32 product of an exactly-representable bf16 row element and
// an fp32 query element, accumulation is fp32.  The oracle is the fp32 matvec over the rounded
// matrix upcast to fp32.  Storage halves the bytes per row (768 B at dim 384), so the HBM-bound
// scans run at twice the rows per second; the selection (rr_select) is unchanged.
//
// rr_scan_bf16<NB>      1..8 queries per matrix read, VALU.  Same shape as rr_scan_f32: a
//                       16-lane DPP row owns a matrix row, lane j loads 16 B (8 bf16) chunks
//                       j, j+16, j+32 (a wave instruction = four rows x 256 contiguous bytes),
//                       one fmaf chain over its 24 elements, rr_row16_sum over the row.
// rr_scan_mfma_bf16<NQT> 16/32/64 queries per read on the f32-input matrix cores: the bf16
//                       fragments are widened to fp32 in registers (a shift), so scores are
//                       still exact-f32 fmaf chains.  Same stream structure as
//                       rr_scan_mfma_f32: fragment loads straight to a VGPR ring (12 x 16 B per
//                       lane per 16-row M-tile, two M-tiles deep), queries in LDS, inline-asm
//                       loads with counted waits, M-tile-major score layout.
#include "rr_common.h"
#include "rr_dense.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rr_bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float rr_bf16_hi(unsigned int w) { return __uint_as_float(w & 0xFFFF0000u); }

// ------------------------------------------------------------------ fp32 -> bf16 rows
// Round to nearest even on the f32 bits; NaN stays a (quiet) NaN.
__device__ __forceinline__ unsigned int rr_f32_to_bf16_bits(float f) {
    const unsigned int u = __float_as_uint(f);
    if (f != f) return (u >> 16) | 0x0040u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// One wave per row: optional l2 normalisation in fp32 (utils.py:40-44), then rounding.
__global__ void rr_rows_to_bf16(const float* __restrict__ src, int dim, unsigned short* __restrict__ dst,
                                int dim_pad, int64_t n_rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* p = src + row * dim;
    float scale = 1.f;
    if (eps > 0.f) {
        float ss = 0.f;
        for (int i = lane; i < dim; i += 64) ss = __builtin_fmaf(p[i], p[i], ss);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
        scale = fmaxf(sqrtf(ss), eps);
    }
    unsigned short* d = dst + row * dim_pad;
    for (int i = lane; i < dim_pad; i += 64) {
        const float v = i < dim ? (eps > 0.f ? p[i] / scale : p[i]) : 0.f;
        d[i] = (unsigned short)rr_f32_to_bf16_bits(v);
    }
}

int rr_store_rows_bf16(rr_index* ix, int64_t first_row, int64_t n, float* d_rows_f32, float eps, hipStream_t st) {
    if (n == 0) return RR_OK;
    unsigned short* dst = reinterpret_cast<unsigned short*>(ix->d_matrix) + first_row * ix->dim_pad;
    hipLaunchKernelGGL(rr_rows_to_bf16, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, d_rows_f32, ix->dim, dst,
                       ix->dim_pad, n, eps);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

// ------------------------------------------------------------------ VALU scan, dim 384
template <int NB>
__global__ __launch_bounds__(RR_SCAN_THREADS, (NB <= 1 ? 4 : 2)) void rr_scan_bf16(
    const u32x4* __restrict__ mat, rr_scan_geom G, const float* __restrict__ queries,  // NB x 384
    float* __restrict__ sims, float* __restrict__ gmax, uint32_t* __restrict__ smax) {
    constexpr int ROWC = 48;                  // 16-byte chunks per row
    __shared__ f32x4 qs[NB][96];
    const int tid = threadIdx.x;
    for (int i = tid; i < NB * 96; i += RR_SCAN_THREADS) qs[i / 96][i % 96] = reinterpret_cast<const f32x4*>(queries)[i];
    __syncthreads();

    const int lane = tid & 63;
    const int sub = lane & 15;
    const int grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (RR_SCAN_THREADS / 64) + (tid >> 6);
    if (wave >= G.n_waves) return;
    const int64_t t0 = wave * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;

    // chunk c = sub + 16*i holds elements 8c .. 8c+7 = float4 2c and 2c+1 of the query
    f32x4 qreg[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        qreg[2 * i] = (NB == 1) ? qs[0][2 * (sub + 16 * i)] : f32x4{0.f, 0.f, 0.f, 0.f};
        qreg[2 * i + 1] = (NB == 1) ? qs[0][2 * (sub + 16 * i) + 1] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_rows = [&](u32x4 (&dst)[3], int64_t row) {
        row = row < G.n_rows ? row : G.n_rows - 1;
        const u32x4* p = mat + row * ROWC + sub;
#pragma unroll
        for (int i = 0; i < 3; ++i) dst[i] = __builtin_nontemporal_load(p + 16 * i);
    };
    float gm[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) gm[b] = -INFINITY;

    u32x4 bufA[3], bufB[3];
    load_rows(bufA, t0 * 64 + grp);
#pragma unroll 1
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t row0 = tile * 64;
        float mine[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) mine[b] = 0.f;
#pragma unroll(NB == 1 ? 8 : 1)
        for (int it = 0; it < 16; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                u32x4(&cur)[3] = half ? bufB : bufA;
                u32x4(&nxt)[3] = half ? bufA : bufB;
                load_rows(nxt, row0 + 4 * (it + half + 1) + grp);      // it + half == 15: next tile's first rows
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (NB > 1) asm volatile("" ::: "memory");          // keep one query slice live, re-read LDS
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const f32x4 q0 = (NB == 1) ? qreg[2 * i] : qs[b][2 * (sub + 16 * i)];
                        const f32x4 q1 = (NB == 1) ? qreg[2 * i + 1] : qs[b][2 * (sub + 16 * i) + 1];
                        const u32x4 x = cur[i];
                        acc = __builtin_fmaf(rr_bf16_lo(x.x), q0.x, acc);
                        acc = __builtin_fmaf(rr_bf16_hi(x.x), q0.y, acc);
                        acc = __builtin_fmaf(rr_bf16_lo(x.y), q0.z, acc);
                        acc = __builtin_fmaf(rr_bf16_hi(x.y), q0.w, acc);
                        acc = __builtin_fmaf(rr_bf16_lo(x.z), q1.x, acc);
                        acc = __builtin_fmaf(rr_bf16_hi(x.z), q1.y, acc);
                        acc = __builtin_fmaf(rr_bf16_lo(x.w), q1.z, acc);
                        acc = __builtin_fmaf(rr_bf16_hi(x.w), q1.w, acc);
                    }
                    acc = rr_row16_sum(acc);
                    mine[b] = (sub == it + half) ? acc : mine[b];
                }
            }
        }
        const int64_t my_row = row0 + 4 * sub + grp;
        const bool valid = my_row < G.n_rows;
        const bool group_end = tile == t1 - 1;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v = mine[b];
            v = (valid && v == v) ? v : -INFINITY;
            sims[(int64_t)b * G.n_pad + my_row] = v;
            const float m = rr_wave_max(v);
            gm[b] = fmaxf(gm[b], m);
            if (lane == 0) {
                gmax[(int64_t)b * G.n_tiles + tile] = m;
                if (group_end) smax[(int64_t)b * G.n_waves + wave] = rr_f2key(gm[b]);
            }
        }
    }
}

template <int NB>
static rr_scan_geom rr_launch_scan_bf16(rr_index* ix, const float* d_q, hipStream_t st) {
    static int waves = 0;
    if (!waves) waves = rr_resident_waves((const void*)rr_scan_bf16<NB>, RR_SCAN_THREADS, ix->device);
    rr_scan_geom G = rr_make_geom(ix, waves / 4);
    hipLaunchKernelGGL((rr_scan_bf16<NB>), dim3((G.n_waves + 3) / 4), dim3(RR_SCAN_THREADS), 0, st,
                       reinterpret_cast<const u32x4*>(ix->d_matrix), G, d_q, ix->d_sims, ix->d_gmax, ix->d_smax);
    return G;
}

int rr_dense_chunk_bf16(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                        float* d_scores, hipStream_t st) {
    const int slot = rr_scan_events_begin(ix, st);
    rr_scan_note(ix, 2, nq <= 1 ? 1 : nq <= 2 ? 2 : nq <= 4 ? 4 : 8, nq, 0);
    rr_scan_geom G;
    switch (nq) {
        case 1: G = rr_launch_scan_bf16<1>(ix, d_q, st); break;
        case 2: G = rr_launch_scan_bf16<2>(ix, d_q, st); break;
        case 3: case 4: G = rr_launch_scan_bf16<4>(ix, d_q, st); break;
        default: G = rr_launch_scan_bf16<8>(ix, d_q, st); break;
    }
    rr_scan_events_end(ix, slot, st);
    rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

// ------------------------------------------------------------------ matrix-core scan, dim 384
template <int NQT>
__global__ __launch_bounds__((NQT == 4 ? 512 : 256), 2) void rr_scan_mfma_bf16(
    const u32x4* __restrict__ mat, rr_scan_geom G, const float* __restrict__ queries,  // (16*NQT) x 384
    float* __restrict__ sims, float* __restrict__ gmax, uint32_t* __restrict__ smax) {
    constexpr int ROWF4 = 96;
    constexpr int ROWC = 48;
    constexpr int THREADS = NQT == 4 ? 512 : 256;
    __shared__ f32x4 qs[NQT * 16 * ROWF4];
    const int tid = threadIdx.x;
    for (int i = tid; i < NQT * 16 * ROWF4; i += THREADS) {
        const int q = i / ROWF4, f = i % ROWF4;
        qs[q * ROWF4 + ((f & ~15) | ((f ^ q) & 15))] = reinterpret_cast<const f32x4*>(queries)[i];
    }
    __syncthreads();

    const int lane = tid & 63;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (THREADS / 64) + (tid >> 6);
    if (wave >= G.n_waves) return;
    const int64_t t0 = wave * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;
    const int64_t m0 = t0 * 4, m1 = t1 * 4;          // 16-row M-tiles of this wave

    auto row_ptr = [&](int64_t mt) {
        mt = mt < m1 ? mt : m1 - 1;
        int64_t row = mt * 16 + r;
        row = row < G.n_rows ? row : G.n_rows - 1;
        return mat + row * ROWC + g;
    };
    // ring: chunk (g + 4j), j = 0..11, of two consecutive M-tiles (24 x 16 B per lane in flight)
#define RR_RING_LOAD16(dst, ptr, byteoff) \
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(byteoff) : "memory")
    // younger operations when chunk (M-tile, j) is needed: 23 ring loads + the score stores of the two
    // M-tiles in between (tile-maximum stores only make the wait conservative)
#define RR_RING_WAIT16(reg) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(23 + 2 * NQT) : "memory")
    u32x4 a[2][12];
    {
        const u32x4* p0 = row_ptr(m0);
        const u32x4* p1 = row_ptr(m0 + 1);
#pragma unroll
        for (int j = 0; j < 12; ++j) RR_RING_LOAD16(a[0][j], p0, 64 * j);
#pragma unroll
        for (int j = 0; j < 12; ++j) RR_RING_LOAD16(a[1][j], p1, 64 * j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float tile_max[NQT], gm[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) tile_max[t] = gm[t] = -INFINITY;

#pragma unroll 1
    for (int64_t mt = m0; mt < m1; mt += 2) {        // m1 - m0 is a multiple of 4
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int64_t cur = mt + half;
            const u32x4* pn = row_ptr(cur + 2);
            f32x4 acc[NQT][4];
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) acc[t][qd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int f = 2 * (g + 4 * j);           // first of the two query float4 this chunk meets
                const int qd = j / 3;
                RR_RING_WAIT16(a[half][j]);
                const u32x4 x = a[half][j];
                const float x0 = rr_bf16_lo(x.x), x1 = rr_bf16_hi(x.x), x2 = rr_bf16_lo(x.y), x3 = rr_bf16_hi(x.y);
                const float x4 = rr_bf16_lo(x.z), x5 = rr_bf16_hi(x.z), x6 = rr_bf16_lo(x.w), x7 = rr_bf16_hi(x.w);
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    const f32x4 b0 = qs[(16 * t + r) * ROWF4 + ((f & ~15) | ((f ^ r) & 15))];
                    const f32x4 b1 = qs[(16 * t + r) * ROWF4 + (((f + 1) & ~15) | (((f + 1) ^ r) & 15))];
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, b0.x, acc[t][qd], 0, 0, 0);
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, b0.y, acc[t][qd], 0, 0, 0);
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, b0.z, acc[t][qd], 0, 0, 0);
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x3, b0.w, acc[t][qd], 0, 0, 0);
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4, b1.x, acc[t][qd], 0, 0, 0);
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x5, b1.y, acc[t][qd], 0, 0, 0);
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x6, b1.z, acc[t][qd], 0, 0, 0);
                    acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x7, b1.w, acc[t][qd], 0, 0, 0);
                }
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2"
                             : "=v"(a[half][j]) : "v"(pn), "n"(64 * j), "v"(acc[NQT - 1][qd]) : "memory");
            }
            const int64_t row0 = cur * 16 + 4 * g;
            const bool tile_end = (cur & 3) == 3;
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                f32x4 v = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
                v.x = (row0 + 0 < G.n_rows && v.x == v.x) ? v.x : -INFINITY;
                v.y = (row0 + 1 < G.n_rows && v.y == v.y) ? v.y : -INFINITY;
                v.z = (row0 + 2 < G.n_rows && v.z == v.z) ? v.z : -INFINITY;
                v.w = (row0 + 3 < G.n_rows && v.w == v.w) ? v.w : -INFINITY;
                *reinterpret_cast<f32x4*>(sims + ((cur * (16 * NQT) + 16 * t + r) * 16 + 4 * g)) = v;
                float m4 = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
                m4 = fmaxf(m4, __shfl_xor(m4, 16, 64));
                m4 = fmaxf(m4, __shfl_xor(m4, 32, 64));
                tile_max[t] = fmaxf(tile_max[t], m4);
                if (tile_end) {
                    if (g == 0) gmax[(cur >> 2) * (16 * NQT) + 16 * t + r] = tile_max[t];
                    gm[t] = fmaxf(gm[t], tile_max[t]);
                    tile_max[t] = -INFINITY;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (redundant) loads
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NQT; ++t) smax[wave * (16 * NQT) + 16 * t + r] = rr_f2key(gm[t]);
    }
}

template <int NQT>
static int rr_dense_chunk_mfma_bf16_t(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                                      float* d_scores, hipStream_t st) {
    constexpr int THREADS = NQT == 4 ? 512 : 256;
    static int waves = 0;
    if (!waves) waves = rr_resident_waves((const void*)rr_scan_mfma_bf16<NQT>, THREADS, ix->device);
    rr_scan_geom G = rr_make_geom(ix, waves / 4);
    G.qs = 16 * NQT;
    const int slot = rr_scan_events_begin(ix, st);
    rr_scan_note(ix, 7, NQT, nq, 0);
    hipLaunchKernelGGL((rr_scan_mfma_bf16<NQT>), dim3((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), dim3(THREADS),
                       0, st, reinterpret_cast<const u32x4*>(ix->d_matrix), G, d_q, ix->d_sims, ix->d_gmax, ix->d_smax);
    rr_scan_events_end(ix, slot, st);
    rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

int rr_dense_chunk_mfma_bf16(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                             float* d_scores, hipStream_t st) {
    if (nq <= 16) return rr_dense_chunk_mfma_bf16_t<1>(ix, d_q, nq, pool, d_rows, d_scores, st);
    if (nq <= 32) return rr_dense_chunk_mfma_bf16_t<2>(ix, d_q, nq, pool, d_rows, d_scores, st);
    return rr_dense_chunk_mfma_bf16_t<4>(ix, d_q, nq, pool, d_rows, d_scores, st);
}
