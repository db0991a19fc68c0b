// rr_dense.hip -- K1: dense dot-product scan + exact top-pool selection (gfx950).
//
// Replaces cosine_similarity_search (utils.py:111-124), _cosine_pool
// (app/app_product_search.py:192-195) and cosine_search (app/test.py:125-132):
//   sims = M @ q ; argpartition(-sims, pool-1)[:pool] ; argsort.
//
// Data layout in HBM
//   M      n_rows x dim_pad fp32, row-major, dim_pad % 64 == 0 (zero padded)
//   sims   [query][n_pad]    fp32, n_pad = 64 * n_tiles          (4 B / row / query)
//   gmax   [query][n_tiles]  fp32, max of each 64-row tile        (1/16 B / row / query)
//   smax   [query][n_waves]  u32 ordered key of the max of each wave's run of tiles
//                            (a "group"; at most 8192 per query)
//
// rr_scan_f32: HBM-bound streaming kernel.  A 16-lane DPP row owns one matrix row:
// lane j reads float4 j, j+16, ... of the row (NF = dim_pad/64 loads, each wave
// instruction covers four rows x 256 contiguous bytes) and runs one fp32 fmaf
// chain over its 4*NF elements; the 16 partials are added by rr_row16_sum.
// Every row therefore has the same summation order wherever it sits, so a
// score does not depend on sharding, batch size or launch geometry.
// A wave owns a contiguous run of `tiles_per_wave` 64-row tiles; it walks a tile in 16
// steps of 4 rows, double-buffered in registers (2 x NF loads of 16 B per lane in
// flight, prefetching across the tile seam), keeps row (4*it+grp) in lane
// (it + 16*grp), and ends the tile with one 256-B store of the 64 scores plus
// their maximum.  No atomics, no inter-wave traffic.
//
// rr_select: one 1024-thread workgroup per query; see the comment on the kernel.
#include <stdlib.h>

#include "rr_common.h"
#include "rr_dense.h"
#include "rr_x3.h"

#define RR_SEL_THREADS 1024
#define RR_SEL_GCAP 4096    // slow path: tiles kept in LDS
#define RR_SEL_CCAP 12288   // candidate keys in LDS: the fast paths collect into the first half (6144) and sort in <= 8192
#define RR_SEL_SORTCAP 8192 // largest power of two inside it: the generic path collects up to this many
#define RR_SEL_RCAP 8192    // rr_select_rescored's key area for pools up to 512 (64 KB: two of its workgroups share a CU; rows are collected
                            // into the first half); larger pools take RR_SEL_CCAP
#define RR_SEL_LCAP 4096    // groups opened by the fast path

// ------------------------------------------------------------------ scan
template <int NF>
__device__ __forceinline__ void rr_load_rows(f32x4 (&dst)[NF], const f32x4* __restrict__ mat,
                                             int64_t row, int64_t n_rows, int sub) {
    row = row < n_rows ? row : n_rows - 1;  // tail rows re-read the last row; masked later
    const f32x4* p = mat + row * (int64_t)(NF * 16) + sub;
#pragma unroll
    for (int i = 0; i < NF; ++i) dst[i] = __builtin_nontemporal_load(p + 16 * i);
}

template <int NF, int NB, bool FROM_LDS>
__device__ __forceinline__ void rr_dot_rows(const f32x4 (&x)[NF], const f32x4 (&qreg)[NF],
                                            const f32x4 (*qs)[NF * 16], int sub, int it,
                                            float (&mine)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (FROM_LDS) asm volatile("" ::: "memory");  // keep one query slice live, re-read LDS
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const f32x4 q = FROM_LDS ? qs[b][16 * i + sub] : qreg[i];
            acc = __builtin_fmaf(x[i].x, q.x, acc);
            acc = __builtin_fmaf(x[i].y, q.y, acc);
            acc = __builtin_fmaf(x[i].z, q.z, acc);
            acc = __builtin_fmaf(x[i].w, q.w, acc);
        }
        acc = rr_row16_sum(acc);
        mine[b] = (sub == it) ? acc : mine[b];
    }
}

// End of a tile: lane (sub, grp) holds row row0 + 4*sub + grp for every query.
template <int NB>
__device__ __forceinline__ void rr_finish_tile(const rr_scan_geom& G, int64_t tile, int64_t wave,
                                               int64_t t0, int64_t t1, int lane, int sub, int grp,
                                               float (&mine)[NB], float (&gm)[NB],
                                               float* __restrict__ sims, float* __restrict__ gmax,
                                               uint32_t* __restrict__ smax) {
    const int64_t my_row = tile * 64 + 4 * sub + grp;
    const bool valid = my_row < G.n_rows;
    const bool group_end = tile == t1 - 1;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float v = mine[b];
        v = (valid && v == v) ? v : -INFINITY;  // NaN scores and pad rows rank last
        sims[(int64_t)b * G.n_pad + my_row] = v;
        const float m = rr_wave_max(v);
        gm[b] = fmaxf(gm[b], m);
        if (lane == 0) {
            gmax[(int64_t)b * G.n_tiles + tile] = m;
            if (group_end)
                smax[(int64_t)b * G.n_waves + wave] = rr_f2key(gm[b]);
        }
        if (group_end) gm[b] = -INFINITY;
    }
}

// The queries of a call whose flag is up, if there are at most NB of them: every wave works the list out for itself (four
// coalesced loads + ballots), so that no launch stands between the flags and their readers.  Returns the count -- 0 when
// MORE than NB flags are up (a clustered corpus flags every query: a pass of this scan for eight of them would only add to
// the split-operand passes the others need anyway); slots past the count repeat the last query.
template <int NB>
__device__ __forceinline__ int rr_flagged_list(const int32_t* __restrict__ flags, int nq, int lane, int (&list)[NB]) {
    int n = 0, total = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) list[b] = 0;
    for (int i = 0; i < (nq + 63) / 64; ++i) {
        const int q = 64 * i + lane;
        unsigned long long m = __ballot(q < nq && flags[q] != 0);
        total += __builtin_popcountll(m);
        while (m && n < NB) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (j == n) list[j] = 64 * i + b;
            ++n;
        }
    }
    if (total > NB) return 0;
#pragma unroll
    for (int j = 1; j < NB; ++j)
        if (j >= n && n > 0) list[j] = list[j - 1];
    return n;
}

// LISTED: the queries are those of `queries` whose flag is up, when there are at most NB of them (the filter path's flagged
// queries, served by this kernel's per-row chain: bit for bit the single-query answer); nothing (or too much) flagged ->
// every workgroup returns at once.
template <int NF, int NB, bool LISTED = false>
__global__ __launch_bounds__(RR_SCAN_THREADS, (NB <= 1 ? 4 : 2)) void rr_scan_f32(
    const f32x4* __restrict__ mat, rr_scan_geom G, const float* __restrict__ queries,  // NB x (NF*64)
    float* __restrict__ sims, float* __restrict__ gmax, uint32_t* __restrict__ smax,
    const int32_t* __restrict__ flags = nullptr, int nq_total = 0, int32_t* __restrict__ list_out = nullptr) {
    __shared__ f32x4 qs[NB][NF * 16];
    const int tid = threadIdx.x;
    if (LISTED) {
        int list[NB];
        const int n = rr_flagged_list<NB>(flags, nq_total, tid & 63, list);
        if (blockIdx.x == 0 && tid == 0) {
            list_out[0] = n;                          // (also when it is 0: the selection behind this launch reads it)
#pragma unroll
            for (int b = 0; b < NB; ++b) list_out[1 + b] = list[b];
        }
        if (n == 0) return;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            for (int i = tid; i < NF * 16; i += RR_SCAN_THREADS)
                qs[b][i] = reinterpret_cast<const f32x4*>(queries)[(int64_t)list[b] * (NF * 16) + i];
    } else {
        for (int i = tid; i < NB * NF * 16; i += RR_SCAN_THREADS)
            qs[i / (NF * 16)][i % (NF * 16)] = reinterpret_cast<const f32x4*>(queries)[i];
    }
    __syncthreads();

    const int lane = tid & 63;
    const int sub = lane & 15;
    const int grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (RR_SCAN_THREADS / 64) + (tid >> 6);
    if (wave >= G.n_waves) return;
    const int64_t t0 = wave * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;

    // NB == 1 keeps the query slice in registers; larger tiles re-read LDS
    // (conflict-free: the four rows of a wave broadcast the same 16 float4).
    f32x4 qreg[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) qreg[i] = (NB == 1) ? qs[0][16 * i + sub] : f32x4{0.f, 0.f, 0.f, 0.f};

    float gm[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) gm[b] = -INFINITY;

    f32x4 bufA[NF], bufB[NF];
    rr_load_rows<NF>(bufA, mat, t0 * 64 + grp, G.n_rows, sub);
#pragma unroll 1
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t row0 = tile * 64;
        float mine[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) mine[b] = 0.f;
#pragma unroll(NB == 1 ? 8 : 1)
        for (int it = 0; it < 16; it += 2) {
            rr_load_rows<NF>(bufB, mat, row0 + 4 * (it + 1) + grp, G.n_rows, sub);
            rr_dot_rows<NF, NB, (NB > 1)>(bufA, qreg, qs, sub, it, mine);
            // it == 14 prefetches the first rows of the next tile (clamped on the last one)
            rr_load_rows<NF>(bufA, mat, row0 + 4 * (it + 2) + grp, G.n_rows, sub);
            rr_dot_rows<NF, NB, (NB > 1)>(bufB, qreg, qs, sub, it + 1, mine);
        }
        rr_finish_tile<NB>(G, tile, wave, t0, t1, lane, sub, grp, mine, gm, sims, gmax, smax);
    }
}

// Generic-dimension variant (runtime NF); same per-row summation order.
template <int NB>
__global__ __launch_bounds__(RR_SCAN_THREADS) void rr_scan_f32_generic(
    const f32x4* __restrict__ mat, rr_scan_geom G, int nf, const float* __restrict__ queries,
    float* __restrict__ sims, float* __restrict__ gmax, uint32_t* __restrict__ smax) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = lane & 15;
    const int grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (RR_SCAN_THREADS / 64) + (tid >> 6);
    if (wave >= G.n_waves) return;
    const int64_t t0 = wave * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;
    const f32x4* q4 = reinterpret_cast<const f32x4*>(queries);
    float gm[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) gm[b] = -INFINITY;
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t row0 = tile * 64;
        float mine[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) mine[b] = 0.f;
        for (int it = 0; it < 16; ++it) {
            int64_t row = row0 + 4 * it + grp;
            row = row < G.n_rows ? row : G.n_rows - 1;
            const f32x4* p = mat + row * (int64_t)(nf * 16) + sub;
            float acc[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = 0.f;
            for (int i = 0; i < nf; ++i) {
                const f32x4 x = p[16 * i];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const f32x4 q = q4[(int64_t)b * nf * 16 + 16 * i + sub];
                    acc[b] = __builtin_fmaf(x.x, q.x, acc[b]);
                    acc[b] = __builtin_fmaf(x.y, q.y, acc[b]);
                    acc[b] = __builtin_fmaf(x.z, q.z, acc[b]);
                    acc[b] = __builtin_fmaf(x.w, q.w, acc[b]);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float sacc = rr_row16_sum(acc[b]);
                mine[b] = (sub == it) ? sacc : mine[b];
            }
        }
        rr_finish_tile<NB>(G, tile, wave, t0, t1, lane, sub, grp, mine, gm, sims, gmax, smax);
    }
}

// ------------------------------------------------------------------ batched scan (MFMA)
// rr_scan_mfma_f32<NQT>: the same scan for 16*NQT queries per launch (NQT = 1, 2, 4) on the
// f32-input matrix cores.  v_mfma_f32_16x16x4_f32 is exact f32 (a k-ordered fmaf chain) at the
// f32 vector rate, but one MFMA reuses each operand dword 16 times, so operand traffic per
// FLOP drops 16x and the kernel stays HBM-bound up to ~32 queries per pass
// (2*B flop/byte against a ~20 flop/byte f32 ridge), compute-bound beyond.
//
// Like rr_scan_f32 every wave is an independent stream over its own contiguous run of
// 64-row tiles: no barriers, no inter-wave traffic after the prologue.  Measured on this
// chip the scan rate follows the bytes in flight per CU (Little's law at ~5-8 us loaded
// latency), and only the register file can hold ~190 KB per CU: LDS-staged variants
// (register or LDS-DMA staging, 64-96 KB in flight) stopped at 4.6-5.2 TB/s.  So:
//   A operand  matrix rows go straight to VGPRs in fragment shape: lane (r = lane&15,
//              g = lane>>4) loads float4 (g + 4j), j = 0..23, of row r of a 16-row M-tile.
//              The 24 registers are a ring: as soon as the MFMAs of step j have consumed
//              a register, the load of the same j of the NEXT M-tile is issued into it, so a
//              wave keeps 24 KB in flight continuously (8 waves per CU: 192 KB).
//   B operand  the 16*NQT queries sit in LDS for the whole launch, XOR-swizzled
//              (slot = (f & ~15) | ((f ^ query) & 15)) so the fragment ds_read_b128 is
//              conflict-free; one read feeds four MFMAs.
// Lane (r, g) feeds MFMA (j, c) with component c of float4 (g + 4j) of row r / query r.
// The 384 dims are cut into four quarters (j in [6q, 6q+6)), each with its own accumulator;
// every score is (acc0 + acc1) + (acc2 + acc3), each acc a fixed fmaf chain, whatever NQT is
// and wherever the row sits: scores do not depend on batch size (within this kernel),
// sharding or grid.  (The order differs from rr_scan_f32's: the two kernels agree to f32
// rounding, ~1e-8, not bit for bit.)
template <int NQT>
__global__ __launch_bounds__((NQT == 4 ? 512 : 256), 2) void rr_scan_mfma_f32(
    const f32x4* __restrict__ mat, rr_scan_geom G, const float* __restrict__ queries,  // (16*NQT) x 384
    float* __restrict__ sims, float* __restrict__ gmax, uint32_t* __restrict__ smax) {
    constexpr int ROWF4 = 96;
    __shared__ f32x4 qs[NQT * 16 * ROWF4];
    constexpr int THREADS = NQT == 4 ? 512 : 256;   // 64 queries = 96 KB of LDS: one workgroup per CU, 8 waves
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
        int64_t row = mt * 16 + r;
        row = row < G.n_rows ? row : G.n_rows - 1;   // tail rows re-read the last row; masked later
        return mat + row * ROWF4 + g;
    };

    // The ring loads are issued from inline asm so that they stay interleaved with the MFMAs
    // (hipcc sinks plain loads to the end of the iteration and waits vmcnt(0) at its top);
    // the waits are therefore counted by hand.  RR_RING_WAIT(j) ties the wait to the register
    // it guards ("+v"), so the MFMAs that read it cannot be scheduled above it.
#define RR_RING_LOAD(dst, ptr, byteoff) \
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(byteoff) : "memory")
    // younger memory operations when the load of (M-tile, j) is needed: the 23 other ring loads
    // plus the NQT score stores of the previous M-tile (its occasional tile-maximum stores only
    // make the wait conservative)
#define RR_RING_WAIT(reg) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(reg) : "n"(23 + NQT) : "memory")
    f32x4 a[24];
    {
        const f32x4* p = row_ptr(m0);
#pragma unroll
        for (int j = 0; j < 24; ++j) RR_RING_LOAD(a[j], p, 64 * j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first M-tile: nothing older to count against
    }
    float tile_max[NQT], gm[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) tile_max[t] = gm[t] = -INFINITY;

#pragma unroll 1
    for (int64_t mt = m0; mt < m1; ++mt) {
        const f32x4* pn = row_ptr(mt + 1 < m1 ? mt + 1 : mt);
        f32x4 acc[NQT][4];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) acc[t][qd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            const int f = g + 4 * j;
            const int qd = j / 6;
            RR_RING_WAIT(a[j]);
            const f32x4 x = a[j];
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                const f32x4 b = qs[(16 * t + r) * ROWF4 + ((f & ~15) | ((f ^ r) & 15))];
                acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, b.x, acc[t][qd], 0, 0, 0);
                acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, b.y, acc[t][qd], 0, 0, 0);
                acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, b.z, acc[t][qd], 0, 0, 0);
                acc[t][qd] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, b.w, acc[t][qd], 0, 0, 0);
            }
            // the same j of the next M-tile goes into the register just consumed
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2"
                         : "=v"(a[j]) : "v"(pn), "n"(64 * j), "v"(acc[NQT - 1][qd]) : "memory");
        }
        // lane (r, g) holds rows 4*g .. 4*g+3 of the M-tile for query 16*t + r
        const int64_t row0 = mt * 16 + 4 * g;
        const bool tile_end = (mt & 3) == 3;
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            f32x4 v = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
            v.x = (row0 + 0 < G.n_rows && v.x == v.x) ? v.x : -INFINITY;   // NaN scores and pad rows rank last
            v.y = (row0 + 1 < G.n_rows && v.y == v.y) ? v.y : -INFINITY;
            v.z = (row0 + 2 < G.n_rows && v.z == v.z) ? v.z : -INFINITY;
            v.w = (row0 + 3 < G.n_rows && v.w == v.w) ? v.w : -INFINITY;
            // [M-tile][query slot][16 rows]: the wave's 16*NQT queries x 64 B form one contiguous block
            *reinterpret_cast<f32x4*>(sims + ((mt * (16 * NQT) + 16 * t + r) * 16 + 4 * g)) = v;
            float m4 = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            m4 = fmaxf(m4, __shfl_xor(m4, 16, 64));
            m4 = fmaxf(m4, __shfl_xor(m4, 32, 64));
            tile_max[t] = fmaxf(tile_max[t], m4);
            if (tile_end) {                              // fourth M-tile: the 64-row tile is complete
                if (g == 0) gmax[(mt >> 2) * (16 * NQT) + 16 * t + r] = tile_max[t];
                gm[t] = fmaxf(gm[t], tile_max[t]);
                tile_max[t] = -INFINITY;
            }
        }
    }
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < NQT; ++t) smax[wave * (16 * NQT) + 16 * t + r] = rr_f2key(gm[t]);
    }
}

// ------------------------------------------------------------------ select
// Suffix scan over a 256-bin histogram: finds the bin holding the k-th largest
// element (counting from the top bin) and the count strictly above it.
// Called by all threads; hist/scratch in LDS.  Returns via sel[0]=bin, sel[1]=above.
__device__ __forceinline__ void rr_pick_bin(const uint32_t* hist, uint32_t* wsum, uint32_t* sel,
                                            uint32_t k) {
    const int tid = threadIdx.x;
    uint32_t incl = 0, mine = 0;
    if (tid < 256) {
        // bin order: thread t looks at bin 255 - t, so an inclusive prefix over t
        // is the count of elements in bins >= 255 - t.
        mine = hist[255 - tid];
        incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d) incl += o;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
        incl += base;
        const uint32_t above = incl - mine;
        if (above < k && incl >= k) {
            sel[0] = 255 - tid;
            sel[1] = above;
        }
    }
    __syncthreads();
}

// Bitonic sort, descending, of n (power of two) 64-bit keys in LDS.
__device__ __forceinline__ void rr_bitonic_desc(uint64_t* keys, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (n >> 1); i += blockDim.x) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// Exact for any input, but slow on large N: 8-bit radix passes over every tile maximum with
// an LDS histogram (heavily contended when the keys share their top bits).  Kept as the
// fallback of rr_select for inputs whose candidate lists overflow the fast path's LDS.
// Leaves the ordered top-pool keys in cand[0..pool).
template <typename ScoreAt, typename TileMaxAt>
__device__ void rr_select_slow(ScoreAt score_at, TileMaxAt tile_max_at,
                               int64_t n_rows, int64_t n_tiles, int pool, uint32_t* hist,
                               uint32_t* wsum, uint32_t* sel, uint32_t* counters, uint32_t* glist,
                               uint64_t* cand) {
    const int tid = threadIdx.x;

    // ---- 1. tau = pool-th largest tile maximum (or "everything" if few tiles)
    uint32_t tau_key = 0;
    if (n_tiles > pool) {
        uint32_t prefix = 0, mask = 0, k = (uint32_t)pool;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int64_t i = tid; i < n_tiles; i += RR_SEL_THREADS) {
                const uint32_t key = rr_f2key(tile_max_at(i));
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            rr_pick_bin(hist, wsum, sel, k);
            prefix |= sel[0] << shift;
            mask |= 255u << shift;
            k -= sel[1];
            __syncthreads();
        }
        tau_key = prefix;
    }

    // ---- 2. tiles that can hold a top-pool row
    if (tid == 0) counters[0] = 0, counters[1] = 0;
    __syncthreads();
    bool all_tiles = (n_tiles <= pool);
    if (!all_tiles) {
        for (int64_t i = tid; i < n_tiles; i += RR_SEL_THREADS) {
            if (rr_f2key(tile_max_at(i)) >= tau_key) {
                const uint32_t slot = atomicAdd(&counters[0], 1u);
                if (slot < RR_SEL_GCAP) glist[slot] = (uint32_t)i;
            }
        }
        __syncthreads();
        if (counters[0] > RR_SEL_GCAP) all_tiles = true;  // massive ties: scan everything
    }
    const int64_t n_list = all_tiles ? n_tiles : (int64_t)counters[0];
    const int64_t n_slots = n_list * 64;

    // ---- 3. candidate rows: key >= tau inside the listed tiles
    for (int64_t i = tid; i < n_slots; i += RR_SEL_THREADS) {
        const int64_t t = all_tiles ? (i >> 6) : (int64_t)glist[i >> 6];
        const int64_t row = t * 64 + (i & 63);
        if (row < n_rows) {
            const uint32_t key = rr_f2key(score_at(row));
            if (key >= tau_key) {
                const uint32_t slot = atomicAdd(&counters[1], 1u);
                if (slot < RR_SEL_SORTCAP)
                    cand[slot] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)row);
            }
        }
    }
    __syncthreads();
    uint32_t n_cand = counters[1];

    if (n_cand > RR_SEL_SORTCAP) {
        // ---- 3b. too many survivors: exact radix select of the pool-th largest
        // 64-bit key over the same rows, then keep keys >= it (exactly pool of them).
        uint64_t prefix = 0, mask = 0;
        uint32_t k = (uint32_t)pool;
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int64_t i = tid; i < n_slots; i += RR_SEL_THREADS) {
                const int64_t t = all_tiles ? (i >> 6) : (int64_t)glist[i >> 6];
                const int64_t row = t * 64 + (i & 63);
                if (row < n_rows) {
                    const uint64_t key = ((uint64_t)rr_f2key(score_at(row)) << 32) |
                                         (uint64_t)(0xFFFFFFFFu - (uint32_t)row);
                    if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            rr_pick_bin(hist, wsum, sel, k);
            prefix |= (uint64_t)sel[0] << shift;
            mask |= (uint64_t)255u << shift;
            k -= sel[1];
            __syncthreads();
        }
        if (tid == 0) counters[1] = 0;
        __syncthreads();
        for (int64_t i = tid; i < n_slots; i += RR_SEL_THREADS) {
            const int64_t t = all_tiles ? (i >> 6) : (int64_t)glist[i >> 6];
            const int64_t row = t * 64 + (i & 63);
            if (row < n_rows) {
                const uint64_t key = ((uint64_t)rr_f2key(score_at(row)) << 32) |
                                     (uint64_t)(0xFFFFFFFFu - (uint32_t)row);
                if (key >= prefix) {
                    const uint32_t slot = atomicAdd(&counters[1], 1u);
                    if (slot < RR_SEL_SORTCAP) cand[slot] = key;
                }
            }
        }
        __syncthreads();
        n_cand = counters[1] < RR_SEL_SORTCAP ? counters[1] : RR_SEL_SORTCAP;  // == pool
    }

    // ---- 4. order the survivors (score desc, row asc) and emit the first pool
    if (n_cand > RR_SEL_SORTCAP) n_cand = RR_SEL_SORTCAP;   // cannot happen for consistent inputs; never overrun LDS
    int n_sort = 1;
    while (n_sort < (int)n_cand) n_sort <<= 1;
    for (int i = tid; i < n_sort; i += RR_SEL_THREADS)
        if (i >= (int)n_cand) cand[i] = 0;
    rr_bitonic_desc(cand, n_sort);
}

// k-th largest of n (<= 8 * 1024) keys, each thread holding up to 8 of them in registers.
// Bisection on the key bits, two bits per step (three trial thresholds), counting with wave
// ballots: no atomics and no shuffles, so runs of equal keys (the usual state of the high
// bits) cost nothing.  One barrier per step; the 3 x 16 per-wave counts are folded by every
// wave with one LDS read per lane and a 16-lane DPP sum.
// `max_steps` caps the bits resolved: the value returned is then a lower bound of the k-th
// largest (count(key >= result) >= k always holds), which is all the selection needs.
#define RR_SEL_RK 8
template <int RK>
__device__ uint32_t rr_kth_largest_reg(const uint32_t (&r)[RR_SEL_RK], int n_mine, uint32_t k,
                                       uint32_t (*cnt)[3][16], int& phase, int max_steps) {
    const int tid = threadIdx.x;
    const int ln = tid & 63;
    uint32_t mx = 0, mn = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < RK; ++j)
        if (j < n_mine) {
            mx = r[j] > mx ? r[j] : mx;
            mn = r[j] < mn ? r[j] : mn;
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t a = __shfl_xor(mx, m, 64), b = __shfl_xor(mn, m, 64);
        mx = a > mx ? a : mx;
        mn = b < mn ? b : mn;
    }
    if (ln == 0) {
        cnt[phase][0][tid >> 6] = mx;
        cnt[phase][1][tid >> 6] = mn;
    }
    __syncthreads();
    for (int w = 0; w < 16; ++w) {
        mx = cnt[phase][0][w] > mx ? cnt[phase][0][w] : mx;
        mn = cnt[phase][1][w] < mn ? cnt[phase][1][w] : mn;
    }
    phase ^= 1;
    if (mx == mn) return mx;
    int bit = 31 - __clz(mx ^ mn);                        // highest bit in which keys differ
    uint32_t prefix = (bit == 31) ? 0u : (mx >> (bit + 1)) << (bit + 1);
    for (int step = 0; bit >= 0 && step < max_steps; ++step) {
        const int lo = bit >= 1 ? bit - 1 : 0;            // this step decides bits [lo, bit]
        const bool two = bit >= 1;
        const uint32_t t1 = prefix | (1u << lo);                       // ..01
        const uint32_t t2 = prefix | (1u << bit);                      // ..10 (or ..1)
        const uint32_t t3 = prefix | (1u << bit) | (1u << lo);         // ..11
        uint32_t c1 = 0, c2 = 0, c3 = 0;                               // wave-uniform counts
#pragma unroll
        for (int j = 0; j < RK; ++j) {
            const bool have = j < n_mine;
            c1 += (uint32_t)__popcll(__ballot(have && r[j] >= t1));
            c2 += (uint32_t)__popcll(__ballot(have && r[j] >= t2));
            c3 += (uint32_t)__popcll(__ballot(have && r[j] >= t3));
        }
        if (ln == 0) {
            cnt[phase][0][tid >> 6] = c1;
            cnt[phase][1][tid >> 6] = c2;
            cnt[phase][2][tid >> 6] = c3;
        }
        __syncthreads();
        int v = ln < 48 ? (int)(&cnt[phase][0][0])[ln] : 0;
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
        const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane(v, 0);
        const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane(v, 16);
        const uint32_t s3 = (uint32_t)__builtin_amdgcn_readlane(v, 32);
        phase ^= 1;
        if (two) {
            if (s3 >= k) prefix = t3;
            else if (s2 >= k) prefix = t2;
            else if (s1 >= k) prefix = t1;
        } else if (s2 >= k) {
            prefix = t2;
        }
        bit = lo - 1;
    }
    return prefix;
}

// Orders n (<= 1024) distinct keys descending by counting, for each key, the keys above it.
__device__ void rr_rank_sort_desc(const uint64_t* keys, int n, uint64_t* out) {
    const int tid = threadIdx.x;
    uint64_t mine = 0;
    int rank = 0;
    if (tid < n) {
        mine = keys[tid];
        for (int j = 0; j < n; ++j) rank += keys[j] > mine ? 1 : 0;   // LDS broadcast reads
    }
    __syncthreads();
    if (tid < n) out[rank] = mine;
    __syncthreads();
}

// tau = the pool-th largest group maximum (1 = "everything" when there are no more than `pool`
// groups), and the groups that reach it into list2 (count in counters[0], which may exceed
// RR_SEL_LCAP: the caller checks).  All threads of the 1024-thread workgroup call this.
// `open_key` (in/out, optional): on entry nothing; when `eps2` >= 0 the groups are listed down to
// key(tau - eps2) instead of tau (approximate scores, rr_dense_flt.hip) and that key is returned in it.
template <typename GroupKeyAt>
__device__ uint32_t rr_sel_open_groups(GroupKeyAt group_key_at, int ng, int pool, uint32_t (*cnt)[3][16],
                                       int& phase, uint32_t* counters, uint32_t* list2,
                                       float eps2 = -1.f, uint32_t* open_key = nullptr, uint32_t thr_floor = 0u) {
    const int tid = threadIdx.x;
    uint32_t r[RR_SEL_RK];
    int n_mine = 0;
#pragma unroll
    for (int j = 0; j < RR_SEL_RK; ++j) {
        const int i = tid + j * RR_SEL_THREADS;
        r[j] = i < ng ? group_key_at(i) : 0u;
        n_mine += i < ng ? 1 : 0;
    }
    // 7 two-bit steps below the highest differing bit: tau is within 2^-14 of the spread of
    // the group maxima below the exact pool-th largest, i.e. it opens a handful more groups
    uint32_t tau = 1u;
    if (ng > pool) {
        if (ng <= 1 * RR_SEL_THREADS) tau = rr_kth_largest_reg<1>(r, n_mine, (uint32_t)pool, cnt, phase, 7);
        else if (ng <= 2 * RR_SEL_THREADS) tau = rr_kth_largest_reg<2>(r, n_mine, (uint32_t)pool, cnt, phase, 7);
        else if (ng <= 4 * RR_SEL_THREADS) tau = rr_kth_largest_reg<4>(r, n_mine, (uint32_t)pool, cnt, phase, 7);
        else tau = rr_kth_largest_reg<8>(r, n_mine, (uint32_t)pool, cnt, phase, 7);
    }
    if (tau == 0u) tau = 1u;                           // key 0 marks padding, never a score
    uint32_t thr = tau;
    if (eps2 >= 0.f) {
        thr = rr_f2key(rr_key2f(tau) - eps2);          // (-inf - x = -inf; a NaN bound gives key(NaN): see caller)
        if (thr == 0u) thr = 1u;
        if (thr_floor > thr) thr = thr_floor;          // row shards: the corpus-wide lower bound opens fewer groups
        if (open_key) *open_key = thr;
    }
#pragma unroll
    for (int j = 0; j < RR_SEL_RK; ++j) {
        if (j < n_mine && r[j] >= thr) {
            const uint32_t slot = atomicAdd(&counters[0], 1u);
            if (slot < RR_SEL_LCAP) list2[slot] = (uint32_t)(tid + j * RR_SEL_THREADS);
        }
    }
    __syncthreads();
    return tau;
}

// rr_select: one 1024-thread workgroup per query; exact top-pool by (score desc, row asc).
//
// The scan leaves three levels behind: the score of every row, the maximum of every 64-row
// tile, and the maximum of every wave's run of tiles (a "group"; at most 8192 of them).
// tau = the pool-th largest group maximum is a lower bound for the pool-th largest score:
// at least `pool` groups hold a row that reaches it.  So one bisection over the group
// maxima (in registers) fixes tau, and two filtering passes open only what can matter:
//   groups with max >= tau (~pool)  ->  their tiles with max >= tau (~pool)  ->
//   their rows with score >= tau (~pool + ties)  ->  ordered by the key (score, ~row).
// If a list outgrows LDS (pool > 256, massive ties, clustered rows) the generic radix path
// takes over; both are exact.
__global__ __launch_bounds__(RR_SEL_THREADS) void rr_select(
    rr_scan_geom G, const float* __restrict__ sims, const float* __restrict__ gmax,
    const uint32_t* __restrict__ smax, int pool, int64_t row_offset,
    int64_t* __restrict__ out_rows, float* __restrict__ out_scores, int32_t* __restrict__ dbg,
    const int32_t* __restrict__ only_if, int n_total, int64_t sims_slice, int64_t gmax_slice, int64_t smax_slice,
    const int32_t* __restrict__ qlist, int32_t* __restrict__ clear_flags) {
    if (qlist) {
        // scores of slot blockIdx.x belong to query qlist[1 + blockIdx.x] of the call (rr_scan_f32<.., LISTED>): its answer
        // goes there and its flag comes down (the split-operand fallback behind this launch skips it)
        if ((int)blockIdx.x >= qlist[0]) return;
        const int qo = qlist[1 + blockIdx.x] - (int)blockIdx.x;
        out_rows += (int64_t)qo * pool; out_scores += (int64_t)qo * pool; dbg += qo * 16;
        if (threadIdx.x == 0) clear_flags[qo + (int)blockIdx.x] = 0;
    }
    if (gridDim.y > 1) {
        // the sliced fallback (rr_dense_x3w_fallback_all): slice y = queries 64 y .. of the call, its own scratch
        const int y = blockIdx.y;
        if (64 * y + (int)blockIdx.x >= n_total) return;
        only_if += 64 * y;
        out_rows += (int64_t)64 * y * pool; out_scores += (int64_t)64 * y * pool; dbg += 64 * y * 16;
        sims += y * sims_slice; gmax += y * gmax_slice; smax += y * smax_slice;
    }
    if (only_if && !only_if[blockIdx.x]) return;      // fallback launch: this query was served already
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t cnt[2][3][16];
    __shared__ uint32_t sel[2];
    __shared__ uint32_t counters[4];
    __shared__ uint32_t list2[RR_SEL_LCAP];     // groups opened
    __shared__ uint32_t list1[RR_SEL_GCAP];     // tiles opened (aliased: the slow path's tile list)
    __shared__ uint64_t cand[RR_SEL_CCAP];

    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    // the two score layouts (rr_scan_geom::qs)
    const int QS = G.qs;
    auto score_at = [&](int64_t row) -> float {
        return QS ? sims[((row >> 4) * QS + q) * 16 + (row & 15)] : sims[(int64_t)q * G.n_pad + row];
    };
    auto tile_max_at = [&](int64_t t) -> float {
        return QS ? gmax[t * QS + q] : gmax[(int64_t)q * G.n_tiles + t];
    };
    auto group_key_at = [&](int i) -> uint32_t {
        return QS ? smax[(int64_t)i * QS + q] : smax[(int64_t)q * G.n_waves + i];
    };
    const int64_t n_rows = G.n_rows, n_tiles = G.n_tiles;
    int phase = 0;
    long long stamp[8];
    int n_stamp = 0;
#define RR_STAMP() do { if (n_stamp < 8) stamp[n_stamp++] = clock64(); } while (0)
    RR_STAMP();
    const int ng = G.n_waves;
    bool fast = ng <= RR_SEL_RK * RR_SEL_THREADS;

    if (tid < 4) counters[tid] = 0;
    __syncthreads();

    if (fast) {
        // ---- tau: pool-th largest group maximum; the groups that reach it
        const uint32_t tau = rr_sel_open_groups(group_key_at, ng, pool, cnt, phase, counters, list2);
        RR_STAMP();   // 1: tau found, groups listed
        if (counters[0] > RR_SEL_LCAP) fast = false;

        // ---- tiles of the opened groups whose maximum reaches tau
        if (fast) {
            const int C = (int)G.tiles_per_wave;
            const int64_t n2 = (int64_t)counters[0] * C;
            for (int64_t i = tid; i < n2; i += RR_SEL_THREADS) {
                const int64_t t = (int64_t)list2[i / C] * C + (i % C);
                if (t < n_tiles && rr_f2key(tile_max_at(t)) >= tau) {
                    const uint32_t slot = atomicAdd(&counters[1], 1u);
                    if (slot < RR_SEL_GCAP) list1[slot] = (uint32_t)t;
                }
            }
            __syncthreads();
            if (counters[1] > RR_SEL_GCAP) fast = false;
        }
        RR_STAMP();   // 2: tiles listed
        // ---- rows of the opened tiles whose score reaches tau
        if (fast) {
            const int n1 = (int)counters[1] * 64;
            for (int i = tid; i < n1; i += RR_SEL_THREADS) {
                const uint32_t row = list1[i >> 6] * 64u + (uint32_t)(i & 63);
                if ((int64_t)row < n_rows) {
                    const uint32_t key = rr_f2key(score_at(row));
                    if (key >= tau) {
                        const uint32_t slot = atomicAdd(&counters[2], 1u);
                        if (slot < RR_SEL_CCAP / 2)
                            cand[slot] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - row);
                    }
                }
            }
            __syncthreads();
            RR_STAMP();   // 3: candidate rows listed
            const uint32_t n_cand = counters[2];
            if (n_cand > RR_SEL_CCAP / 2 || n_cand < (uint32_t)pool) {
                fast = false;                              // massive ties at the cut
            } else if (n_cand <= RR_SEL_THREADS) {
                rr_rank_sort_desc(cand, (int)n_cand, cand + RR_SEL_CCAP / 2);
                for (int i = tid; i < pool; i += RR_SEL_THREADS) cand[i] = cand[RR_SEL_CCAP / 2 + i];
                __syncthreads();
            } else {
                int n_sort = 1;
                while (n_sort < (int)n_cand) n_sort <<= 1;
                for (int i = tid; i < n_sort; i += RR_SEL_THREADS)
                    if (i >= (int)n_cand) cand[i] = 0;
                rr_bitonic_desc(cand, n_sort);
            }
        }
    }
    RR_STAMP();   // 4: candidates ordered
    if (tid == 0) {   // per-query trace of the path taken (read by rr_index_select_trace)
        dbg[q * 16 + 0] = fast ? 1 : 0;
        dbg[q * 16 + 1] = (int32_t)counters[0];
        dbg[q * 16 + 2] = (int32_t)counters[1];
        dbg[q * 16 + 3] = (int32_t)counters[2];
        for (int i = 1; i < 8; ++i) dbg[q * 16 + 3 + i] = i < n_stamp ? (int32_t)(stamp[i] - stamp[i - 1]) : -1;
    }
    if (!fast) {
        __syncthreads();
        rr_select_slow(score_at, tile_max_at, n_rows, n_tiles, pool, hist, wsum, sel, counters, list1, cand);
    }
    for (int i = tid; i < pool; i += RR_SEL_THREADS) {
        const uint64_t key = cand[i];
        const uint32_t row = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
        out_rows[(int64_t)q * pool + i] = (int64_t)row + row_offset;
        out_scores[(int64_t)q * pool + i] = rr_key2f((uint32_t)(key >> 32));
    }
}

// ------------------------------------------------------------------ two-pass selection (split-operand scan)
// The split-operand scan (rr_dense_x3.hip) keeps no per-row scores: only the maximum of every
// 16-row M-tile ([tile][query][4]) and of every wave's run.  Selection therefore runs in two steps
// around rr_rescore_x3:
//   rr_select_mtiles   tau from the group maxima, then the M-tiles whose maximum reaches tau
//                      (~pool + ties) -> x3 scratch (ids, count, tau)
//   rr_rescore_x3      recomputes the 16 scores of each listed M-tile, bit for bit as the scan did
//   rr_select_rescored rows with score >= tau among them, ordered by (score desc, row asc)
// A query whose lists outgrow the scratch raises its flag in `fb`; the caller then re-runs the
// storing scan + rr_select for the flagged queries (both return at once when no flag is up).
__global__ __launch_bounds__(RR_SEL_THREADS) void rr_select_mtiles(
    rr_scan_geom G, const float* __restrict__ mmax, const uint32_t* __restrict__ smax, int pool,
    uint32_t* __restrict__ out_mtiles, int32_t* __restrict__ out_count, uint32_t* __restrict__ out_tau,
    int32_t* __restrict__ fb, int32_t* __restrict__ dbg, const float* __restrict__ eps,
    const float* __restrict__ sigma, int nq_a, int nq_b, int64_t mmax_set_stride, int64_t smax_set_stride,
    const float* __restrict__ floor) {      // [nq_a + nq_b] or null: a lower bound of the corpus-wide pool-th best SCORE
    __shared__ uint32_t cnt[2][3][16];
    __shared__ uint32_t counters[4];
    __shared__ uint32_t list2[RR_SEL_LCAP];
    const int tid = threadIdx.x;
    // Query Q of the launch belongs to scan launch (set) 0 or 1; inside its set it is slot q.  Per-query OUTPUT arrays
    // (mtiles, count, tau, fb, dbg) are indexed by Q; the set's maxima by slot q; eps / sigma by set * RR_FLT_MAXQ + q.
    const int Q = blockIdx.x;
    const int set = (nq_b > 0 && Q >= nq_a) ? 1 : 0;
    const int q = set ? Q - nq_a : Q;
    const int nq_set = set ? nq_b : nq_a;
    mmax += set * mmax_set_stride;
    smax += set * smax_set_stride;
    if (eps) eps += set * RR_FLT_MAXQ;
    if (sigma) sigma += set * RR_FLT_MAXQ;
    out_mtiles += (int64_t)(Q - q) * RR_X3_MCAP;
    out_count += Q - q; out_tau += Q - q; fb += Q - q; dbg += (Q - q) * 16;
    const int QS = G.qs;
    auto group_key_at = [&](int i) -> uint32_t { return smax[(int64_t)i * QS + q]; };
    const int gpw = G.gpw > 1 ? G.gpw : 1;
    const int ng = G.n_waves * gpw;
    int phase = 0;
    const long long c0 = clock64();
    long long c1 = c0, c2 = c0;
    if (tid < 4) counters[tid] = 0;
    __syncthreads();
    bool ok = ng <= RR_SEL_RK * RR_SEL_THREADS && ng > pool;   // few groups = a small matrix: take the stored path
    uint32_t tau = 1u;          // rows are kept down to this key ...
    uint32_t open = 1u;         // ... M-tiles (and groups) opened down to this one
    if (ok) {
        // approximate scan scores (|s~ - s| <= e): >= pool rows reach tau~ - e, every one of the true
        // top-pool has s~ >= tau~ - 2e.  2.05 / 1.02: slack for the roundings of this arithmetic itself.
        const float e = eps ? eps[q] : -1.f;
        if (eps && !(e >= 0.f && e < 3.0e38f)) ok = false;     // no finite bound for this query
        if (ok) {
            // (row shards, DESIGN.md section 5: a row of the corpus-wide top-pool has score >= floor, hence filter score
            //  >= floor - e: nothing below floor - 1.05 e has to be opened, whatever this shard's own threshold says)
            const float fl = (floor && eps) ? floor[Q] - 1.05f * e : -INFINITY;
            if (fl > -3.0e38f) {
                // A finite corpus-wide floor F (a lower bound of the corpus-wide pool-th best SCORE): every row that can be in
                // the merged answer has score >= F, filter score >= F - e.  No threshold search of this shard's own: groups and
                // M-tiles are opened down to F - 1.05 e and the row cut is F itself (rows below it may come back as -inf:
                // they only fill this shard's list up, the merge never takes them).
                open = rr_f2key(fl);
                if (open == 0u) open = 1u;
                for (int i = tid; i < ng; i += RR_SEL_THREADS) {
                    if (group_key_at(i) >= open) {
                        const uint32_t slot = atomicAdd(&counters[0], 1u);
                        if (slot < RR_SEL_LCAP) list2[slot] = (uint32_t)i;
                    }
                }
                __syncthreads();
                tau = rr_f2key(floor[Q]);
                if (counters[0] > RR_SEL_LCAP) {
                    // A hot shard (skewed / clustered corpus): the floor is the MINIMUM over the shards, far below this shard's
                    // own pool-th best row, and more groups reach it than the list holds.  The rows of the corpus-wide top-pool
                    // that live here are among this shard's OWN top-pool, so its own threshold is a valid (higher) cut as
                    // well: search it, like a shard without a floor does, and keep whichever of the two is higher.
                    __syncthreads();
                    if (tid == 0) counters[0] = 0;
                    __syncthreads();
                    const uint32_t floor_open = open, floor_tau = tau;
                    tau = rr_sel_open_groups(group_key_at, ng, pool, cnt, phase, counters, list2, 2.05f * e, &open, floor_open);
                    tau = rr_f2key(rr_key2f(tau) - 1.02f * e);
                    if (floor_tau > tau) tau = floor_tau;
                }
            } else {
                tau = rr_sel_open_groups(group_key_at, ng, pool, cnt, phase, counters, list2, eps ? 2.05f * e : -1.f, &open, 0u);
                if (!eps) open = tau;
                else tau = rr_f2key(rr_key2f(tau) - 1.02f * e);
            }
            if (tau == 0u) tau = 1u;
            if (counters[0] > RR_SEL_LCAP) ok = false;
        }
    }
    c1 = clock64();
    if (ok) {
        const int C = (int)G.tiles_per_group;
        const int64_t n2 = (int64_t)counters[0] * C;
        const f32x4* mm4 = reinterpret_cast<const f32x4*>(mmax);
        if (G.mm_pairs == 3) {
            // 8-row M-tiles (filter scan): per 32-row tile and query one word = bf16 tile maximum (rounded up) + four
            // 4-bit gap codes; upper bound of M-tile g = max - steps(code_g) * step.  Four 64-row tiles per thread and pass, all
            // loads issued before the first is looked at (the loop is otherwise one HBM round trip per pass).
            constexpr int U = 4;
            const float step = rr_flt_gap_step(eps, nq_set);     // the resolution the set's scan launch encoded its gaps with
            const float openf = open <= 0x007FFFFFu ? -INFINITY : rr_key2f(open);   // (keys below key(-inf) are not scores)
            // store prefilter of the scan: tiles whose maximum stayed below sigma[q] may not have been stored.  They
            // cannot hold a candidate iff sigma[q] <= open; otherwise this query takes the exact fallback.  (Words of
            // skipped tiles are stale: they can only open extra M-tiles, which the rescoring then discards.)
            if (sigma && !(sigma[q] <= openf)) ok = false;
            const int n2i = (int)n2;                                       // (<= 4096 groups x tiles per group)
            for (int i0 = tid; i0 < n2i; i0 += U * RR_SEL_THREADS) {
                int64_t tt[U];
                uint32_t w0[U], w1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * RR_SEL_THREADS;
                    tt[u] = -1;
                    if (i < n2i) {
                        const int g = (int)list2[i / C], k = g % gpw, j = i % C;
                        const int64_t in_run = (int64_t)k * C + j;             // tile of the wave's run
                        const int64_t t = (int64_t)(g / gpw) * G.tiles_per_wave + in_run;
                        if (in_run < G.tiles_per_wave && t < G.n_tiles) tt[u] = t;
                    }
                    const int64_t ts = tt[u] < 0 ? 0 : tt[u];
                    w0[u] = reinterpret_cast<const uint32_t*>(mmax)[(2 * ts) * QS + q];
                    w1[u] = reinterpret_cast<const uint32_t*>(mmax)[(2 * ts + 1) * QS + q];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (tt[u] < 0) continue;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const uint32_t w = half ? w1[u] : w0[u];
                        const float mx = __uint_as_float(w << 16);
                        if (!(mx >= openf)) continue;                          // the whole 32-row tile is below the threshold
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const uint32_t code = (w >> (16 + 4 * g)) & 15u;       // rr_flt_gap_code: 0..7 steps, then 8, 10 .. 22
                            const float bound = mx - rr_flt_gap_steps(code) * step;
                            if (bound >= openf) {
                                const uint32_t slot = atomicAdd(&counters[1], 1u);
                                if (slot < RR_X3_MCAP) out_mtiles[(int64_t)q * RR_X3_MCAP + slot] = (uint32_t)(tt[u] * 8 + 4 * half + g);
                            }
                        }
                    }
                }
            }
        } else
        for (int64_t i = tid; i < n2; i += RR_SEL_THREADS) {
            const int g = (int)list2[i / C], k = g % gpw, j = (int)(i % C);
            const int64_t in_run = (int64_t)k * C + j;                     // tile of the wave's run
            const int64_t t = (int64_t)(g / gpw) * G.tiles_per_wave + in_run;
            if (in_run >= G.tiles_per_wave || t >= G.n_tiles) continue;
            float v[4];
            if (G.mm_pairs) {
                const float2 m0 = reinterpret_cast<const float2*>(mmax)[(2 * t) * QS + q];
                const float2 m1 = reinterpret_cast<const float2*>(mmax)[(2 * t + 1) * QS + q];
                v[0] = m0.x, v[1] = m0.y, v[2] = m1.x, v[3] = m1.y;
            } else {
                const f32x4 m = mm4[t * QS + q];
                v[0] = m.x, v[1] = m.y, v[2] = m.z, v[3] = m.w;
            }
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                if (rr_f2key(v[sub]) >= open) {
                    const uint32_t slot = atomicAdd(&counters[1], 1u);
                    if (slot < RR_X3_MCAP) out_mtiles[(int64_t)q * RR_X3_MCAP + slot] = (uint32_t)(t * 4 + sub);
                }
            }
        }
        __syncthreads();
        if (counters[1] > RR_X3_MCAP) ok = false;
    }
    c2 = clock64();
    if (tid == 0) {
        out_count[q] = ok ? (int32_t)counters[1] : 0;
        out_tau[q] = tau;
        fb[q] = ok ? 0 : 1;
        dbg[q * 16 + 0] = ok ? 2 : 0;                 // 2 = two-pass (may still fall back in the second step)
        dbg[q * 16 + 1] = (int32_t)counters[0];
        dbg[q * 16 + 2] = (int32_t)counters[1];
        dbg[q * 16 + 3] = 0;
        dbg[q * 16 + 4] = (int32_t)(c1 - c0);         // shader cycles: threshold + group list | M-tile list
        dbg[q * 16 + 5] = (int32_t)(c2 - c1);
        for (int i = 6; i < 16; ++i) dbg[q * 16 + i] = -1;
    }
}

template <int RCAP>
__global__ __launch_bounds__(RR_SEL_THREADS) void rr_select_rescored(
    rr_scan_geom G, const uint32_t* __restrict__ mtiles, const int32_t* __restrict__ count,
    const uint32_t* __restrict__ tau_of, const float* __restrict__ sc, int pool, int64_t row_offset,
    int64_t* __restrict__ out_rows, float* __restrict__ out_scores, int32_t* __restrict__ fb,
    int32_t* __restrict__ dbg, int floor_mode) {
    __shared__ uint32_t counters[2];
    __shared__ uint64_t cand[RCAP];
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    if (fb[q]) return;
    if (tid < 2) counters[tid] = 0;
    __syncthreads();
    const uint32_t tau = tau_of[q];
    const int sh = G.mm_pairs == 3 ? 3 : 4;           // rows per listed M-tile: 8 or 16
    const int n1 = count[q] << sh;
    // (row shards under a corpus-wide floor: the best `pool` of ALL rescored rows is the answer whether or not `pool` of them
    //  reach the cut -- one pass when they fit)
    const bool all_rows = floor_mode && (uint32_t)n1 <= RCAP / 2;
    for (int i = tid; i < n1 && !all_rows; i += RR_SEL_THREADS) {
        const int64_t at = (int64_t)q * RR_X3_MCAP + (i >> sh);
        const int r = i & ((1 << sh) - 1);
        const uint32_t row = (mtiles[at] << sh) + (uint32_t)r;
        if ((int64_t)row < G.n_rows) {
            const uint32_t key = rr_f2key(sc[(at << sh) + r]);
            if (key >= tau) {
                const uint32_t slot = atomicAdd(&counters[0], 1u);
                if (slot < RCAP / 2) cand[slot] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - row);
            }
        }
    }
    __syncthreads();
    uint32_t n_cand = counters[0];
    if (all_rows || (floor_mode && n_cand < (uint32_t)pool && (uint32_t)n1 <= RCAP / 2)) {
        // Row shards with a corpus-wide floor: fewer M-tiles were opened than this shard's own top-pool needs, so fewer
        // than `pool` rows may reach its own cut.  Every row of the corpus-wide top-pool that lives here IS among the
        // rescored rows; the list is filled up with the best of the other rescored rows (exact scores, below the
        // corpus-wide cut: the merge never takes them).
        __syncthreads();
        if (tid == 0) counters[0] = 0;
        __syncthreads();
        for (int i = tid; i < n1; i += RR_SEL_THREADS) {
            const int64_t at = (int64_t)q * RR_X3_MCAP + (i >> sh);
            const int r = i & ((1 << sh) - 1);
            const uint32_t row = (mtiles[at] << sh) + (uint32_t)r;
            if ((int64_t)row < G.n_rows) {
                const uint32_t key = rr_f2key(sc[(at << sh) + r]);
                const uint32_t slot = atomicAdd(&counters[0], 1u);
                if (slot < RCAP / 2) cand[slot] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - row);
            }
        }
        __syncthreads();
        n_cand = counters[0];
    }
    if (tid == 0) dbg[q * 16 + 3] = (int32_t)n_cand;
    if (n_cand > RCAP / 2 || n_cand < (uint32_t)pool) {    // massive ties at the cut
        if (tid == 0) {
            fb[q] = 1;
            dbg[q * 16 + 0] = 0;
        }
        return;
    }
    if (n_cand <= RR_SEL_THREADS) {
        rr_rank_sort_desc(cand, (int)n_cand, cand + RCAP / 2);
        for (int i = tid; i < pool; i += RR_SEL_THREADS) cand[i] = cand[RCAP / 2 + i];
        __syncthreads();
    } else {
        int n_sort = 1;
        while (n_sort < (int)n_cand) n_sort <<= 1;
        for (int i = tid; i < n_sort; i += RR_SEL_THREADS)
            if (i >= (int)n_cand) cand[i] = 0;
        rr_bitonic_desc(cand, n_sort);
    }
    for (int i = tid; i < pool; i += RR_SEL_THREADS) {
        const uint64_t key = cand[i];
        const uint32_t row = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
        out_rows[(int64_t)q * pool + i] = (int64_t)row + row_offset;
        out_scores[(int64_t)q * pool + i] = rr_key2f((uint32_t)(key >> 32));
    }
}

rr_x3_scratch rr_x3_scratch_of(const rr_index* ix) {
    rr_x3_scratch s;
    char* p = static_cast<char*>(ix->d_x3);
    s.mtiles = reinterpret_cast<uint32_t*>(p);  p += sizeof(uint32_t) * RR_SEL_MAXQ * RR_X3_MCAP;
    s.count = reinterpret_cast<int32_t*>(p);    p += sizeof(int32_t) * RR_SEL_MAXQ;
    s.tau = reinterpret_cast<uint32_t*>(p);     p += sizeof(uint32_t) * RR_SEL_MAXQ;
    s.fb = reinterpret_cast<int32_t*>(p);       p += sizeof(int32_t) * RR_SEL_MAXQ;
    p += sizeof(float) * RR_SEL_MAXQ;            // (eps used to live here: it is per scan slot now, ix->d_eps)
    s.eps = ix->d_eps;
    s.sc = reinterpret_cast<float*>(p);
    return s;
}
size_t rr_x3_scratch_bytes() {
    return sizeof(uint32_t) * RR_SEL_MAXQ * RR_X3_MCAP + 4 * sizeof(int32_t) * RR_SEL_MAXQ +
           sizeof(float) * (size_t)RR_SEL_MAXQ * RR_X3_MCAP * 16;
}
void rr_launch_select_mtiles(rr_index* ix, const rr_scan_geom& G, int nq, int pool, hipStream_t st, const float* eps,
                             const float* sigma, int nq_b, int64_t mmax_set_stride, int64_t smax_set_stride,
                             const float* floor) {
    const rr_x3_scratch s = rr_x3_scratch_of(ix);
    hipLaunchKernelGGL(rr_select_mtiles, dim3(nq + nq_b), dim3(RR_SEL_THREADS), 0, st, G, ix->d_gmax, ix->d_smax, pool,
                       s.mtiles, s.count, s.tau, s.fb, ix->d_sel_trace, eps, sigma, nq, nq_b, mmax_set_stride,
                       smax_set_stride, floor);
}
void rr_launch_select_rescored(rr_index* ix, const rr_scan_geom& G, int nq, int pool, int64_t* d_rows,
                               float* d_scores, hipStream_t st, bool floor_mode) {
    const rr_x3_scratch s = rr_x3_scratch_of(ix);
    // (pools up to 512 -- every BASELINE configuration -- fit the 64 KB key area: two workgroups per CU)
    if (pool <= 512)
        hipLaunchKernelGGL((rr_select_rescored<RR_SEL_RCAP>), dim3(nq), dim3(RR_SEL_THREADS), 0, st, G, s.mtiles, s.count, s.tau,
                           s.sc, pool, ix->row_offset, d_rows, d_scores, s.fb, ix->d_sel_trace, floor_mode ? 1 : 0);
    else
        hipLaunchKernelGGL((rr_select_rescored<RR_SEL_CCAP>), dim3(nq), dim3(RR_SEL_THREADS), 0, st, G, s.mtiles, s.count, s.tau,
                           s.sc, pool, ix->row_offset, d_rows, d_scores, s.fb, ix->d_sel_trace, floor_mode ? 1 : 0);
}

// bound[Q] = (a lower bound of) the kth largest group maximum of the filter scores of query Q, minus 1.01 eps: at least
// kth rows of this matrix have a SCORE >= bound[Q].  Row shards exchange the minimum of these (kth = ceil(pool / shards):
// the union of the shards' kth best groups holds >= pool rows), which is then a lower bound of the corpus-wide pool-th
// best score -- rr_select_mtiles' `floor`.  -inf where there are no more than kth groups or no finite eps.
__global__ __launch_bounds__(RR_SEL_THREADS) void rr_group_kth(rr_scan_geom G, const uint32_t* __restrict__ smax, int kth,
                                                               const float* __restrict__ eps, float* __restrict__ bound,
                                                               int nq_a, int nq_b, int64_t smax_set_stride) {
    __shared__ uint32_t cnt[2][3][16];
    const int tid = threadIdx.x;
    const int Q = blockIdx.x;
    const int set = (nq_b > 0 && Q >= nq_a) ? 1 : 0;
    const int q = set ? Q - nq_a : Q;
    smax += set * smax_set_stride;
    eps += set * RR_FLT_MAXQ;
    const int QS = G.qs;
    const int gpw = G.gpw > 1 ? G.gpw : 1;
    const int ng = G.n_waves * gpw;
    int phase = 0;
    float out = -INFINITY;
    if (ng > kth && ng <= RR_SEL_RK * RR_SEL_THREADS) {      // (uniform)
        uint32_t r[RR_SEL_RK];
        int n_mine = 0;
#pragma unroll
        for (int j = 0; j < RR_SEL_RK; ++j) {
            const int i = tid + j * RR_SEL_THREADS;
            r[j] = i < ng ? smax[(int64_t)i * QS + q] : 0u;
            n_mine += i < ng ? 1 : 0;
        }
        uint32_t tau;
        if (ng <= 1 * RR_SEL_THREADS) tau = rr_kth_largest_reg<1>(r, n_mine, (uint32_t)kth, cnt, phase, 7);
        else if (ng <= 2 * RR_SEL_THREADS) tau = rr_kth_largest_reg<2>(r, n_mine, (uint32_t)kth, cnt, phase, 7);
        else if (ng <= 4 * RR_SEL_THREADS) tau = rr_kth_largest_reg<4>(r, n_mine, (uint32_t)kth, cnt, phase, 7);
        else tau = rr_kth_largest_reg<8>(r, n_mine, (uint32_t)kth, cnt, phase, 7);
        const float e = eps[q];
        if (tau > 0x007FFFFFu && e >= 0.f && e < 3.0e38f) out = rr_key2f(tau) - 1.01f * e;
        if (!(out == out)) out = -INFINITY;
    }
    if (tid == 0) bound[Q] = out;
}
void rr_launch_group_kth(rr_index* ix, const rr_scan_geom& G, int nq_a, int nq_b, int kth, const float* eps, float* d_bound,
                         int64_t smax_set_stride, hipStream_t st) {
    hipLaunchKernelGGL(rr_group_kth, dim3(nq_a + nq_b), dim3(RR_SEL_THREADS), 0, st, G, ix->d_smax, kth, eps, d_bound, nq_a,
                       nq_b, smax_set_stride);
}

// ------------------------------------------------------------------ l2 normalize
// l2_normalize (utils.py:40-44): x / max(||x||, eps), one 64-lane wave per row.
// (numpy's norm is sqrt of a pairwise float32 sum of squares; this sum order differs
// in the last bit, which is why parity tests normalise once and share the result.)
__global__ void rr_l2norm_f32(float* __restrict__ mat, int64_t n_rows, int dim_pad, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    float* p = mat + row * dim_pad;
    float ss = 0.f;
    for (int i = lane; i < dim_pad; i += 64) ss = __builtin_fmaf(p[i], p[i], ss);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    const float nrm = fmaxf(sqrtf(ss), eps);
    for (int i = lane; i < dim_pad; i += 64) p[i] = p[i] / nrm;
}

// ------------------------------------------------------------------ host side
// Tile / group maxima of the ACTIVE scan slot, for launches of up to `nq` queries.
static int rr_ensure_maxima(rr_index* ix, int nq) {
    const size_t nm = nq >= RR_MFMA_MAXQ ? RR_FLT_MAXQ : nq;
    if (ix->maxima_q >= (int)nm) return RR_OK;
    const int64_t n_tiles = rr_round_up(ix->n_rows, 64) / 64;
    if (ix->d_gmax) hipFree(ix->d_gmax);
    if (ix->d_smax) hipFree(ix->d_smax);
    ix->d_gmax = nullptr;
    ix->d_smax = nullptr;
    ix->maxima_q = 0;
    const size_t groups_cap = (size_t)RR_MAX_SCAN_WAVES;   // one group maximum per scan wave
    // tile / group maxima: up to RR_FLT_MAXQ queries per launch (x4: per-M-tile maxima of the matrix-core scans)
    // (the filter scan uses 2 words per 64-row tile and query: two sets of them fit, + one line per scan wave behind them)
    const size_t gmax_words = nm * n_tiles * 4 + (size_t)RR_MAX_SCAN_WAVES * RR_FLT_MAXQ;
    RR_HIP_TRY(hipMalloc(&ix->d_gmax, sizeof(float) * gmax_words));
    // rr_scan_flt's store prefilter leaves the words of skipped tiles as they were: start from "-inf, no gaps" so that
    // a never-written word cannot open anything (stale words of earlier launches can only add rescoring work)
    RR_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ix->d_gmax, 0x0000FF80, gmax_words, nullptr));
    RR_HIP_TRY(hipStreamSynchronize(nullptr));
    RR_HIP_TRY(hipMalloc(&ix->d_smax, sizeof(uint32_t) * 2 * nm * groups_cap));
    ix->maxima_q = (int32_t)nm;
    return RR_OK;
}

// Score scratch (shared by both slots: only selections and fallbacks touch it) + the active slot's maxima.
static int rr_ensure_scratch(rr_index* ix, int nq) {
    const int rc = rr_ensure_maxima(ix, nq);
    if (rc) return rc;
    if (ix->scratch_q >= nq || (ix->scratch_small && ix->scratch_q >= RR_MFMA_MAXQ)) return RR_OK;
    const int64_t n_tiles = rr_round_up(ix->n_rows, 64) / 64;
    if (ix->d_sims) hipFree(ix->d_sims);
    ix->d_sims = nullptr;
    ix->scratch_q = 0;
    // (more than 64 score slots only serve the sliced fallback of a filter call: when there is no room for them -- a very
    //  large index -- 64 slots do, and the fallback goes block by block: rr_dense_x3w_fallback_all)
    static const bool force_small = getenv("RR_SCRATCH_SMALL") != nullptr;   // (tests: the block-by-block fallback on a small index)
    if ((force_small && nq > RR_MFMA_MAXQ) ||
        hipMalloc(&ix->d_sims, sizeof(float) * (size_t)nq * n_tiles * 64) != hipSuccess) {
        (void)hipGetLastError();
        ix->d_sims = nullptr;
        if (nq <= RR_MFMA_MAXQ) { rr_set_error("rr_dense_topk: out of device memory for the score scratch"); return RR_E_NOMEM; }
        nq = RR_MFMA_MAXQ;
        RR_HIP_TRY(hipMalloc(&ix->d_sims, sizeof(float) * (size_t)nq * n_tiles * 64));
        ix->scratch_small = true;
    }
    ix->scratch_q = nq;
    return RR_OK;
}

// ---- scan slots (rr_common.h: rr_scan_slot)
static void rr_slot_store(const rr_index* ix, rr_scan_slot& s) {
    s.d_q = ix->d_q; s.d_qplanes = ix->d_qplanes; s.d_eps = ix->d_eps; s.d_gmax = ix->d_gmax; s.d_smax = ix->d_smax;
    s.maxima_q = ix->maxima_q; s.d_flt_samp = ix->d_flt_samp; s.d_flt_sigma = ix->d_flt_sigma; s.d_flt_prog = ix->d_flt_prog;
    s.flt_seq = ix->flt_seq; s.flt_prep_fresh = ix->flt_prep_fresh; s.flt_pending = ix->flt_pending;
}
static void rr_slot_fetch(rr_index* ix, const rr_scan_slot& s) {
    ix->d_q = s.d_q; ix->d_qplanes = s.d_qplanes; ix->d_eps = s.d_eps; ix->d_gmax = s.d_gmax; ix->d_smax = s.d_smax;
    ix->maxima_q = s.maxima_q; ix->d_flt_samp = s.d_flt_samp; ix->d_flt_sigma = s.d_flt_sigma; ix->d_flt_prog = s.d_flt_prog;
    ix->flt_seq = s.flt_seq; ix->flt_prep_fresh = s.flt_prep_fresh; ix->flt_pending = s.flt_pending;
}
// (rr_index_destroy) every slot's buffers into ix->parked[], none left in the index fields
void rr_slot_park(rr_index* ix) {
    rr_slot_store(ix, ix->parked[ix->cur_slot]);
    rr_slot_fetch(ix, rr_scan_slot());
}
// Makes `slot` the one the index fields describe (caller holds ix->mu).  Its small buffers are allocated at first use.
int rr_slot_activate(rr_index* ix, int slot) {
    RR_REQUIRE(slot >= 0 && slot < RR_SCAN_SLOTS, "scan slot %d outside [0, %d)", slot, RR_SCAN_SLOTS);
    if (slot != ix->cur_slot) {
        rr_slot_store(ix, ix->parked[ix->cur_slot]);
        rr_slot_fetch(ix, ix->parked[slot]);
        ix->parked[slot] = rr_scan_slot();
        ix->cur_slot = slot;
    }
    if (!ix->d_q) {
        RR_HIP_TRY(hipSetDevice(ix->device));
        RR_HIP_TRY(hipMalloc(&ix->d_qplanes, (size_t)2 * 3 * 64 * 384 * 2));
        RR_HIP_TRY(hipMalloc((void**)&ix->d_eps, sizeof(float) * RR_SEL_MAXQ));
        RR_HIP_TRY(hipMalloc((void**)&ix->d_q, sizeof(float) * (size_t)RR_MAX_BATCH * ix->dim_pad));
    }
    return RR_OK;
}

// Resident workgroups for a kernel: CUs x blocks per CU the register budget admits.
template <typename K>
static int rr_resident_grid(K kernel, int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RR_SCAN_THREADS, 0) != hipSuccess ||
        per_cu < 1)
        per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1)
        cus = 256;
    int grid = per_cu * cus;
    if (grid * (RR_SCAN_THREADS / 64) > RR_MAX_SCAN_WAVES) grid = RR_MAX_SCAN_WAVES / (RR_SCAN_THREADS / 64);
    return grid;
}

// Splits the tiles into equal contiguous runs, one per wave of (at most) a resident grid.
rr_scan_geom rr_make_geom(const rr_index* ix, int resident_blocks) {
    rr_scan_geom G;
    G.n_rows = ix->n_rows;
    G.n_tiles = rr_round_up(ix->n_rows, 64) / 64;
    G.n_pad = G.n_tiles * 64;
    int64_t max_waves = (int64_t)resident_blocks * (RR_SCAN_THREADS / 64);
    // (rr_index_set_scan_cus: the scans' stream is masked to scan_cus of the device's CUs -- a resident grid is that share)
    if (ix->scan_cus > 0 && ix->n_cus > 0) {
        max_waves = max_waves * ix->scan_cus / ix->n_cus;
        if (max_waves < 1) max_waves = 1;
    }
    G.tiles_per_wave = (G.n_tiles + max_waves - 1) / max_waves;
    G.n_waves = (int32_t)((G.n_tiles + G.tiles_per_wave - 1) / G.tiles_per_wave);
    G.qs = 0;
    G.mm_pairs = 0;
    G.gpw = 1;
    G.tiles_per_group = G.tiles_per_wave;
    return G;
}

int rr_scan_events_begin(rr_index* ix, hipStream_t st) {
    const int slot = (int)(ix->ring_head % rr_index::kRing);
    hipEventRecord(ix->ring0[slot], st);
    return slot;
}
void rr_scan_events_end(rr_index* ix, int slot, hipStream_t st) {
    hipEventRecord(ix->ring1[slot], st);
    ix->ring_head++;
    if (ix->ring_head - ix->ring_tail > rr_index::kRing) ix->ring_tail = ix->ring_head - rr_index::kRing;
}
void rr_launch_select(rr_index* ix, const rr_scan_geom& G, int nq, int pool, int64_t* d_rows,
                      float* d_scores, hipStream_t st, const int32_t* only_if, int slices, int64_t sims_slice,
                      int64_t gmax_slice, int64_t smax_slice) {
    const dim3 grid(slices > 1 ? RR_MFMA_MAXQ : nq, slices > 1 ? slices : 1);
    hipLaunchKernelGGL(rr_select, grid, dim3(RR_SEL_THREADS), 0, st, G, ix->d_sims, ix->d_gmax,
                       ix->d_smax, pool, ix->row_offset, d_rows, d_scores, ix->d_sel_trace, only_if, nq, sims_slice,
                       gmax_slice, smax_slice, (const int32_t*)nullptr, (int32_t*)nullptr);
}
int rr_resident_waves(const void* kernel, int threads, int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1)
        per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1)
        cus = 256;
    int waves = per_cu * cus * (threads / 64);
    return waves > RR_MAX_SCAN_WAVES ? RR_MAX_SCAN_WAVES : waves;
}

template <int NB>
static rr_scan_geom rr_launch_scan(rr_index* ix, const float* d_q, hipStream_t st) {
    const f32x4* mat = reinterpret_cast<const f32x4*>(ix->d_matrix);
    const int nf = ix->dim_pad / 64;
    static int cap6 = 0, capg = 0;   // per template instance
    rr_scan_geom G;
    if (nf == 6) {
        if (!cap6) cap6 = rr_resident_grid(rr_scan_f32<6, NB>, ix->device);
        G = rr_make_geom(ix, cap6);
        const int grid = (G.n_waves + 3) / 4;
        hipLaunchKernelGGL((rr_scan_f32<6, NB>), dim3(grid), dim3(RR_SCAN_THREADS), 0, st, mat, G, d_q,
                           ix->d_sims, ix->d_gmax, ix->d_smax, (const int32_t*)nullptr, 0, (int32_t*)nullptr);
    } else {
        if (!capg) capg = rr_resident_grid(rr_scan_f32_generic<NB>, ix->device);
        G = rr_make_geom(ix, capg);
        const int grid = (G.n_waves + 3) / 4;
        hipLaunchKernelGGL((rr_scan_f32_generic<NB>), dim3(grid), dim3(RR_SCAN_THREADS), 0, st, mat, G, nf,
                           d_q, ix->d_sims, ix->d_gmax, ix->d_smax);
    }
    return G;
}

// The filter path's flagged queries (candidate lists overflowed: massive ties, a crowded cut), when a call has at most eight:
// the single-query scan's per-row chain over the whole matrix + the stored-score selection, bit for bit what a batch of
// one returns; their flags come down.  More than eight: all of them go on to the split-operand pass.  fp32 storage, dim 384.
int rr_dense_listed_fallback(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows, float* d_scores,
                             int32_t* flags, hipStream_t st) {
    static const bool off = getenv("RR_NO_CHAIN_FALLBACK") != nullptr;
    if (off || ix->dtype != RR_DTYPE_F32 || ix->dim_pad != 384 || !ix->d_flag_list) return RR_OK;
    static int cap = 0;
    if (!cap) cap = rr_resident_grid(rr_scan_f32<6, 8, true>, ix->device);
    rr_scan_geom G = rr_make_geom(ix, cap);
    hipLaunchKernelGGL((rr_scan_f32<6, 8, true>), dim3((G.n_waves + 3) / 4), dim3(RR_SCAN_THREADS), 0, st,
                       reinterpret_cast<const f32x4*>(ix->d_matrix), G, d_q, ix->d_sims, ix->d_gmax, ix->d_smax,
                       (const int32_t*)flags, nq, ix->d_flag_list);
    hipLaunchKernelGGL(rr_select, dim3(8), dim3(RR_SEL_THREADS), 0, st, G, ix->d_sims, ix->d_gmax, ix->d_smax, pool,
                       ix->row_offset, d_rows, d_scores, ix->d_sel_trace, (const int32_t*)nullptr, 8, (int64_t)0, (int64_t)0,
                       (int64_t)0, (const int32_t*)ix->d_flag_list, flags);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

// Scan + select for up to 8 queries already on the device (padded to dim_pad).
static int rr_dense_chunk(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                          float* d_scores, hipStream_t st, bool time_it) {
    if (time_it) hipEventRecord(ix->ev0, st);
    const int slot = (int)(ix->ring_head % rr_index::kRing);
    hipEventRecord(ix->ring0[slot], st);
    rr_scan_note(ix, 1, nq <= 1 ? 1 : nq <= 2 ? 2 : nq <= 4 ? 4 : 8, nq, 0);
    rr_scan_geom G;
    switch (nq) {
        case 1: G = rr_launch_scan<1>(ix, d_q, st); break;
        case 2: G = rr_launch_scan<2>(ix, d_q, st); break;
        case 3: case 4: G = rr_launch_scan<4>(ix, d_q, st); break;
        default: G = rr_launch_scan<8>(ix, d_q, st); break;
    }
    hipEventRecord(ix->ring1[slot], st);
    ix->ring_head++;
    if (ix->ring_head - ix->ring_tail > rr_index::kRing) ix->ring_tail = ix->ring_head - rr_index::kRing;
    if (time_it) {
        hipEventRecord(ix->ev1, st);
        ix->timing_valid = true;
    }
    hipLaunchKernelGGL(rr_select, dim3(nq), dim3(RR_SEL_THREADS), 0, st, G, ix->d_sims, ix->d_gmax,
                       ix->d_smax, pool, ix->row_offset, d_rows, d_scores, ix->d_sel_trace, (const int32_t*)nullptr, nq, (int64_t)0, (int64_t)0, (int64_t)0,
                       (const int32_t*)nullptr, (int32_t*)nullptr);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

// Scan + select for 9..64 queries on the matrix cores (fp32 storage, dim 384 only).
template <int NQT>
static int rr_dense_chunk_mfma(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                               float* d_scores, hipStream_t st) {
    constexpr int THREADS = NQT == 4 ? 512 : 256;
    static int cap_waves = 0;
    if (!cap_waves) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rr_scan_mfma_f32<NQT>, THREADS, 0) != hipSuccess ||
            per_cu < 1)
            per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix->device) != hipSuccess || cus < 1)
            cus = 256;
        cap_waves = per_cu * cus * (THREADS / 64);
        if (cap_waves > RR_MAX_SCAN_WAVES) cap_waves = RR_MAX_SCAN_WAVES;
    }
    rr_scan_geom G = rr_make_geom(ix, cap_waves / 4);  // one run of tiles per wave, like rr_scan_f32
    G.qs = 16 * NQT;
    const int slot = (int)(ix->ring_head % rr_index::kRing);
    hipEventRecord(ix->ring0[slot], st);
    rr_scan_note(ix, 6, NQT, nq, 0);
    hipLaunchKernelGGL((rr_scan_mfma_f32<NQT>), dim3((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), dim3(THREADS), 0, st,
                       reinterpret_cast<const f32x4*>(ix->d_matrix), G, d_q, ix->d_sims, ix->d_gmax, ix->d_smax);
    hipEventRecord(ix->ring1[slot], st);
    ix->ring_head++;
    if (ix->ring_head - ix->ring_tail > rr_index::kRing) ix->ring_tail = ix->ring_head - rr_index::kRing;
    hipLaunchKernelGGL(rr_select, dim3(nq), dim3(RR_SEL_THREADS), 0, st, G, ix->d_sims, ix->d_gmax,
                       ix->d_smax, pool, ix->row_offset, d_rows, d_scores, ix->d_sel_trace, (const int32_t*)nullptr, nq, (int64_t)0, (int64_t)0, (int64_t)0,
                       (const int32_t*)nullptr, (int32_t*)nullptr);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

// Orders this call's use of the handle's scratch behind the previous call's when the stream changed.  Two resources:
// the ACTIVE slot's scan state (written by its scan, read -- and, by the fallbacks, overwritten -- by the parts of its
// selection) and the selection scratch all slots share (M-tile lists, rescored rows, stored scores): `select` says whether
// the call touches the second.  Each records the last launch sequence that used it; a call on another stream waits for it.
static int rr_scratch_enter(rr_index* ix, hipStream_t st, bool scan = true, bool select = true) {
    (void)scan;
    const int s = ix->cur_slot;
    if (ix->slot_has[s] && ix->slot_stream[s] != st) RR_HIP_TRY(hipStreamWaitEvent(st, ix->slot_ev[s], 0));
    if (select && ix->has_done && st != ix->last_stream) RR_HIP_TRY(hipStreamWaitEvent(st, ix->ev_done, 0));
    return RR_OK;
}
static int rr_scratch_leave(rr_index* ix, hipStream_t st, bool scan = true, bool select = true) {
    (void)scan;
    const int s = ix->cur_slot;
    RR_HIP_TRY(hipEventRecord(ix->slot_ev[s], st));
    ix->slot_stream[s] = st;
    ix->slot_has[s] = true;
    if (select) {
        RR_HIP_TRY(hipEventRecord(ix->ev_done, st));
        ix->last_stream = st;
        ix->has_done = true;
    }
    return RR_OK;
}

static int rr_dense_topk_impl(rr_index* ix, const float* d_q_padded, int nq, int pool,
                              int64_t* d_rows, float* d_scores, hipStream_t st) {
    RR_HIP_TRY(hipSetDevice(ix->device));
    const bool bf16 = ix->dtype == RR_DTYPE_BF16;
    const bool mfma_ok = ix->dim_pad == 384 && ix->n_rows >= 64;
    // batches run on the bf16 matrix cores with exactly-split operands (rr_dense_x3.hip);
    // RR_SCAN_F32_CHAIN=1 selects the f32-input MFMA kernels (scores = pure fmaf chains) instead
    static const bool f32_chain = getenv("RR_SCAN_F32_CHAIN") != nullptr;
    static const bool exact_scan = getenv("RR_SCAN_EXACT") != nullptr;
    // measured crossover (10M rows): the per-row-chain VALU scans win up to 4 queries per read
    // (2.4-2.5 ms fp32, 1.3-1.5 ms bf16); from 5 on the matrix-core scan does (2.7-2.9 / 1.8 ms
    // for up to 16 queries, against 3.0 / 2.4 ms for the 8-query VALU scan)
    const int valu_max = mfma_ok ? 4 : 8;
    // (score scratch: one 64-query slice per 64 queries of a filter call -- its flagged queries are served in slices of one
    //  launch pair, rr_dense_x3w_fallback_all)
    const int scratch_slots = (int)rr_round_up(nq < RR_SEL_MAXQ ? nq : RR_SEL_MAXQ, RR_MFMA_MAXQ);
    int rc = rr_ensure_scratch(ix, (mfma_ok && nq > valu_max) ? scratch_slots : 8);
    if (rc) return rc;
    int q0 = 0;
    while (q0 < nq) {
        const int left = nq - q0;
        const float* q = d_q_padded + (int64_t)q0 * ix->dim_pad;
        int64_t* rows = d_rows + (int64_t)q0 * pool;
        float* scores = d_scores + (int64_t)q0 * pool;
        int n;
        bool small = false;        // the filter path declined: too few tiles
        if (mfma_ok && left > valu_max && !f32_chain && !exact_scan && ix->scan_mode == RR_SCAN_MODE_DEFAULT &&
            (ix->n_rows + 63) / 64 < 8 * (int64_t)pool)
            small = true;
        if (mfma_ok && left > valu_max && !small) {
            // 5..128 queries share one read of the matrix: bf16 filter scan + exact rescoring of the
            // candidates (rr_dense_flt.hip); without a finite row-norm bound, or with RR_SCAN_EXACT=1 /
            // RR_SCAN_MODE_STORED, the exact split-operand scans, 64 queries per read
            n = left < RR_MFMA_MAXQ ? left : RR_MFMA_MAXQ;
            bool done = false;
            if (!f32_chain && !exact_scan && ix->scan_mode == RR_SCAN_MODE_DEFAULT) {
                // up to 128 queries per scan launch; 193 .. 256 remaining queries go as a PAIR of launches that share one
                // selection / rescoring sequence (the second part must still use the 128-slot kernel: > 64 queries)
                static const bool no_pair = getenv("RR_NO_PAIR") != nullptr;
                int nf = left < RR_FLT_MAXQ ? left : RR_FLT_MAXQ;
                if (!no_pair && left > RR_FLT_MAXQ + RR_MFMA_MAXQ) nf = left < RR_SEL_MAXQ ? left : RR_SEL_MAXQ;
                rc = rr_dense_chunk_flt(ix, q, nf, pool, rows, scores, st);
                if (rc != RR_FLT_NO_BOUND && rc != RR_FLT_SMALL) {
                    done = true;
                    n = nf;
                }
            }
            if (done) {
            } else if (!f32_chain) rc = rr_dense_chunk_x3(ix, q, n, pool, rows, scores, st);
            else if (bf16) rc = rr_dense_chunk_mfma_bf16(ix, q, n, pool, rows, scores, st);
            else if (n <= 16) rc = rr_dense_chunk_mfma<1>(ix, q, n, pool, rows, scores, st);
            else if (n <= 32) rc = rr_dense_chunk_mfma<2>(ix, q, n, pool, rows, scores, st);
            else rc = rr_dense_chunk_mfma<4>(ix, q, n, pool, rows, scores, st);
        } else {
            // the VALU scan kernels read NB = 1/2/4/8 query slots; slots past n hold zeros
            const int vmax = small ? 8 : valu_max;
            n = left < vmax ? left : vmax;
            rc = bf16 ? rr_dense_chunk_bf16(ix, q, n, pool, rows, scores, st)
                      : rr_dense_chunk(ix, q, n, pool, rows, scores, st, q0 == 0);
        }
        if (rc) return rc;
        q0 += n;
    }
    return RR_OK;
}

// Pads queries (nq x dim) into the staging buffer (nq_slots x dim_pad, zero filled).
__global__ void rr_pad_queries(const float* __restrict__ src, float* __restrict__ dst, int nq,
                               int dim, int dim_pad, int slots) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)slots * dim_pad) return;
    const int qi = (int)(i / dim_pad), c = (int)(i % dim_pad);
    dst[i] = (qi < nq && c < dim) ? src[(int64_t)qi * dim + c] : 0.f;
}

// Two-phase K1 (include/rr_hip.h): phase 1 = pad + prepare + scan [+ per-query bound, kth > 0] into scan slot `slot`.
extern "C" int rr_dense_scan_slot_dev(rr_index* ix, int32_t slot, const float* d_queries, int32_t n_queries, int32_t pool,
                                      int32_t kth, float* d_bound, int32_t* applied, void* stream) {
    RR_REQUIRE(ix && d_queries && applied, "rr_dense_scan_dev: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH, "rr_dense_scan_dev: n_queries %d out of [1,%d]", n_queries, RR_MAX_BATCH);
    RR_REQUIRE(pool >= 1 && pool <= RR_MAX_POOL && pool <= ix->n_rows, "rr_dense_scan_dev: pool %d out of range", pool);
    RR_REQUIRE(kth >= 0 && kth <= pool && (kth == 0) == (d_bound == nullptr),
               "rr_dense_scan_dev: kth %d out of [1, pool] (or 0 with a NULL bound: no bound wanted)", kth);
    RR_REQUIRE(ix->d_matrix, "rr_dense_scan_dev: index has no matrix");
    *applied = 0;
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    // only the batched filter path has a scan that can be split from its selection (5 .. 256 queries, matrix-core
    // kernels, default scan mode); everything else answers "not applied" and is served by rr_dense_topk_dev
    static const bool exact_scan = getenv("RR_SCAN_EXACT") != nullptr, f32_chain = getenv("RR_SCAN_F32_CHAIN") != nullptr;
    const bool mfma_ok = ix->dim_pad == 384 && ix->n_rows >= 64;
    if (!mfma_ok || n_queries <= 4 || n_queries > RR_SEL_MAXQ || exact_scan || f32_chain || ix->scan_mode != RR_SCAN_MODE_DEFAULT ||
        (n_queries > RR_FLT_MAXQ && n_queries <= RR_FLT_MAXQ + RR_MFMA_MAXQ))       // (129 .. 192 go as two chunks)
        return RR_OK;
    hipStream_t st = (hipStream_t)stream;
    const void* q_vis = nullptr;
    int rc = rr_device_visible(d_queries, &q_vis, "rr_dense_scan_dev (queries)");
    if (rc) return rc;
    d_queries = static_cast<const float*>(q_vis);
    rc = rr_slot_activate(ix, slot);
    if (rc) return rc;
    rc = rr_scratch_enter(ix, st, true, false);
    if (rc) return rc;
    rc = rr_ensure_scratch(ix, (int)rr_round_up(n_queries, RR_MFMA_MAXQ));
    if (rc) return rc;
    const int slots = (int)rr_round_up(n_queries, RR_MFMA_MAXQ);
    const int64_t total = (int64_t)slots * ix->dim_pad;
    // (padding + planes + bounds in one launch; without a finite row-norm bound the plain padding, and the call is declined below)
    if (rr_flt_pad_prep(ix, d_queries, n_queries, slots, st) != RR_OK)
        hipLaunchKernelGGL(rr_pad_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           d_queries, ix->d_q, n_queries, ix->dim, ix->dim_pad, slots);
    rc = rr_dense_chunk_flt(ix, ix->d_q, n_queries, pool, nullptr, nullptr, st, 1, kth, d_bound, nullptr);
    if (rc == RR_FLT_NO_BOUND || rc == RR_FLT_SMALL) return rr_scratch_leave(ix, st, true, false);     // not applied
    if (rc) return rc;
    *applied = 1;
    return rr_scratch_leave(ix, st, true, false);
}

// phase 2 = selection / rescoring / ordering of the scan phase 1 left in `slot`, M-tiles opened no further down than `d_floor`.
// `parts` (include/rr_hip.h): RR_SELECT_LIST | RR_SELECT_RESCORE | RR_SELECT_ORDER, each possibly on a stream of its own.
extern "C" int rr_dense_select_part_dev(rr_index* ix, int32_t slot, int32_t parts, int32_t n_queries, int32_t pool,
                                        const float* d_floor, int64_t* d_out_rows, float* d_out_scores, void* stream) {
    RR_REQUIRE(ix, "rr_dense_select_dev: NULL handle");
    RR_REQUIRE(parts >= 1 && parts <= 7, "rr_dense_select_dev: parts %d is not a combination of RR_SELECT_LIST | _RESCORE | _ORDER", parts);
    RR_REQUIRE(!(parts & 4) || (d_out_rows && d_out_scores), "rr_dense_select_dev: NULL output");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    int rc = rr_slot_activate(ix, slot);
    if (rc) return rc;
    rc = rr_scratch_enter(ix, st, false, true);
    if (rc) return rc;
    rc = rr_dense_chunk_flt(ix, ix->d_q, n_queries, pool, d_out_rows, d_out_scores, st, 2, 0, nullptr, d_floor, parts);
    RR_REQUIRE(rc != RR_FLT_SMALL, "rr_dense_select_dev: no scan of these %d queries (pool %d) is pending in slot %d: call "
               "rr_dense_scan_dev first and use its `applied` answer", n_queries, pool, slot);
    if (rc) return rc;
    return rr_scratch_leave(ix, st, false, true);
}

extern "C" int rr_dense_select_slot_dev(rr_index* ix, int32_t slot, int32_t n_queries, int32_t pool, const float* d_floor,
                                        int64_t* d_out_rows, float* d_out_scores, void* stream) {
    return rr_dense_select_part_dev(ix, slot, 7, n_queries, pool, d_floor, d_out_rows, d_out_scores, stream);
}

extern "C" int rr_dense_scan_dev(rr_index* ix, const float* d_queries, int32_t n_queries, int32_t pool, int32_t kth,
                                 float* d_bound, int32_t* applied, void* stream) {
    RR_REQUIRE(kth >= 1 && d_bound, "rr_dense_scan_dev: kth %d out of [1, pool] / NULL bound", kth);
    return rr_dense_scan_slot_dev(ix, 0, d_queries, n_queries, pool, kth, d_bound, applied, stream);
}

extern "C" int rr_dense_select_dev(rr_index* ix, const float* d_queries, int32_t n_queries, int32_t pool,
                                   const float* d_floor, int64_t* d_out_rows, float* d_out_scores, void* stream) {
    RR_REQUIRE(d_queries, "rr_dense_select_dev: NULL argument");
    return rr_dense_select_slot_dev(ix, 0, n_queries, pool, d_floor, d_out_rows, d_out_scores, stream);
}

extern "C" int rr_dense_topk_slot_dev(rr_index* ix, int32_t slot, const float* d_queries, int32_t n_queries,
                                      int32_t pool, int64_t* d_out_rows, float* d_out_scores,
                                      void* stream) {
    RR_REQUIRE(ix && d_queries && d_out_rows && d_out_scores, "rr_dense_topk_dev: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH, "rr_dense_topk_dev: n_queries %d out of [1,%d]",
               n_queries, RR_MAX_BATCH);
    RR_REQUIRE(pool >= 1 && pool <= RR_MAX_POOL && pool <= ix->n_rows,
               "rr_dense_topk_dev: pool %d out of [1,min(%d,n_rows=%lld)]", pool, RR_MAX_POOL,
               (long long)ix->n_rows);
    RR_REQUIRE(ix->d_matrix, "rr_dense_topk_dev: index has no matrix");
    RR_REQUIRE(ix->dtype == RR_DTYPE_F32 || ix->dim_pad == 384,
               "rr_dense_topk_dev: bf16 storage is built for dim 384 only");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream
    int rc0 = rr_slot_activate(ix, slot);
    if (rc0) return rc0;
    rr_flt_drop_pending(ix);
    // (the queries may sit in pinned host memory: rr_pad_queries then reads them over PCIe, once -- no copy command)
    const void* q_vis = nullptr;
    int rc = rr_device_visible(d_queries, &q_vis, "rr_dense_topk_dev (queries)");
    if (rc) return rc;
    d_queries = static_cast<const float*>(q_vis);
    rc = rr_scratch_enter(ix, st);
    if (rc) return rc;
    const int slots = (int)rr_round_up(n_queries, RR_MFMA_MAXQ);   // kernels read whole query tiles
    const int64_t total = (int64_t)slots * ix->dim_pad;
    // a call the batched filter path will serve in one chunk: padding + bf16 planes + bounds in ONE launch
    static const bool plain_front = getenv("RR_SCAN_EXACT") != nullptr || getenv("RR_SCAN_F32_CHAIN") != nullptr;
    const bool filter_call = ix->dim_pad == 384 && ix->n_rows >= 64 && n_queries > 4 && n_queries <= RR_SEL_MAXQ && !plain_front &&
                             ix->scan_mode == RR_SCAN_MODE_DEFAULT && (ix->n_rows + 63) / 64 >= 8 * (int64_t)pool;
    if (!filter_call || rr_flt_pad_prep(ix, d_queries, n_queries, slots, st) != RR_OK)
        hipLaunchKernelGGL(rr_pad_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           d_queries, ix->d_q, n_queries, ix->dim, ix->dim_pad, slots);
    rc = rr_dense_topk_impl(ix, ix->d_q, n_queries, pool, d_out_rows, d_out_scores, st);
    ix->flt_prep_fresh = false;
    if (rc) return rc;
    return rr_scratch_leave(ix, st);
}

extern "C" int rr_dense_topk_dev(rr_index* ix, const float* d_queries, int32_t n_queries,
                                 int32_t pool, int64_t* d_out_rows, float* d_out_scores,
                                 void* stream) {
    return rr_dense_topk_slot_dev(ix, 0, d_queries, n_queries, pool, d_out_rows, d_out_scores, stream);
}

extern "C" int rr_dense_topk(rr_index* ix, const float* h_queries, int32_t n_queries, int32_t pool,
                             int64_t* h_out_rows, float* h_out_scores, int32_t* pool_out) {
    RR_REQUIRE(ix && h_queries && pool_out, "rr_dense_topk: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH, "rr_dense_topk: n_queries %d out of [1,%d]",
               n_queries, RR_MAX_BATCH);
    RR_REQUIRE(pool >= 0, "rr_dense_topk: negative pool");
    // utils.py:116-117: top_k is clamped to the number of rows
    int eff = pool;
    if ((int64_t)eff > ix->n_rows) eff = (int)ix->n_rows;
    RR_REQUIRE(eff <= RR_MAX_POOL, "rr_dense_topk: pool %d exceeds RR_MAX_POOL %d", eff, RR_MAX_POOL);
    *pool_out = eff;
    if (eff == 0) return RR_OK;  // top_k == 0 returns two empty arrays
    RR_REQUIRE(h_out_rows && h_out_scores, "rr_dense_topk: NULL output");
    RR_REQUIRE(ix->d_matrix, "rr_dense_topk: index has no matrix");
    RR_REQUIRE(ix->dtype == RR_DTYPE_F32 || ix->dim_pad == 384,
               "rr_dense_topk: bf16 storage is built for dim 384 only");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    // The scan scratch of the handle is shared with rr_dense_topk_dev callers, who run on the
    // device's default stream unless they say otherwise: use that stream here too so two host
    // threads (one on each entry point) stay ordered on the scratch.
    hipStream_t st = nullptr;
    int rce = rr_slot_activate(ix, 0);
    if (rce) return rce;
    rce = rr_scratch_enter(ix, st);
    if (rce) return rce;
    const int slots = (int)rr_round_up(n_queries, RR_MFMA_MAXQ);   // kernels read whole query tiles
    RR_HIP_TRY(hipMemsetAsync(ix->d_q, 0, sizeof(float) * (size_t)slots * ix->dim_pad, st));
    RR_HIP_TRY(hipMemcpy2DAsync(ix->d_q, sizeof(float) * ix->dim_pad, h_queries,
                                sizeof(float) * ix->dim, sizeof(float) * ix->dim, n_queries,
                                hipMemcpyHostToDevice, st));
    int rc = rr_dense_topk_impl(ix, ix->d_q, n_queries, eff, ix->d_rows_out, ix->d_scores_out, st);
    if (rc) return rc;
    RR_HIP_TRY(hipMemcpyAsync(h_out_rows, ix->d_rows_out, sizeof(int64_t) * (size_t)n_queries * eff,
                              hipMemcpyDeviceToHost, st));
    RR_HIP_TRY(hipMemcpyAsync(h_out_scores, ix->d_scores_out, sizeof(float) * (size_t)n_queries * eff,
                              hipMemcpyDeviceToHost, st));
    RR_HIP_TRY(hipStreamSynchronize(st));
    return rr_scratch_leave(ix, st);
}

extern "C" int rr_index_last_scan_ms(rr_index* ix, float* out_ms) {
    RR_REQUIRE(ix && out_ms, "rr_index_last_scan_ms: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_REQUIRE(ix->timing_valid, "rr_index_last_scan_ms: no scan has been timed yet");
    RR_HIP_TRY(hipSetDevice(ix->device));
    RR_HIP_TRY(hipEventSynchronize(ix->ev1));
    RR_HIP_TRY(hipEventElapsedTime(out_ms, ix->ev0, ix->ev1));
    return RR_OK;
}

extern "C" int rr_index_last_scan_info(rr_index* ix, int32_t* out8) {
    RR_REQUIRE(ix && out8, "rr_index_last_scan_info: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    for (int i = 0; i < 8; ++i) out8[i] = i < 5 ? ix->last_scan[i] : 0;
    return RR_OK;
}

extern "C" int rr_index_set_shadow(rr_index* ix, int32_t enable) {
    RR_REQUIRE(ix != nullptr, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->use_shadow = enable ? 1 : 0;
    if (!enable && ix->d_shadow) {
        RR_HIP_TRY(hipSetDevice(ix->device));
        RR_HIP_TRY(hipDeviceSynchronize());
        hipFree(ix->d_shadow);
        ix->d_shadow = nullptr;
        ix->shadow_valid = false;
    }
    return RR_OK;
}

extern "C" int rr_index_set_scan_mode(rr_index* ix, int32_t mode) {
    RR_REQUIRE(ix != nullptr, "null index");
    RR_REQUIRE(mode == RR_SCAN_MODE_DEFAULT || mode == RR_SCAN_MODE_STORED, "unknown scan mode %d", mode);
    ix->scan_mode = mode;
    return RR_OK;
}

extern "C" int rr_index_select_trace(rr_index* ix, int32_t* out4) {
    RR_REQUIRE(ix && out4, "rr_index_select_trace: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    RR_HIP_TRY(hipDeviceSynchronize());
    RR_HIP_TRY(hipMemcpy(out4, ix->d_sel_trace, sizeof(int32_t) * 16, hipMemcpyDeviceToHost));
    return RR_OK;
}

extern "C" int rr_index_scan_stats(rr_index* ix, double* out_total_ms, int64_t* out_launches) {
    RR_REQUIRE(ix && out_total_ms && out_launches, "rr_index_scan_stats: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    double total = 0.0;
    int64_t n = 0;
    for (; ix->ring_tail < ix->ring_head; ++ix->ring_tail) {
        const int slot = (int)(ix->ring_tail % rr_index::kRing);
        RR_HIP_TRY(hipEventSynchronize(ix->ring1[slot]));
        float ms = 0.f;
        RR_HIP_TRY(hipEventElapsedTime(&ms, ix->ring0[slot], ix->ring1[slot]));
        total += ms;
        ++n;
    }
    *out_total_ms = total;
    *out_launches = n;
    return RR_OK;
}

int rr_l2norm_rows_f32(rr_index* ix, int64_t first_row, int64_t n, float eps, hipStream_t st) {
    if (n == 0) return RR_OK;
    hipLaunchKernelGGL(rr_l2norm_f32, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st,
                       (float*)ix->d_matrix + first_row * ix->dim_pad, n, ix->dim_pad, eps);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

extern "C" int rr_index_l2_normalize(rr_index* ix, float eps) {
    RR_REQUIRE(ix && ix->d_matrix, "rr_index_l2_normalize: index has no matrix");
    RR_REQUIRE(ix->dtype == RR_DTYPE_F32,
               "rr_index_l2_normalize: a bf16 index is normalised in fp32 before rounding (rr_index_upload_rows_f32)");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    rr_matrix_written(ix);
    const int rows_per_block = 4;
    const unsigned grid = (unsigned)((ix->n_rows + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(rr_l2norm_f32, dim3(grid), dim3(64 * rows_per_block), 0, ix->stream,
                       (float*)ix->d_matrix, ix->n_rows, ix->dim_pad, eps);
    RR_HIP_TRY(hipGetLastError());
    RR_HIP_TRY(hipStreamSynchronize(ix->stream));
    return RR_OK;
}
