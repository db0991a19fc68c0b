// rr_dense_x3.hip -- K1 batched scan on the bf16 matrix cores with split operands.
//
// v_mfma_f32_16x16x32_bf16 runs at 16x the rate of the f32-input MFMA.  An fp32 number is the
// exact sum of three bf16 numbers (8 + 8 + 8 significant bits: x1 = top 16 bits of x,
// x2 = top 16 bits of x - x1, x3 = x - x1 - x2, every subtraction exact), and a product of two
// bf16 numbers is exact in fp32.  So  a*q = sum_{i,j} a_i*q_j  with every partial product exact;
// the three smallest of the nine terms (a2*q3, a3*q2, a3*q3 <= 2^-24 |a*q|, below fp32's own
// rounding of the product) are dropped:
//   fp32 matrix  : a1q1 + a1q2 + a2q1 + a2q2 + a1q3 + a3q1   6 MFMAs per 32 dims ( 96 cycles; f32 MFMA: 256)
//   bf16 matrix  : a q1 + a q2 + a q3 (a is one term, exact)  3 MFMAs per 32 dims ( 48 cycles)
// Accumulation is the matrix core's fp32 accumulator.  The result is not an fmaf chain any more,
// but its error (~1e-8 on unit vectors) is that of fp32 rounding itself; the parity bar (rows
// exact wherever the float64 gap exceeds 4e-7, scores within 1e-5) is the same as for every
// other scan, and every score is still independent of where the row sits, of the shard and of
// the grid (one fixed instruction sequence per 16-row M-tile x 16-query tile).
//
// Stream structure = rr_scan_mfma_f32: every wave walks its own run of 64-row tiles, fragment
// loads straight into a VGPR ring (fp32: for each of the 12 K-blocks lane (r, kg) loads floats
// [32b+4kg, +4) and [32b+16+4kg, +4) of row r, so the four lanes of a row read 64 contiguous bytes
// per instruction; the query planes are permuted to the same k order), inline-asm loads with
// counted waits, M-tile-major scores.  The queries are
// split once per launch by rr_split_queries into three bf16 planes that each workgroup copies
// to LDS, XOR-swizzled in 16-B units (unit u of query q sits at u ^ (q & 15)) so the B-fragment
// ds_read_b128 is conflict-free.  64 queries x 3 planes = 144 KB: one 512-thread workgroup per
// CU, 8 waves x 24 KB in flight.
#include "rr_x3.h"

// queries (slots x 384 fp32) -> planes[3][slots][384] bf16 (truncating split, exact sum), in the k
// order `order` (rr_x3.h) inside every 32-dim group.
__global__ void rr_split_queries(const float* __restrict__ q, unsigned short* __restrict__ planes, int slots,
                                 int order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots * 384) return;
    int src = i;
    if (order == RR_X3_ORDER_PAIR64) {
        const int e = i & 31, kg = e >> 3, j = e & 7;
        src = (i & ~31) + (j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4));
    } else if (order == RR_X3_ORDER_WIDE_BF16) {
        const int e = i & 31, v = e >> 4, h = (e >> 3) & 1, j = e & 7;
        src = (i & ~31) + 16 * h + 8 * v + j;
    }
    const float x = q[src];
    const float x1 = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
    const float r1 = x - x1;
    const float x2 = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    const float x3 = r1 - x2;
    planes[i] = (unsigned short)(__float_as_uint(x1) >> 16);
    planes[slots * 384 + i] = (unsigned short)(__float_as_uint(x2) >> 16);
    planes[2 * slots * 384 + i] = (unsigned short)(__float_as_uint(x3) >> 16);
}
void rr_launch_split_queries(const float* d_q, unsigned short* planes, int slots, int order, hipStream_t st) {
    hipLaunchKernelGGL(rr_split_queries, dim3((slots * 384 + 255) / 256), dim3(256), 0, st, d_q, planes, slots, order);
}

template <bool A_BF16>
__device__ __forceinline__ void rr_x3_mma(const rr_x3_afrag& f, bf16x8 q1, bf16x8 q2, bf16x8 q3, f32x4& acc) {
    if (A_BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, q3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, q2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, q1, acc, 0, 0, 0);
    } else {                                                 // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a3, q1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, q3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a2, q2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a2, q1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, q2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a1, q1, acc, 0, 0, 0);
    }
}

// STORE = true writes every score (M-tile-major) as well as the tile / group maxima: the fallback
// for batches whose candidate lists overflow the rescoring path, launched with `fallback` flags and
// returning at once when no query of the batch raised its flag.  STORE = false (the normal scan)
// writes only the maxima: the score stores cost more than a third of the scan time
// (tools/x3 ablation: 3.35 -> 2.52 ms at 64 queries, 2.78 -> 2.43 ms at 16), the candidate tiles are
// rescored afterwards by rr_rescore_x3 instead (~1 % of the matrix per 64 queries).
template <int NQT, bool A_BF16, bool STORE>
__global__ __launch_bounds__((NQT == 4 ? 512 : 256), 2) void rr_scan_mfma_x3(
    const u32x4* __restrict__ mat, rr_scan_geom G, const u32x4* __restrict__ planes,  // [3][16*NQT][48] units
    float* __restrict__ sims, float* __restrict__ gmax, uint32_t* __restrict__ smax,
    const int32_t* __restrict__ fallback, int n_flags) {
    constexpr int THREADS = NQT == 4 ? 512 : 256;
    if (STORE && fallback) {
        int any = 0;
        for (int i = 0; i < n_flags; ++i) any |= fallback[i];
        if (!any) return;                                  // wave-uniform: nobody needs the stored scores
    }
    constexpr int QN = 16 * NQT;
    constexpr int ROWU = A_BF16 ? 48 : 96;            // 16-byte units per matrix row
    constexpr int RING = A_BF16 ? 12 : 24;            // units a lane holds per M-tile
    __shared__ u32x4 qs[3 * QN * RR_X3_UNITS];
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * QN * RR_X3_UNITS; i += THREADS) {
        const int u = i % RR_X3_UNITS, pq = i / RR_X3_UNITS;      // pq = plane * QN + query
        qs[pq * RR_X3_UNITS + (u ^ (pq & 15))] = planes[i];       // QN % 16 == 0: pq & 15 == query & 15
    }
    __syncthreads();

    const int lane = tid & 63;
    const int r = lane & 15;
    const int kg = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (THREADS / 64) + (tid >> 6);
    if (wave >= G.n_waves) return;
    const int64_t t0 = wave * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;
    const int64_t m0 = t0 * 4, m1 = t1 * 4;

    // LOAD mapping: lane L reads row (L >> 2), 16-byte piece (L & 3) of each 64-byte group: four
    // ADJACENT lanes cover 64 contiguous bytes, which the vector L1 looks up once; with the MFMA
    // mapping (row = lane & 15) adjacent lanes are 16 different rows = 64 lookups of 16 B per
    // instruction, and the L1 tag rate, not HBM, paced the scan (TA_ADDR_STALLED_BY_TC 80 % of the
    // time).  The registers are moved to the MFMA mapping (lane r + 16*kg <- lane 4*r + kg) with
    // ds_bpermute after they land.
    const int lrow = lane >> 2, lkg = lane & 3;
    const int bperm_src = 4 * (4 * r + kg);          // byte address of the source lane for ds_bpermute
    auto row_ptr = [&](int64_t mt) {
        mt = mt < m1 ? mt : m1 - 1;
        int64_t row = mt * 16 + lrow;
        row = row < G.n_rows ? row : G.n_rows - 1;
        return mat + row * ROWU + lkg;
    };
    auto to_mfma_lanes = [&](u32x4 v) {
        u32x4 o;
        o.x = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.x);
        o.y = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.y);
        o.z = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.z);
        o.w = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.w);
        return o;
    };
    // unit j of the ring: fp32: K-block j/2, float4 (kg + 4*(j&1)) of it: the four lanes of a row read
    //                           64 contiguous bytes (half a line) per instruction -> offset 128*(j/2) + 64*(j&1);
    //                     bf16: K-block j, unit kg of it (64 contiguous bytes)    -> offset 64*j
#define RR_X3_OFF(j) (A_BF16 ? 64 * (j) : 128 * ((j) / 2) + 64 * ((j) & 1))
#define RR_X3_LOAD(dst, ptr, j) \
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(RR_X3_OFF(j)) : "memory")
    u32x4 a[RING];
    {
        const u32x4* p = row_ptr(m0);
#pragma unroll
        for (int j = 0; j < RING; ++j) RR_X3_LOAD(a[j], p, j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float tile_max[NQT], gm[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) tile_max[t] = gm[t] = -INFINITY;

#pragma unroll 1
    for (int64_t mt = m0; mt < m1; ++mt) {
        const u32x4* pn = row_ptr(mt + 1);
        const bool maxima_pending = (mt & 3) == 0 && mt > m0;
        f32x4 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 12; ++b) {
            // B fragments first: their LDS latency hides behind the wait for the matrix units and
            // (fp32) the operand split.  TB query tiles at a time (register budget), the next
            // group's reads issued before the current group's MFMAs.
            constexpr int TB = (!A_BF16 && NQT == 4) ? 2 : NQT;
            const int u = 4 * b + kg;                         // 16-byte unit of the query planes
            bf16x8 qf[NQT][3];
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    qf[t][pl] = __builtin_bit_cast(bf16x8, qs[(pl * QN + 16 * t + r) * RR_X3_UNITS + (u ^ r)]);
            if (b % 6 == 0) {
                // The ring is refilled in two bursts per M-tile (units of K-blocks 0-5 after block 5,
                // of K-blocks 6-11 after block 11): a burst asks for 6 adjacent 128-B lines of each
                // of the 16 rows at once, which keeps the DRAM pages open, where one line per row
                // every K-block does not.  Younger operations than the burst this half needs: the
                // other half's burst, and the NQT stores of the previous M-tile's epilogue (score
                // stores when STORE -- its maxima only make the wait conservative --, else the
                // M-tile maxima written after every fourth M-tile).
                constexpr int H = RING / 2;
                if (STORE || maxima_pending) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H + NQT) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H) : "memory");
#pragma unroll
                for (int j = 0; j < H; ++j) asm volatile("" : "+v"(a[(b / 6) * H + j]));   // uses stay below the wait
            }
            const rr_x3_afrag af = A_BF16 ? rr_x3_split<true>(to_mfma_lanes(a[b]), u32x4{})
                                          : rr_x3_split<false>(to_mfma_lanes(a[A_BF16 ? b : 2 * b]),
                                                               to_mfma_lanes(a[A_BF16 ? b : 2 * b + 1]));
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                if (t % TB == 0 && t + TB < NQT) {            // prefetch the next group of query tiles
#pragma unroll
                    for (int t2 = t + TB; t2 < t + 2 * TB; ++t2)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            qf[t2][pl] = __builtin_bit_cast(bf16x8, qs[(pl * QN + 16 * t2 + r) * RR_X3_UNITS + (u ^ r)]);
                }
                rr_x3_mma<A_BF16>(af, qf[t][0], qf[t][1], qf[t][2], acc[t]);
            }
            if (b % 6 == 5) {
                // this half's units of the next M-tile, as one burst, into the registers just consumed
                constexpr int H = RING / 2;
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    const int unit = (b / 6) * H + j;
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2"
                                 : "=v"(a[unit]) : "v"(pn), "n"(RR_X3_OFF(unit)), "v"(acc[NQT - 1]) : "memory");
                }
            }
        }
        // lane (r, kg) holds rows 4*kg .. 4*kg+3 of the M-tile for query 16*t + r
        const int64_t row0 = mt * 16 + 4 * kg;
        const bool tile_end = (mt & 3) == 3;
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const f32x4 v = rr_x3_canon(acc[t], row0, G.n_rows);
            if (STORE) *reinterpret_cast<f32x4*>(sims + ((mt * QN + 16 * t + r) * 16 + 4 * kg)) = v;
            float m4 = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            m4 = fmaxf(m4, __shfl_xor(m4, 16, 64));
            m4 = fmaxf(m4, __shfl_xor(m4, 32, 64));      // maximum of the M-tile for query 16*t + r, in all four kg lanes
            if (STORE) {
                tile_max[t] = fmaxf(tile_max[t], m4);
                if (tile_end) {
                    if (kg == 0) gmax[(mt >> 2) * QN + 16 * t + r] = tile_max[t];
                    gm[t] = fmaxf(gm[t], tile_max[t]);
                    tile_max[t] = -INFINITY;
                }
            } else {
                // lane (r, kg) keeps the maximum of M-tile kg of the 64-row tile: [tile][query][4]
                gm[t] = fmaxf(gm[t], m4);
                if ((int)(mt & 3) == kg) tile_max[t] = m4;
                if (tile_end) gmax[(((mt >> 2) * QN + 16 * t + r) << 2) + kg] = tile_max[t];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (redundant) loads
    if (kg == 0) {
#pragma unroll
        for (int t = 0; t < NQT; ++t) smax[wave * QN + 16 * t + r] = rr_f2key(gm[t]);
    }
}

// Recomputes the 16 scores of every M-tile rr_select_mtiles listed, one wave per (query, M-tile):
// same load mapping, lane permutation, operand split, MFMA order and K-block order as the scan,
// with the query's planes in every B column (an output element depends only on its own row of A
// and column of B), so each score is the scan's bit for bit.  Output: sc[query][slot][16].
template <bool A_BF16>
__global__ __launch_bounds__(256) void rr_rescore_x3(
    const u32x4* __restrict__ mat, int64_t n_rows, const u32x4* __restrict__ planes, int QN,
    const uint32_t* __restrict__ mtiles, const int32_t* __restrict__ count, const int32_t* __restrict__ fb,
    float* __restrict__ sc) {
    constexpr int ROWU = A_BF16 ? 48 : 96;
    constexpr int RING = A_BF16 ? 12 : 24;
    const int q = blockIdx.y;
    if (fb[q]) return;
    const int n = count[q];
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kg = lane >> 4;
    const int lrow = lane >> 2, lkg = lane & 3;
    const int bperm_src = 4 * (4 * r + kg);
    auto to_mfma_lanes = [&](u32x4 v) {
        u32x4 o;
        o.x = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.x);
        o.y = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.y);
        o.z = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.z);
        o.w = (unsigned)__builtin_amdgcn_ds_bpermute(bperm_src, (int)v.w);
        return o;
    };
    for (int slot = blockIdx.x * 4 + (threadIdx.x >> 6); slot < n; slot += gridDim.x * 4) {
        const int64_t mt = mtiles[(int64_t)q * RR_X3_MCAP + slot];
        int64_t row = mt * 16 + lrow;
        row = row < n_rows ? row : n_rows - 1;
        const u32x4* p = mat + row * ROWU + lkg;
        u32x4 a[RING];
#pragma unroll
        for (int j = 0; j < RING; ++j) a[j] = p[RR_X3_OFF(j) / 16];
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 12; ++b) {
            const int u = 4 * b + kg;
            const bf16x8 q1 = __builtin_bit_cast(bf16x8, planes[(0 * QN + q) * RR_X3_UNITS + u]);
            const bf16x8 q2 = __builtin_bit_cast(bf16x8, planes[(1 * QN + q) * RR_X3_UNITS + u]);
            const bf16x8 q3 = __builtin_bit_cast(bf16x8, planes[(2 * QN + q) * RR_X3_UNITS + u]);
            const rr_x3_afrag af = A_BF16 ? rr_x3_split<true>(to_mfma_lanes(a[b]), u32x4{})
                                          : rr_x3_split<false>(to_mfma_lanes(a[A_BF16 ? b : 2 * b]),
                                                               to_mfma_lanes(a[A_BF16 ? b : 2 * b + 1]));
            rr_x3_mma<A_BF16>(af, q1, q2, q3, acc);
        }
        const f32x4 v = rr_x3_canon(acc, mt * 16 + 4 * kg, n_rows);
        if (r == 0) *reinterpret_cast<f32x4*>(sc + ((int64_t)q * RR_X3_MCAP + slot) * 16 + 4 * kg) = v;
    }
}

bool rr_x3_stored_path(const rr_index* ix) {
    if (ix->scan_mode == RR_SCAN_MODE_STORED) return true;
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("RR_X3_STORED");       // diagnostic: single-pass scan that stores every score
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

template <int NQT, bool A_BF16>
static int rr_dense_chunk_x3_t(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                               float* d_scores, hipStream_t st) {
    constexpr int THREADS = NQT == 4 ? 512 : 256;
    constexpr int QN = 16 * NQT;
    static int waves = 0;
    if (!waves) waves = rr_resident_waves((const void*)rr_scan_mfma_x3<NQT, A_BF16, false>, THREADS, ix->device);
    rr_scan_geom G = rr_make_geom(ix, waves / 4);
    G.qs = QN;
    // the split query planes live behind the staged queries in the index's query buffer
    unsigned short* planes = reinterpret_cast<unsigned short*>(ix->d_qplanes);
    const u32x4* mat = reinterpret_cast<const u32x4*>(ix->d_matrix);
    const u32x4* pl4 = reinterpret_cast<const u32x4*>(planes);
    const dim3 grid((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), block(THREADS);
    const rr_x3_scratch X = rr_x3_scratch_of(ix);
    rr_launch_split_queries(d_q, planes, QN, A_BF16 ? RR_X3_ORDER_NATURAL : RR_X3_ORDER_PAIR64, st);
    if (rr_x3_stored_path(ix)) {
        const int slot = rr_scan_events_begin(ix, st);
    rr_scan_note(ix, 3, NQT, nq, A_BF16 ? 3 : 6);
        hipLaunchKernelGGL((rr_scan_mfma_x3<NQT, A_BF16, true>), grid, block, 0, st, mat, G, pl4, ix->d_sims,
                           ix->d_gmax, ix->d_smax, (const int32_t*)nullptr, 0);
        rr_scan_events_end(ix, slot, st);
        rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st);
        RR_HIP_TRY(hipGetLastError());
        return RR_OK;
    }
    const int slot = rr_scan_events_begin(ix, st);
    rr_scan_note(ix, 3, NQT, nq, A_BF16 ? 3 : 6);
    hipLaunchKernelGGL((rr_scan_mfma_x3<NQT, A_BF16, false>), grid, block, 0, st, mat, G, pl4, ix->d_sims,
                       ix->d_gmax, ix->d_smax, (const int32_t*)nullptr, 0);
    rr_scan_events_end(ix, slot, st);
    rr_launch_select_mtiles(ix, G, nq, pool, st);
    hipLaunchKernelGGL((rr_rescore_x3<A_BF16>), dim3(64, nq), dim3(256), 0, st, mat, G.n_rows, pl4, QN,
                       X.mtiles, X.count, X.fb, X.sc);
    rr_launch_select_rescored(ix, G, nq, pool, d_rows, d_scores, st);
    // Fallback for the queries that raised their flag: both launches return at once otherwise.
    hipLaunchKernelGGL((rr_scan_mfma_x3<NQT, A_BF16, true>), grid, block, 0, st, mat, G, pl4, ix->d_sims,
                       ix->d_gmax, ix->d_smax, (const int32_t*)X.fb, nq);
    rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st, X.fb);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

int rr_dense_chunk_x3(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                      float* d_scores, hipStream_t st) {
    const bool b = ix->dtype == RR_DTYPE_BF16;
    static int narrow = -1;
    if (narrow < 0) {
        const char* e = getenv("RR_X3_NARROW");       // diagnostic: 16x16x32 tiles for every batch size
        narrow = (e && e[0] == '1') ? 1 : 0;
    }
    if (nq > 16 && !narrow) return rr_dense_chunk_x3w(ix, d_q, nq, pool, d_rows, d_scores, st);
    if (nq <= 16)
        return b ? rr_dense_chunk_x3_t<1, true>(ix, d_q, nq, pool, d_rows, d_scores, st)
                 : rr_dense_chunk_x3_t<1, false>(ix, d_q, nq, pool, d_rows, d_scores, st);
    if (nq <= 32)
        return b ? rr_dense_chunk_x3_t<2, true>(ix, d_q, nq, pool, d_rows, d_scores, st)
                 : rr_dense_chunk_x3_t<2, false>(ix, d_q, nq, pool, d_rows, d_scores, st);
    return b ? rr_dense_chunk_x3_t<4, true>(ix, d_q, nq, pool, d_rows, d_scores, st)
             : rr_dense_chunk_x3_t<4, false>(ix, d_q, nq, pool, d_rows, d_scores, st);
}

