// rr_dense_flt.hip -- K1 batched scan, 5..128 queries per matrix read: bf16 FILTER scan + exact rescoring.
//
// The split-operand scans (rr_dense_x3*.hip) compute every one of the 10M x B scores to fp32
// accuracy on the matrix cores and are paced by that arithmetic (6 MFMA terms + a 48-op operand
// split per 32 rows x 16 dims), not by HBM.  But only ~pool rows per query survive, and the
// selection already rescans its candidates.  So the scan only has to be a FILTER with a known
// error bound:
//
//   s~(row) = sum_k bf16(a_k) * bf16(q_k)          one bf16 MFMA term, rounding to nearest even
//   |s~ - s| <= eps_q = 2^-7 (1 + margin) * max_row ||a|| * ||q||     (bf16 unit roundoff 2^-8 on each
//                       factor, Cauchy-Schwarz; bf16 storage: 2^-8, the row is exact)
//
//   tau~  = pool-th largest group maximum of s~   =>  >= pool rows have s >= tau~ - eps
//   every true top-pool row has s >= tau~ - eps, hence s~ >= tau~ - 2 eps
//   => the 16-row M-tiles whose s~ maximum reaches tau~ - 2 eps contain the exact top-pool.
//
// Those M-tiles (~pool + a few: the score density at the cut is ~1e4 rows per unit score at 10M
// rows) are rescored by rr_rescore_chain with THE PER-ROW FMAF CHAIN OF THE SINGLE-QUERY SCAN
// (rr_scan_f32 / rr_scan_bf16: 16 lanes per row, fixed DPP sum), rows below tau~ - eps are dropped
// and the rest ordered by (score desc, row asc).  The result is the exact top-pool of the
// per-row-chain scores: a query's answer is bitwise the same alone, in any batch, on any shard.
// A query whose candidate lists overflow (eps too loose for its score density, massive ties, tiny
// matrices) raises its flag and is served by the stored-score pass of rr_dense_x3w.hip instead.
//
// The scan itself is rr_scan_x3w's stream with one MFMA term: 32x32x16 tiles, 16-rows x 64-B load
// instructions turned into MFMA lanes by v_permlane16_swap, a 24-unit register ring refilled by
// halves in bursts, M-tile maxima stored one M-tile late behind the next burst, software pipeline
// pinned with sched_barrier.  Per 32-row K-step: NQ2 MFMAs, 4 lane swaps + 4 v_cvt_pk_bf16_f32;
// one query plane of 768 B in LDS, so 128 queries fit (98 KB): a quarter of the matrix reads of
// the 64-query kernels at batch 256.
#include "rr_x3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define RR_FLT_QSTRIDE 49   // 16-B units per query in LDS: 48 + 1 pad (eight consecutive queries, eight bank groups)

// One workgroup (64 lanes) per query slot: the bf16 plane (round to nearest even, memory order: both filter scans take
// their A operands as they sit in the rows) and the slot's error bound (see rr_flt_bounds).
struct rr_flt_bounds {
    float row_norm;      // >= max over rows ||a||
    float row_delta;     // >= max over rows ||a - bf16(a)||   (0 for a bf16 matrix)
};
__global__ __launch_bounds__(64) void rr_flt_prep_queries(const float* __restrict__ q, unsigned short* __restrict__ plane,
                                                          float* __restrict__ eps, rr_flt_bounds B) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    const float* src = q + (int64_t)slot * 384;
    float ss = 0.f, sr = 0.f, sd = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pos = lane + 64 * i;
        const float x = src[pos];
        const __bf16 r = (__bf16)x;                                // round to nearest even
        const float xr = (float)r, d = x - xr;
        ss = __builtin_fmaf(x, x, ss);
        sr = __builtin_fmaf(xr, xr, sr);
        sd = __builtin_fmaf(d, d, sd);
        plane[(int64_t)slot * 384 + pos] = __builtin_bit_cast(unsigned short, r);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        ss += __shfl_xor(ss, m, 64);
        sr += __shfl_xor(sr, m, 64);
        sd += __shfl_xor(sd, m, 64);
    }
    // s~ - s = sum (a~ - a) q~ + sum a (q~ - q): |.| <= ||a - a~|| ||q~|| + ||a|| ||q - q~||  (Cauchy-Schwarz, twice);
    // + 2^-14 ||a|| ||q|| for the fp32 accumulations of the scan and of the rescoring chain (384 terms each);
    // 1 % on top for the roundings of these norms themselves.  NaN / inf anywhere gives a NaN / inf bound: no filtering.
    if (lane == 0)
        eps[slot] = 1.01f * (B.row_delta * sqrtf(sr) + B.row_norm * sqrtf(sd) + 6.1035156e-5f * B.row_norm * sqrtf(ss));
}

// The same with the padding of the caller's queries in front (rr_pad_queries): slot < nq reads the caller's row (nq x dim,
// device memory or mapped pinned host memory), the rest is zeros; the padded fp32 row goes to `padded` as well.
__global__ __launch_bounds__(64) void rr_flt_pad_prep_queries(const float* __restrict__ src, int nq, int dim, float* __restrict__ padded,
                                                              unsigned short* __restrict__ plane, float* __restrict__ eps, rr_flt_bounds B) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f, sr = 0.f, sd = 0.f;
    // (dim 384, the reference's encoders: the row as 96 sixteen-byte loads -- the queries may sit in pinned host memory, where
    //  a wave instruction of 1 KiB crosses PCIe four times as well as one of 256 B; lane order of the sums is unchanged)
    __shared__ float row[384];
    const bool wide = dim == 384 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
    if (wide && slot < nq) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(src + (int64_t)slot * 384);
        for (int u = lane; u < 96; u += 64) *reinterpret_cast<f32x4*>(row + 4 * u) = s4[u];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pos = lane + 64 * i;
        const float x = wide ? (slot < nq ? row[pos] : 0.f) : ((slot < nq && pos < dim) ? src[(int64_t)slot * dim + pos] : 0.f);
        padded[(int64_t)slot * 384 + pos] = x;
        const __bf16 r = (__bf16)x;                                // round to nearest even
        const float xr = (float)r, d = x - xr;
        ss = __builtin_fmaf(x, x, ss);
        sr = __builtin_fmaf(xr, xr, sr);
        sd = __builtin_fmaf(d, d, sd);
        plane[(int64_t)slot * 384 + pos] = __builtin_bit_cast(unsigned short, r);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        ss += __shfl_xor(ss, m, 64);
        sr += __shfl_xor(sr, m, 64);
        sd += __shfl_xor(sd, m, 64);
    }
    if (lane == 0)        // (the bound of rr_flt_prep_queries, term for term)
        eps[slot] = 1.01f * (B.row_delta * sqrtf(sr) + B.row_norm * sqrtf(sd) + 6.1035156e-5f * B.row_norm * sqrtf(ss));
}

// max over rows of ||a|| and of ||a - bf16(a)||, as the bits of non-negative floats (atomicMax on uint).
// 16 lanes per row, 16-byte loads (four rows per wave instruction), 64 rows per wave.
template <bool A_BF16>
__global__ __launch_bounds__(256) void rr_row_norm_max(const void* __restrict__ mat, int64_t n_rows, int dim_pad,
                                                       unsigned int* __restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    const int units = dim_pad / (A_BF16 ? 8 : 4);            // 16-byte units per row
    float best = 0.f, bestd = 0.f;
    for (int it = 0; it < 16; ++it) {
        const int64_t row = row0 + 4 * it + grp;
        float ss = 0.f, sd = 0.f;
        if (row < n_rows) {
            const u32x4* p = static_cast<const u32x4*>(mat) + row * units;
            for (int u = sub; u < units; u += 16) {
                const u32x4 w = p[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (A_BF16) {
                        const float x0 = __uint_as_float(w[e] << 16), x1 = __uint_as_float(w[e] & 0xFFFF0000u);
                        ss = __builtin_fmaf(x0, x0, ss);
                        ss = __builtin_fmaf(x1, x1, ss);
                    } else {
                        const float x = __uint_as_float(w[e]);
                        const float d = x - (float)(__bf16)x;
                        ss = __builtin_fmaf(x, x, ss);
                        sd = __builtin_fmaf(d, d, sd);
                    }
                }
            }
        }
        ss = rr_row16_sum(ss);
        sd = rr_row16_sum(sd);
        best = fmaxf(best, (ss == ss) ? sqrtf(ss) : INFINITY);       // a NaN row: no finite bound
        bestd = fmaxf(bestd, (sd == sd) ? sqrtf(sd) : INFINITY);
    }
    best = rr_wave_max(best);
    bestd = rr_wave_max(bestd);
    if (lane == 0) {
        atomicMax(out, __float_as_uint(best));
        atomicMax(out + 1, __float_as_uint(bestd));
    }
}

// ------------------------------------------------------------------ the filter scan
// v_max_f32 / v_max3_f32 spelled out: fmaxf() brings a canonicalising v_max x, x, x per operand with it
// (three instructions per maximum); like fmaxf they return the other operand for a NaN.
__device__ __forceinline__ float rr_vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float rr_vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// rr_scan_flt below is the scan of FP32 rows (an fp32 index without its bf16 filter plane: RR_NO_SHADOW=1, or no room
// for the plane); a bf16 stream -- the plane, or a bf16 matrix -- is scanned by rr_scan_flt16 further down.
// RR_FLT_THREADS(NQ2): workgroup size.  Four query tiles want 270 registers for a B-fragment lead of two
// K-steps; RR_FLT_ONE_WAVE=1 builds that variant (one wave per SIMD, the whole register file, lead 2):
// measured equal to two waves per SIMD with lead 1 (scan(128 q) / scan(32 q) = 1.14 either way), so off.
#define RR_FLT_THREADS(NQ2) ((NQ2) >= 4 && RR_FLT_ONE_WAVE ? 256 : 512)
#ifndef RR_FLT_ONE_WAVE
#define RR_FLT_ONE_WAVE 0
#endif
// DBG != 0: timing-only ablations (rr_debug_scan_flt, tools/flt_ablate.py; wrong results): bit 0 no epilogue
// (accumulators kept alive), bit 1 no B-fragment reads, bit 2 no MFMAs, bit 3 no lane swaps / conversions, bit 4 no M-tile maxima stores.
template <int NQ2, int DBG = 0>
__global__ __launch_bounds__(RR_FLT_THREADS(NQ2), (RR_FLT_THREADS(NQ2) == 256 ? 1 : 2)) void rr_scan_flt(
    const u32x4* __restrict__ mat, rr_scan_geom G, const u32x4* __restrict__ plane,   // [32*NQ2][48] units
    float* __restrict__ gmax, uint32_t* __restrict__ smax, const float* __restrict__ eps, int nq,
    const float* __restrict__ sigma,      // [32*NQ2] store prefilter (null: every tile word is stored), see rr_flt_sample
    uint32_t* __restrict__ dummy) {       // [n_waves][32*NQ2] lines that absorb the skipped stores
    constexpr int THREADS = RR_FLT_THREADS(NQ2);
    constexpr int QN = 32 * NQ2;
    constexpr int ROWU = 96;                          // 16-byte units per fp32 matrix row
    constexpr int SEGS = 2;                           // ring segments (24 units per lane) per 32-row M-tile
    constexpr int STEPS = 12;                         // K-steps (16 dims) per ring segment
    __shared__ u32x4 qs[QN * RR_FLT_QSTRIDE];
    __shared__ float sg[QN];
    const int tid = threadIdx.x;
    for (int i = tid; i < QN * RR_X3_UNITS; i += THREADS)
        qs[(i / RR_X3_UNITS) * RR_FLT_QSTRIDE + (i % RR_X3_UNITS)] = plane[i];
    for (int i = tid; i < QN; i += THREADS) sg[i] = sigma ? sigma[i] : -INFINITY;
    __syncthreads();

    const int lane = tid & 63;
    const int c = lane & 31;                          // MFMA: A row / B and C column
    const int h = lane >> 5;                          //       k half (A, B); C rows 8g + 4h + i
    const int64_t wave = (int64_t)blockIdx.x * (THREADS / 64) + (tid >> 6);
    if (wave >= G.n_waves) return;
    const int64_t t0 = wave * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;
    const int64_t m0 = t0 * 2, m1 = t1 * 2;           // 32-row M-tiles of this wave

    const int lrow = lane & 15, lpc = lane >> 4;      // load order: row of the 16-row half, 16-B piece
    const u32x4* px;
    const u32x4* py;
    auto seg_ptrs = [&](int64_t seg) {                // segment = (M-tile, ring segment of its rows), linear
        int64_t mt = m0 + seg / SEGS;
        const int p = (int)(seg % SEGS);
        mt = mt < m1 ? mt : m1 - 1;                   // (past the end: redundant re-loads, never used)
        int64_t rx = mt * 32 + lrow, ry = rx + 16;
        rx = rx < G.n_rows ? rx : G.n_rows - 1;
        ry = ry < G.n_rows ? ry : G.n_rows - 1;
        px = mat + rx * ROWU + p * 48 + lpc;
        py = mat + ry * ROWU + p * 48 + lpc;
    };
#define RR_FLT_LOAD(dst, j) \
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(((j) & 1) ? py : px), "n"(64 * ((j) / 2)) : "memory")
    u32x4 a[24];
    seg_ptrs(0);
#pragma unroll
    for (int j = 0; j < 24; ++j) RR_FLT_LOAD(a[j], j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int qlane = c * RR_FLT_QSTRIDE + h;               // 16-byte unit index into qs: + 32 t * QSTRIDE + 2 * kk
    float gm[NQ2];                                    // running maximum of the current group (sub-run of tiles)
    uint32_t pend[NQ2];                               // packed maxima of the 32-row tile just finished, stored one tile late
    // Store prefilter: the word of (tile, 32-query group j) is only ever read if some query of the group can have a
    // candidate in the tile, i.e. if its tile maximum reaches that query's sigma (a lower bound, minus the filter
    // margin, of where its threshold will end up: rr_flt_sample).  ~90 % of the 128-byte lines fail that for all 32
    // queries; in place of their store the wave issues a load of a line of its own (cache-resident) -- the COUNT of
    // vector-memory operations stays what the hand-counted vmcnt waits assume, the bytes written back to HBM (what the
    // stores cost) drop with the lines skipped.
    uint32_t pend_keep = 0xFFFFFFFFu;                 // bit j: the pending word of group j is wanted (wave-uniform)
    uint32_t junk = 0u;                               // destination of the loads that stand in for skipped stores
    uint32_t* const my_dummy = dummy ? dummy + (size_t)wave * QN : nullptr;
    const float step = rr_flt_gap_step(eps, nq);      // resolution of the 8-row gaps (half the smallest eps of the launch)
    const float inv_step = step > 0.f ? 0.9999f / step : 0.f;    // (0.9999: the decoded bound never rounds below the maximum)
#pragma unroll
    for (int t = 0; t < NQ2; ++t) {
        gm[t] = -INFINITY;
        pend[t] = 0x0000FF80u;                        // -inf, gaps 0
    }
    auto read_q = [&](int t, int kk) { return __builtin_bit_cast(bf16x8, qs[qlane + 32 * t * RR_FLT_QSTRIDE + 2 * kk]); };
    // fp32 pair (lo, hi = dims 8h .. 8h+7 of the K-step) -> one bf16x8 operand, round to nearest even
    auto cvt_op = [&](int k, const u32x4& lo, const u32x4& hi, u32x4& out) {
        const float x0 = __uint_as_float(k < 2 ? lo[2 * k] : hi[2 * k - 4]);
        const float x1 = __uint_as_float(k < 2 ? lo[2 * k + 1] : hi[2 * k - 3]);
        const bf16x2 r = bf16x2{(__bf16)x0, (__bf16)x1};
        out[k] = __builtin_bit_cast(unsigned int, r);
    };
    // ---- software pipeline (see rr_dense_x3w.hip): during the NQ2 MFMAs of K-step s the vector issue
    // prepares K-step s + 1: 4 lane swaps + 4 packed conversions (fp32 matrix) and NQ2 ds_reads.
    constexpr int NV = 8;
    constexpr int VPS = (NV + NQ2 - 1) / NQ2;
    u32x4 lo, hi, nxt;             // the pair being prepared (MFMA lanes) and its bf16 operand
    bf16x8 af;                     // operand of the current K-step
    bf16x8 qf[2][NQ2];             // B fragments of this K-step and the next (each re-loaded in place for two steps on)
    auto valu_op = [&](int k, const u32x4& x, const u32x4& y) {
        if (k < 4) {
            const auto r = __builtin_amdgcn_permlane16_swap(x[k], y[k], false, false);
            lo[k] = r[0];
            hi[k] = r[1];
        } else if (k < 8) {
            cvt_op(k - 4, lo, hi, nxt);
        }
    };
    {   // prologue: K-step 0 of the first segment
#pragma unroll
        for (int k = 0; k < 8; ++k) valu_op(k, a[0], a[1]);
        af = __builtin_bit_cast(bf16x8, nxt);
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            qf[0][t] = read_q(t, 0);
            qf[1][t] = read_q(t, 1);
        }
    }

#pragma unroll 1
    for (int64_t mt = m0; mt < m1; ++mt) {
        f32x16 acc[NQ2];
#pragma unroll
        for (int t = 0; t < NQ2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int p = 0; p < SEGS; ++p) {
            seg_ptrs((mt - m0) * SEGS + p + 1);           // the bursts of this segment refill the ring for the next
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int cb = (p * STEPS + s) & 1;
                const int s2 = (s + 1) % STEPS;                           // the K-step being prepared
                constexpr int LEAD = (NQ2 >= 4 && THREADS == 512) ? 1 : 2;   // K-steps of B-fragment prefetch (4 tiles x 2 waves/SIMD: no registers for 2)
                const int kk3 = (12 * p + s + LEAD) % 24;                 // K-step of the row LEAD steps ahead
                const bool swap = true;                                   // every K-step starts a new pair of the ring
                const int np = s2;
                constexpr int REFILL_SLOT = 3 / VPS;                      // the slot that issues the last lane swap
                constexpr int STORE_STEP = 5;                             // the K-step after the first burst (np == 5)
                if (swap && (np == 6 || np == 0)) {
                    // ring waits and deferred maxima stores exactly as in rr_scan_x3w
                    // (younger than what the wait needs: the other half's burst, and for the segment's second wait
                    //  the NQ2 maxima stores issued during the K-step after the first burst, see below)
                    if (p == 0 && (np == 0 || s > STORE_STEP) && !(DBG & 16)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(12 + NQ2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 12; ++j) asm volatile("" : "+v"(a[(np / 6) * 12 + j]));
                }
#pragma unroll
                for (int j = 0; j < NQ2; ++j) {
                    if (DBG & 4) asm volatile("" :: "v"(af), "v"(qf[cb][j]));
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf[cb][j], acc[j], 0, 0, 0);
                    if (swap && !(DBG & 8)) {
#pragma unroll
                        for (int k = VPS * j; k < VPS * j + VPS; ++k) valu_op(k, a[2 * np], a[2 * np + 1]);
                    }
                    // B fragment of the K-step after next, into the register the MFMA above just consumed: two
                    // K-steps (2 NQ2 MFMA slots) of lead instead of one (s_waitcnt lgkmcnt sat in front of every MFMA)
                    if (DBG & 2) {
                    } else if (LEAD == 2) qf[cb][j] = read_q(j, kk3);
                    else qf[cb ^ 1][j] = read_q(j, kk3);
                    if (swap && (np == 5 || np == 11) && j == REFILL_SLOT) {
                        // the lane swaps of this half's last pair are issued: re-load the half
#pragma unroll
                        for (int u = 0; u < 12; u += 2) RR_FLT_LOAD(a[(np / 6) * 12 + u], (np / 6) * 12 + u);
#pragma unroll
                        for (int u = 1; u < 12; u += 2) RR_FLT_LOAD(a[(np / 6) * 12 + u], (np / 6) * 12 + u);
                    }
                    // The previous M-tile's maxima (NQ2 stores, always: the ring waits count them), one per MFMA slot
                    // of the K-step after the first burst.  First M-tile of the wave: nothing pending -- the stores
                    // go to its own slot, overwritten by this wave one M-tile later.
                    // What these stores cost (tools/flt_ablate.py, 128 queries, a slow box of the pool): 3.03 ms with
                    // them, 2.49 ms (6.2 TB/s) without -- everything else in the kernel is hidden behind the HBM
                    // stream.  The cost follows the BYTES (320 MB per scan, 2 % of the traffic, 20 % of the time:
                    // neither placement nor halving the instruction count moves it): isolated line write-backs
                    // into a saturated read stream.
                    if (p == 0 && s == STORE_STEP && !(DBG & 16)) {
                        const int64_t mprev = mt > m0 ? mt - 1 : mt;
                        if (my_dummy && !((pend_keep >> j) & 1u)) {
                            // skipped line: a LOAD of this wave's own (cache-resident) line keeps the count of vector-memory
                            // operations in flight that the ring waits assume; a store to such a line still went out to
                            // HBM (PMC WRITE_SIZE unchanged at 165 MB per launch: full-line writes are streamed through)
                            // (`junk` is one register that stays live over the whole scan -- "+v" here, touched again behind
                            //  the final vmcnt(0) -- so nothing else can be allocated to it while such a load is in flight)
                            asm volatile("global_load_dword %0, %1, off" : "+v"(junk) : "v"(my_dummy + lane) : "memory");
                        } else if (h == 0) {
                            reinterpret_cast<uint32_t*>(gmax)[mprev * QN + 32 * j + c] = pend[j];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                af = __builtin_bit_cast(bf16x8, nxt);
            }
        }
        if (DBG & 1) {
#pragma unroll
            for (int t = 0; t < NQ2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) asm volatile("" :: "v"(acc[t][e]));
            continue;
        }
        // lane (c, h), register 4g + i: row 8g + 4h + i of the M-tile, query 32t + c
        const int64_t rbase = mt * 32 + 4 * h;
        const bool full = mt * 32 + 32 <= G.n_rows;
        uint32_t keep_now = 0u;
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            // maxima of the four 8-row M-tiles (rows 8g .. 8g + 7 = registers 4g .. 4g + 3 of both k halves)
            float m8[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                if (!full) v = rr_x3_canon(v, rbase + 8 * g, G.n_rows);            // (fmaxf drops a NaN by itself)
                m8[g] = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            }
            // the other k half's rows (lane l ^ 32)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m8[g]), __float_as_uint(m8[g]), false, false);
                m8[g] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            }
            // 4 bytes per (32-row tile, query): the tile maximum as bf16 ROUNDED UP + for each 8-row M-tile how far
            // below it its own maximum sits, in units of `step`, ROUNDED DOWN to a 4-bit code (rr_flt_gap_code, rr_x3.h).
            // The filter only asks "can this M-tile hold a score >= threshold": the decoded value
            // max_up - steps(code) * step is an upper bound of the M-tile's maximum.
            const float m32 = fmaxf(fmaxf(m8[0], m8[1]), fmaxf(m8[2], m8[3]));
            gm[t] = fmaxf(gm[t], m32);
            keep_now |= (__ballot(m32 >= sg[32 * t + c]) != 0ull ? 1u : 0u) << t;
            const uint32_t b = __float_as_uint(m32);
            uint32_t word = (b >> 31) ? (b >> 16) : ((b + 0xFFFFu) >> 16);           // toward +inf; +-inf stay
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                word |= rr_flt_gap_code(m32, m8[g], inv_step) << (16 + 4 * g);
            }
            pend[t] = word;
        }
        pend_keep = keep_now;
        {   // group k of the wave = tiles [t0 + k Cg, t0 + (k + 1) Cg) of its run.  Its maximum is stored at once:
            // the next ring wait then covers a fresh store and stalls (~2 us), once per ~19 tiles.
            const int in_run = (int)((mt >> 1) - t0), cg = (int)G.tiles_per_group;
            if ((mt & 1) == 1 && ((in_run + 1) % cg == 0 || mt == m1 - 1)) {
                const int64_t group = wave * G.gpw + in_run / cg;
#pragma unroll
                for (int t = 0; t < NQ2; ++t) {
                    if (h == 0) smax[group * QN + 32 * t + c] = rr_f2key(gm[t]);
                    gm[t] = -INFINITY;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (redundant) loads
    asm volatile("" :: "v"(junk));
    if (h == 0) {
        // groups of this wave that hold no tile (a short last run): key 0 = "nothing here"
        const int cg = (int)G.tiles_per_group;
        for (int k = (int)((t1 - t0 + cg - 1) / cg); k < G.gpw; ++k)
#pragma unroll
            for (int t = 0; t < NQ2; ++t) smax[(wave * G.gpw + k) * QN + 32 * t + c] = 0u;
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            if (!my_dummy || ((pend_keep >> t) & 1u)) reinterpret_cast<uint32_t*>(gmax)[(m1 - 1) * QN + 32 * t + c] = pend[t];
        }
    }
}

// ------------------------------------------------------------------ the filter scan over a bf16 stream: 16x16x32 tiles
// The scan of a bf16 matrix / of the bf16 filter plane (rr_scan_flt above stays the scan of fp32 rows).  Same stream,
// same tile words, same bound; what differs is the matrix-core tiling.  In-kernel stamps on the 32x32x16 tiling of this stream (round 2's first) showed
// the K-loop, not HBM, pacing the two-set launch: a wave alone on its SIMD needed 45 cycles per 32-cycle MFMA, two
// waves together no less -- the four v_permlane16_swap per 64 bytes of a row (the 32x32x16 A operand wants lane (row c,
// k half h), the coalesced loads deliver (row l & 15, 16-byte piece l >> 4)) sit on the vector issue between the MFMAs.
// v_mfma_f32_16x16x32_bf16 takes its A operand exactly as the loads deliver it (16 rows x 32 dims: lane l = row l & 15,
// dims 8 (l >> 4) .. + 7), so a ring register IS an operand: no lane swaps, no staging registers.  Per 32-dim K-step
// and wave: 2 NQ2 B fragments (16 queries each, one conflict-free 1 KiB ds_read_b128: the planes sit fragment-major
// in LDS) x 2 row halves = 4 NQ2 MFMAs of 16 cycles -- the same matrix time and LDS traffic as before.
//   C layout (lane l): query l & 15 of the fragment, rows 4 (l >> 4) + i of the 16-row half.  The maxima of the four
// 8-row M-tiles come out of one v_permlane16_swap + one maximum per (row half, fragment pair), in the arrangement the
// code stage wants (lanes < 32: M-tiles 0 / 2, lanes >= 32: M-tiles 1 / 3 of query 32 j + (l & 31)).
//   Ring: 24 registers = one 32-row M-tile; the two registers of a K-step are its two A operands.  Two K-steps = one
// 128-byte line of each row are re-loaded as soon as the second K-step's MFMAs are issued and waited for ten K-steps
// later with a counted wait (the other five groups' 20 loads, and the NQ2 tile-word stores unless they fall inside
// the group's own K-steps, are younger): a group has most of an M-tile of time to arrive, and no burst of twelve
// loads holds the wave's issue while the matrix pipe idles.
template <int NQ2, int DBG = 0, bool DUAL = false>
__global__ __launch_bounds__(512, 2) void rr_scan_flt16(
    const u32x4* __restrict__ mat, rr_scan_geom G, const u32x4* __restrict__ plane,   // [32*NQ2][48] units, NATURAL k order
    float* __restrict__ gmax, uint32_t* __restrict__ smax, const float* __restrict__ eps, int nq,
    const float* __restrict__ sigma, uint32_t* __restrict__ dummy, int nq_b, int64_t gmax_set_stride,
    uint32_t* __restrict__ prog, uint32_t seq, int tune) {
    constexpr int THREADS = 512;
    constexpr int QN = 32 * NQ2;
    constexpr int NF = 2 * NQ2;                       // 16-query B fragments per K-step
    constexpr int KS = 12;                            // 32-dim K-steps per row
    constexpr int STORE_KS = 4;                       // the K-step whose first NQ2 slots carry the previous M-tile's words
    __shared__ u32x4 qs[QN * RR_X3_UNITS];            // [K-step][fragment][k quarter][query of the fragment]: a fragment = 1 KiB, lane l reads unit l
    __shared__ float sg[QN];
    int wg = blockIdx.x;
    int set = 0;
    if (DUAL) {                                       // workgroups b and b + 8 (same XCD): sets 0 and 1 of the same rows
        set = (wg >> 3) & 1;
        wg = ((wg >> 4) << 3) | (wg & 7);
        plane += (size_t)set * (RR_FLT_MAXQ * RR_X3_UNITS);
        eps += set * RR_FLT_MAXQ;
        if (sigma) sigma += set * RR_FLT_MAXQ;
        gmax += set * gmax_set_stride;
        smax += (size_t)set * RR_FLT_MAXQ * RR_MAX_SCAN_WAVES;
        nq = set ? nq_b : nq;
    }
    const int tid = threadIdx.x;
    const uint64_t dbg_entry = (DBG & 128) ? __builtin_amdgcn_s_memrealtime() : 0;      // (100 MHz)
    for (int i = tid; i < QN * RR_X3_UNITS; i += THREADS) {
        const int q = i / RR_X3_UNITS, unit = i % RR_X3_UNITS;          // unit = 4 * K-step + k quarter
        qs[(((unit >> 2) * NF + (q >> 4)) * 4 + (unit & 3)) * 16 + (q & 15)] = plane[i];
    }
    for (int i = tid; i < QN; i += THREADS) sg[i] = sigma ? sigma[i] : -INFINITY;
    __syncthreads();

    const int lane = tid & 63;
    const int c = lane & 31, h = lane >> 5;           // tile words: query 32 j + c; M-tiles h and 2 + h
    const int64_t wave = (int64_t)wg * (THREADS / 64) + (tid >> 6);
    if (wave >= G.n_waves) return;
    const int64_t t0 = wave * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;
    const int64_t m0 = t0 * 2, m1 = t1 * 2;           // 32-row M-tiles of this wave
    if ((DBG & 128) && (tune & 32) && (tid >> 6) >= 4) return;     // timing only (stamped harness kernels): one wave per SIMD

    const int lrow = lane & 15, lpc = lane >> 4;      // load = operand order: row of the 16-row half, 16-B piece (k quarter)
    const u32x4* px;
    const u32x4* py;
    auto seg_ptrs = [&](int64_t seg) {                // M-tile m0 + seg (past the end: redundant re-loads, never used)
        int64_t mt = m0 + seg;
        mt = mt < m1 ? mt : m1 - 1;
        if (DBG & 64) mt = m0 + (mt & 1);             // timing only: every wave re-reads its first two M-tiles (cache hits)
        int64_t rx = mt * 32 + lrow, ry = rx + 16;
        rx = rx < G.n_rows ? rx : G.n_rows - 1;
        ry = ry < G.n_rows ? ry : G.n_rows - 1;
        px = mat + rx * RR_X3_UNITS + lpc;
        py = mat + ry * RR_X3_UNITS + lpc;
    };
    u32x4 a[24];                                      // a[2 k] / a[2 k + 1]: rows 0-15 / 16-31, dims 32 k .. 32 k + 31
    seg_ptrs(0);
#pragma unroll
    for (int j = 0; j < 24; ++j) RR_FLT_LOAD(a[j], j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float gm[NQ2];                                    // running maximum of the current group (sub-run of tiles)
    uint32_t pend[NQ2];                               // packed maxima of the 32-row tile just finished, stored one tile late
    uint32_t pend_keep = 0xFFFFFFFFu;                 // bit j: the pending word of group j is wanted (wave-uniform), see rr_scan_flt
    uint32_t junk = 0u;                               // destination of the loads that stand in for skipped stores
    uint32_t* const my_dummy = dummy ? dummy + ((size_t)set * G.n_waves + wave) * QN : nullptr;
    uint32_t* const my_prog = DUAL && prog ? prog + (size_t)set * G.n_waves + wave : nullptr;
    const uint32_t* partner_prog = nullptr;
    bool coupled = false;
    if (DUAL && prog) {
        const uint64_t pa = reinterpret_cast<uint64_t>(prog + (size_t)(set ^ 1) * G.n_waves + wave);
        const uint32_t lo32 = __builtin_amdgcn_readfirstlane((uint32_t)pa), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32));
        partner_prog = reinterpret_cast<const uint32_t*>(((uint64_t)hi32 << 32) | lo32);
        coupled = true;
    }
    const uint32_t code_shift = 16u + 4u * (uint32_t)h;
    const float step = rr_flt_gap_step(eps, nq);
    const float inv_step = step > 0.f ? 0.9999f / step : 0.f;
#pragma unroll
    for (int t = 0; t < NQ2; ++t) {
        gm[t] = -INFINITY;
        pend[t] = 0x0000FF80u;                        // -inf, gaps 0
    }
    auto read_q = [&](int ks, int f) { return __builtin_bit_cast(bf16x8, qs[(ks * NF + f) * 64 + lane]); };
    bf16x8 qf[NF];                                    // B fragments of the K-step at hand; each re-loaded in place for the next
#pragma unroll
    for (int f = 0; f < NF; ++f) qf[f] = read_q(0, f);

    uint64_t dbg_turn = 0, ts_end = 0, dbg_kloop = 0, dbg_epi = 0;
    const uint64_t dbg_t0 = (DBG & 128) ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t dbg_r0 = (DBG & 128) ? __builtin_amdgcn_s_memrealtime() : 0;
#pragma unroll 1
    for (int64_t mt = m0; mt < m1; ++mt) {
        // The MFMAs are spelled out (accumulating in place, the first K-step with a literal zero C and an early-clobber
        // destination: it must not land on a source operand that dies there): left to the
        // register allocator the unrolled chain took a fresh destination per MFMA, 256 registers and 49-86 spills,
        // whose reloads (vmcnt(0)) sat in the K-loop.  The compiler still sees every operand (it places the lgkmcnt
        // waits for the B fragments); what it cannot see is the MFMA -> VALU distance: see the s_nop behind the K-loop.
        f32x4 acc[2][NF];
        uint64_t ts0 = 0;
        if (DBG & 128) {
            ts0 = __builtin_amdgcn_s_memtime();
            if (mt > m0) dbg_turn += ts0 - ts_end;
        }
        seg_ptrs(mt - m0 + 1);                        // this M-tile's re-loads fill the ring for the next
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks % 2 == 0 && !(DBG & 4)) {
                // the group (K-steps ks, ks + 1) was re-loaded ten K-steps ago; issued since: the other five groups (20
                // loads) and -- unless they sit in this group's own K-steps -- one M-tile's NQ2 tile-word stores
                if (ks / 2 == STORE_KS / 2 || (DBG & 16)) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(20 + NQ2) : "memory");
                asm volatile("" : "+v"(a[2 * ks]), "+v"(a[2 * ks + 1]), "+v"(a[2 * ks + 2]), "+v"(a[2 * ks + 3]));
            }
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, a[2 * ks]), a1 = __builtin_bit_cast(bf16x8, a[2 * ks + 1]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (ks == 0) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[0][f]) : "v"(a0), "v"(qf[f]));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[1][f]) : "v"(a1), "v"(qf[f]));
                } else {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[0][f]) : "v"(a0), "v"(qf[f]));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[1][f]) : "v"(a1), "v"(qf[f]));
                }
                if (!(DBG & 2)) qf[f] = read_q((ks + 1) % KS, f);     // in place: this fragment's turn comes again 2 NF MFMAs on
                if (ks % 2 == 1 && f == NF - 1 && !(DBG & 4)) {
                    // the MFMAs of the group's second K-step are issued: its four registers take the same 128 bytes of
                    // the next M-tile's rows (the two halves of a line back to back)
                    RR_FLT_LOAD(a[2 * ks - 2], 2 * ks - 2);
                    RR_FLT_LOAD(a[2 * ks], 2 * ks);
                    RR_FLT_LOAD(a[2 * ks - 1], 2 * ks - 1);
                    RR_FLT_LOAD(a[2 * ks + 1], 2 * ks + 1);
                }
                if (ks == STORE_KS && f < NQ2 && !(DBG & (16 | 4))) {
                    // the previous M-tile's words: NQ2 vector-memory operations, always (the ring waits count them);
                    // first M-tile of the wave: nothing pending, the stores go to its own slot
                    const int64_t mprev = mt > m0 ? mt - 1 : mt;
                    if (DUAL && f == 0 && my_prog) {
                        // ONE store instruction: group 0's tile words from the h == 0 lanes (if wanted) and this
                        // wave's progress word from lane 32
                        const bool want = !my_dummy || (pend_keep & 1u);
                        if ((h == 0 && want) || lane == 32) {
                            uint32_t* dst = lane == 32 ? my_prog : reinterpret_cast<uint32_t*>(gmax) + mprev * QN + c;
                            const int64_t in_run = mt - m0 + 1;
                            *dst = lane == 32 ? seq + (uint32_t)(in_run < 65535 ? in_run : 65535) : pend[0];
                        }
                    } else if (my_dummy && !((pend_keep >> f) & 1u)) {
                        // skipped line: a load of this wave's own line stands in (see rr_scan_flt)
                        asm volatile("global_load_dword %0, %1, off" : "+v"(junk) : "v"(my_dummy + lane) : "memory");
                    } else if (h == 0) {
                        reinterpret_cast<uint32_t*>(gmax)[mprev * QN + 32 * f + c] = pend[f];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the last MFMAs' results are read by vector instructions below: the wait states the compiler would count for
        // its own MFMAs (at most 18 for any XDL write -> VALU read on gfx950; these are 4-pass)
        // (the accumulators are operands of the wait itself: nothing that reads them can be scheduled in front of it)
        if constexpr (NF == 8)
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]),
                         "+v"(acc[0][5]), "+v"(acc[0][6]), "+v"(acc[0][7]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                         "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7]) :: "memory");
        else if constexpr (NF == 4)
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]),
                         "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]) :: "memory");
        else
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
        uint64_t ts1 = 0;
        if (DBG & 128) {
            ts1 = __builtin_amdgcn_s_memtime();
            dbg_kloop += ts1 - ts0;
        }
        // lane l, accumulator (r, f), register i: row 16 r + 4 (l >> 4) + i of the M-tile, query 16 f + (l & 15)
        if (mt * 32 + 32 > G.n_rows) {                // the matrix's last, short M-tile: rows past the end (and NaNs) -> -inf
            const int64_t rbase = mt * 32 + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[r][f] = rr_x3_canon(acc[r][f], rbase + 16 * r, G.n_rows);
        }
        uint32_t keep_now = 0u;
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            // maxima of rows 4 (l >> 4) .. + 3 per (row half, fragment); v_permlane16_swap(x, y) leaves x = {x.row0, y.row0,
            // x.row2, y.row2}, y = {x.row1, y.row1, x.row3, y.row3} (rows of 16 lanes), so max(x, y) of fragments 2 t and
            // 2 t + 1 of row half r holds, for query 32 t + (l & 31), the 8-row M-tile 2 r in lanes < 32 and 2 r + 1 above
            float uw[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f32x4 va = acc[r][2 * t], vb = acc[r][2 * t + 1];
                const float x = rr_vmax3(va.x, va.y, rr_vmax(va.z, va.w)), y = rr_vmax3(vb.x, vb.y, rr_vmax(vb.z, vb.w));
                const auto rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
                uw[r] = rr_vmax(__uint_as_float(rs[0]), __uint_as_float(rs[1]));
            }
            const float u = uw[0], w = uw[1];         // M-tiles h and 2 + h
            const float mh = rr_vmax(u, w);
            const auto rm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mh), __float_as_uint(mh), false, false);
            const float m32 = rr_vmax(__uint_as_float(rm[0]), __uint_as_float(rm[1]));      // the tile maximum, in both halves
            gm[t] = rr_vmax(gm[t], m32);
            const bool kept = __ballot(m32 >= sg[32 * t + c]) != 0ull;                      // (wave-uniform)
            keep_now |= (kept ? 1u : 0u) << t;
            if (my_dummy && !kept) continue;                                                 // its word is never stored: no codes
            // the word: see rr_scan_flt (bf16 tile maximum rounded up + four 4-bit gap codes rounded down)
            const uint32_t b = __float_as_uint(m32);
            uint32_t word = (b >> 31) ? (b >> 16) : ((b + 0xFFFFu) >> 16);
            const uint32_t cu = rr_flt_gap_code(m32, u, inv_step), cw = rr_flt_gap_code(m32, w, inv_step);
            const uint32_t mine = (cu | (cw << 8)) << code_shift;
            const auto rc = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
            word |= rc[0] | rc[1];
            pend[t] = word;
        }
        pend_keep = keep_now;
        {   // group k of the wave = tiles [t0 + k Cg, t0 + (k + 1) Cg) of its run; its maximum is stored at once
            const int in_run = (int)((mt >> 1) - t0), cg = (int)G.tiles_per_group;
            if ((mt & 1) == 1 && ((in_run + 1) % cg == 0 || mt == m1 - 1)) {
                const int64_t group = wave * G.gpw + in_run / cg;
#pragma unroll
                for (int t = 0; t < NQ2; ++t) {
                    if (h == 0) smax[group * QN + 32 * t + c] = rr_f2key(gm[t]);
                    gm[t] = -INFINITY;
                }
            }
        }
        if (DBG & 128) {
            ts_end = __builtin_amdgcn_s_memtime();
            dbg_epi += ts_end - ts1;
        }
        if (DUAL && coupled) {                        // the wave of the other set with the same rows: stay within L2's reach (see rr_scan_flt)
            const int64_t in_run = mt - m0 + 1;
            const uint32_t mine = seq + (uint32_t)(in_run < 65535 ? in_run : 65535);
            uint32_t theirs;
            int spins = 0;
            for (;;) {
                asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(theirs) : "s"(partner_prog) : "memory");
                if ((int32_t)(theirs - mine) >= 0) break;
                if (++spins >= 64) {
                    coupled = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (redundant) loads
    asm volatile("" :: "v"(junk));
    if ((DBG & 128) && prog && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(prog) + wave * 6;
        o[0] = dbg_kloop; o[1] = dbg_epi; o[2] = dbg_turn; o[3] = __builtin_amdgcn_s_memtime() - dbg_t0;
        o[4] = __builtin_amdgcn_s_memrealtime() - dbg_r0; o[5] = dbg_r0 - dbg_entry;
    }
    if (h == 0) {
        const int cg = (int)G.tiles_per_group;
        for (int k = (int)((t1 - t0 + cg - 1) / cg); k < G.gpw; ++k)
#pragma unroll
            for (int t = 0; t < NQ2; ++t) smax[(wave * G.gpw + k) * QN + 32 * t + c] = 0u;
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            if (!my_dummy || ((pend_keep >> t) & 1u)) reinterpret_cast<uint32_t*>(gmax)[(m1 - 1) * QN + 32 * t + c] = pend[t];
        }
    }
}

// ------------------------------------------------------------------ the two-set scan, query-stationary
// rr_scan_flt16 keeps the QUERIES in LDS and streams the rows through registers: every wave loads its own rows, 24
// fragment-shaped global_load_dwordx4 (16 rows x 64 B each) per 32-row M-tile and 128 queries, and the issue of those
// loads is serial with the MFMAs of the SIMD (~60 cycles each beside 3 072 cycles of MFMA: DESIGN.md section 8).  Here
// the roles are swapped for the 256-query launch:
//   * one workgroup per CU, FOUR waves (one per SIMD, 512 registers each); wave w owns queries 64 w .. 64 w + 63 of the
//     launch's 256 (waves 0, 1 = set 0; 2, 3 = set 1) and keeps their B fragments -- 4 fragments of 16 queries x 12
//     K-steps of 32 dims x 4 registers = 192 -- in ACCUMULATION registers for the whole launch: no LDS planes at all;
//   * the ROWS come through LDS, once per CU: a 32-row M-tile = 24 LDS-DMA pieces of 8 rows x 128 B -- whole lines --,
//     six per wave, three M-tiles ahead in a ring of four 24 KB images; all four waves read their A operands from that one
//     image (ds_read_b128 = 16 rows x 32 dims, one per four MFMAs, conflict-free: a piece is stored 16-byte-swizzled
//     through its SOURCE addresses);
//   * so the stream leaves HBM once for 256 queries without any pairing of workgroups.
// MFMA shape (round 3): v_mfma_f32_16x16x32_bf16.  The kernel is not paced by its cycles but by the clock the chip holds
// under this load (1.2 - 1.5 GHz in-kernel, profiles/r03_fltq_asm_ablations_10M.txt): the same MACs as 16x16x32 MFMAs run
// at a ~12 % higher clock than as 32x32x16 (same cycles per FLOP; MI355X_MICROARCH.md "DVFS give-back" (7)).
// Per M-tile and wave: 96 MFMAs, 24 ds_read_b128, 6 LDS-DMA pieces, one raw s_barrier, the epilogue of its 64 queries
// (two blocks of 32, as in rr_scan_flt16) and two tile-word stores.  Runs: one per workgroup (G.n_waves = number of
// runs), groups = 1/32 of a run.
// ASM: the steady-state M-tiles run through the hand-scheduled loop of rr_fltq_loop.inc (gen_fltq_loop.py): the same
// MFMAs in the same order and the epilogue arithmetic of `piece()` bit for bit, so tile words and group maxima are those
// of the C++ bodies -- which stay for the first five M-tiles of a run, the last few and the matrix's short last M-tile
// (RR_FLTQ_NOASM=1 runs them everywhere: A/B).
// ABL (debug harness only): timing ablation of the hand-scheduled loop (gen_fltq_loop.py --abl), wrong results
template <bool ASM = false, int ABL = 0>
__global__ __launch_bounds__(256, 1) void rr_scan_fltq(
    const u32x4* __restrict__ mat, rr_scan_geom G, const u32x4* __restrict__ plane,   // [256][48] units: set 0, then set 1
    float* __restrict__ gmax, uint32_t* __restrict__ smax, const float* __restrict__ eps, int nq_a, int nq_b,
    int64_t gmax_set_stride, unsigned long long* __restrict__ stamps = nullptr) {      // stamps (ABL != 0): [run][wave][4]
    constexpr int NB = 4;                             // ring of M-tile images
    constexpr int F = 2;                              // 32-query blocks (pairs of 16-query fragments) per wave
    constexpr int PW = 6;                             // LDS-DMA pieces per wave and M-tile
    constexpr int TILE_UNITS = 32 * RR_X3_UNITS;      // 16-byte units per image (24 KB)
    __shared__ u32x4 ring[NB * TILE_UNITS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int run = blockIdx.x;
    if (run >= G.n_waves) return;                     // (whole workgroup)
    const int set = w >> 1;
    const int qoff = 64 * (w & 1);                    // this wave's first query inside its set
    const int nq = set ? nq_b : nq_a;
    float* const gm_out = gmax + set * gmax_set_stride;
    uint32_t* const sm_out = smax + (size_t)set * RR_FLT_MAXQ * RR_MAX_SCAN_WAVES;
    const float* const eps_set = eps + set * RR_FLT_MAXQ;
    const int64_t t0 = (int64_t)run * G.tiles_per_wave;
    const int64_t t1 = t0 + G.tiles_per_wave < G.n_tiles ? t0 + G.tiles_per_wave : G.n_tiles;
    const int64_t m0 = t0 * 2, m1 = t1 * 2;           // 32-row M-tiles of this run

    // ---- B fragments (v_mfma_f32_16x16x32_bf16: 16 queries x 32 dims): query 16 n + (l & 15) of this wave, dims 32 ks +
    // 8 (l >> 4) .. + 7  ->  bq[ks][n], accumulation registers (12 x 4 x 4 = 192)
    u32x4 bq[12][4];
    {
        const u32x4* src = plane + ((size_t)(64 * w + (lane & 15)) * RR_X3_UNITS + (lane >> 4));
#pragma unroll
        for (int ks = 0; ks < 12; ++ks)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(bq[ks][n]) : "v"(src + (16 * n) * RR_X3_UNITS + 4 * ks) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 12; ++ks)
#pragma unroll
            for (int n = 0; n < 4; ++n) asm volatile("" : "+a"(bq[ks][n]));
    }

    // ---- LDS-DMA: wave w brings rows 8 w .. 8 w + 7 of an M-tile (row group w), piece j = bytes 128 j .. 128 j + 127 of
    // each row; lane l = (row r8 = l >> 3, slot l & 7) fetches the 16 bytes (slot ^ f) of its row's segment, f = (r8 >> 1)
    // | (w & 1) << 2, so that the linear image holds 16-byte piece p of row r8 of row group w in slot p ^ f: the A reads
    // below then touch every bank group once per ds_read_b128 lane group
    const int d_r8 = lane >> 3, d_slot = lane & 7;
    auto dma_tile = [&](int64_t mt, int buf, int j0 = 0, int j1 = PW) {      // pieces j0 .. j1 - 1 of the wave's six (default: all)
        mt = mt < m1 ? mt : m1 - 1;                   // (past the end: redundant, never read)
        int64_t row = mt * 32 + 8 * w + d_r8;
        row = row < G.n_rows ? row : G.n_rows - 1;
        const u32x4* src = mat + row * RR_X3_UNITS + (d_slot ^ (d_r8 >> 1) ^ ((w & 1) << 2));
        u32x4* dst = ring + buf * TILE_UNITS + (w * 6) * 64;
#pragma unroll
        for (int j = 0; j < PW; ++j)
            if (j >= j0 && j < j1)
                __builtin_amdgcn_global_load_lds(src + 8 * j, (__attribute__((address_space(3))) void*)(dst + j * 64), 16, 0, 0);
    };
    // ---- A operand reads (16 rows x 32 dims): lane (row R = 16 r + (l & 15), k quarter l >> 4) of K-step ks wants piece
    // P = 4 ks + (l >> 4) of its row:  unit = ((R >> 3) * 6 + (P >> 3)) * 64 + (R & 7) * 8 + ((P & 7) ^ f(R & 7, R >> 3))
    // (P & 7 = 4 (ks & 1) + k quarter: two per-lane addresses, by the parity of ks; P >> 3 = ks >> 1, the row half r and
    // the image go into the offset field.)  A ds_read_b128 is served in four groups of 16 lanes, each holding the 16 rows
    // of the half once with two k quarters that differ in bit 0 (MI355X_MICROARCH.md, LDS): for the eight rows of equal
    // parity, (r8 >> 1) | (row group & 1) << 2 XOR that bit is a bijection onto the eight slots -- conflict-free.
    const int aR = lane & 15, aQ = lane >> 4, ar8 = aR & 7;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)ring;
    uint32_t a_addr[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
        a_addr[m] = ring_lds + 16u * (uint32_t)(((aR >> 3) * 6) * 64 + ar8 * 8 + ((4 * m + aQ) ^ (ar8 >> 1) ^ ((aR >> 3) << 2)));
#define RR_FLTQ_READ_A(dst, tile_off, u) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a_addr[((u) >> 1) & 1] + (tile_off)), \
                 "n"(((u) & 1) * (16 * RR_X3_UNITS * 16) + 1024 * ((u) >> 2)) : "memory")
    const int c = lane & 31, h = lane >> 5;
    const uint32_t code_shift = 16u + 4u * (uint32_t)h;
    const float step = rr_flt_gap_step(eps_set, nq);
    const float inv_step = step > 0.f ? 0.9999f / step : 0.f;
    float gm[F];
#pragma unroll
    for (int t = 0; t < F; ++t) gm[t] = -INFINITY;
    // accumulators: [set P = M-tile parity][32-query block t]: sixteen registers = (row half r, fragment 2 t + nn) at
    // 8 r + 4 nn .. + 3.  C layout of one MFMA (lane l, register i): row 16 r + 4 (l >> 4) + i, query 16 n + (l & 15).
    f32x16 acc[2][F];

    // ---- the epilogue of one M-tile for this wave's 64 queries, block t = queries 32 t .. 32 t + 31, in pieces of at most
    // four vector instructions: placed behind the MFMAs of the NEXT M-tile (C++ bodies) / spread over its MFMA shadows
    // (generated loop), on the accumulator set that M-tile does not write.  `live` = there is a previous M-tile.
    float ep8[F][4], eu[F], ew[F], em32[F], egu[F], egw[F], etmp0[F], etmp1[F];
    uint32_t ecu[F], ecw[F], eword[F];
    auto piece = [&](int P, int t, int k, int64_t tile, bool live) {      // (P, t, k: constants once unrolled)
        if (k < 4) {                                   // lane-local maxima: rows 4 (l >> 4) .. + 3 of (row half k >> 1, fragment 2 t + (k & 1))
            ep8[t][k] = rr_vmax3(acc[P][t][4 * k], acc[P][t][4 * k + 1], rr_vmax(acc[P][t][4 * k + 2], acc[P][t][4 * k + 3]));
        } else if (k == 4 || k == 5) {
            // v_permlane16_swap(x, y) = {x.row0, y.row0, x.row2, y.row2}, {x.row1, y.row1, x.row3, y.row3} (rows of 16 lanes): the
            // maximum of the two holds, for query 32 t + (l & 31), the 8-row M-tile 2 r in lanes < 32 and 2 r + 1 above
            const int g = k == 4 ? 0 : 2;
            const auto rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(ep8[t][g]), __float_as_uint(ep8[t][g + 1]), false, false);
            const float v = rr_vmax(__uint_as_float(rs[0]), __uint_as_float(rs[1]));
            if (k == 4) eu[t] = v;                     // lanes < 32: M-tile 0, lanes >= 32: M-tile 1
            else ew[t] = v;                            // M-tiles 2 / 3
        } else if (k == 6) {
            const float mh = rr_vmax(eu[t], ew[t]);
            const auto rm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mh), __float_as_uint(mh), false, false);
            etmp0[t] = __uint_as_float(rm[0]);
            etmp1[t] = __uint_as_float(rm[1]);
        } else if (k == 7) {
            em32[t] = rr_vmax(etmp0[t], etmp1[t]);     // the tile maximum, in both halves
            gm[t] = live ? rr_vmax(gm[t], em32[t]) : gm[t];
        } else if (k == 8) {
            const uint32_t b = __float_as_uint(em32[t]);
            eword[t] = (b >> 31) ? (b >> 16) : ((b + 0xFFFFu) >> 16);
        } else if (k == 9) {                           // the two gap codes of this lane (rr_flt_gap_code, spelled out in four pieces)
            egu[t] = __builtin_fmaf(em32[t] - eu[t], inv_step, 8.0f);
        } else if (k == 10) {
            egw[t] = __builtin_fmaf(em32[t] - ew[t], inv_step, 8.0f);
        } else if (k == 11) {
            asm("v_min_f32 %0, 0x41ffeb85, %0" : "+v"(egu[t]));
            ecu[t] = (__float_as_uint(egu[t]) >> 20) & 15u;
        } else if (k == 12) {
            asm("v_min_f32 %0, 0x41ffeb85, %0" : "+v"(egw[t]));
            ecw[t] = (__float_as_uint(egw[t]) >> 20) & 15u;
        } else if (k == 13) {
            const uint32_t mine = (ecu[t] | (ecw[t] << 8)) << code_shift;
            const auto rc = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
            ecu[t] = rc[0];
            ecw[t] = rc[1];
        } else if (k == 14) {
            eword[t] |= ecu[t] | ecw[t];
        } else if (k == 15) {
            if (h == 0 && live) reinterpret_cast<uint32_t*>(gm_out)[tile * RR_FLT_MAXQ + qoff + 32 * t + c] = eword[t];   // ONE store
        }
    };
    auto finish_tile = [&](int64_t pm) {              // group bookkeeping of the M-tile whose epilogue has just finished
        const int in_run = (int)((pm >> 1) - t0), cg = (int)G.tiles_per_group;
        if ((pm & 1) == 1 && ((in_run + 1) % cg == 0 || pm == m1 - 1)) {
            const int64_t group = (int64_t)run * G.gpw + in_run / cg;
#pragma unroll
            for (int t = 0; t < F; ++t) {
                if (h == 0) sm_out[group * RR_FLT_MAXQ + qoff + 32 * t + c] = rr_f2key(gm[t]);
                gm[t] = -INFINITY;
            }
        }
    };
    auto canon_set = [&](int P, int64_t tile) {       // rows past the end (and NaNs) of the matrix's last, short M-tile -> -inf
        const int64_t rbase = tile * 32 + 4 * (lane >> 4);
#pragma unroll
        for (int t = 0; t < F; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) {             // registers 4 k ..: row half k >> 1
                f32x4 v = {acc[P][t][4 * k], acc[P][t][4 * k + 1], acc[P][t][4 * k + 2], acc[P][t][4 * k + 3]};
                v = rr_x3_canon(v, rbase + 16 * (k >> 1), G.n_rows);
                acc[P][t][4 * k] = v.x; acc[P][t][4 * k + 1] = v.y; acc[P][t][4 * k + 2] = v.z; acc[P][t][4 * k + 3] = v.w;
            }
    };

#pragma unroll
    for (int t = 0; t < F; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[1][t][e] = 0.f;
    dma_tile(m0, 0);
    dma_tile(m0 + 1, 1);
    dma_tile(m0 + 2, 2);
    const uint64_t dbg_t0 = ABL != 0 ? __builtin_amdgcn_s_memtime() : 0;
    constexpr int AD = 4;                             // A operands requested this many operands (4 MFMAs each) ahead
    bf16x8 a[AD + 1];
    auto body = [&](auto PC, int64_t mt) {
        constexpr int P = decltype(PC)::value;        // accumulator set of THIS M-tile; 1 - P: the previous one's
        const int it = (int)(mt - m0), buf = it & (NB - 1);
        const bool have_prev = it > 0;
        // this wave's six pieces of M-tile mt have landed once at most the two younger M-tiles' pieces (12) and the four
        // word stores issued with them are outstanding; then all four waves' pieces have, behind the barrier -- which also
        // says that every wave is done reading M-tile mt - 1
        if (it < 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PW + F)) : "memory");
        __builtin_amdgcn_s_barrier();
        if (have_prev && (mt - 1) * 32 + 32 > G.n_rows) canon_set(1 - P, mt - 1);
        const uint32_t toff = (uint32_t)buf * (TILE_UNITS * 16);
        f32x4 c4[2][4];                               // [row half][fragment] of this M-tile
#pragma unroll
        for (int i = 0; i < AD; ++i) RR_FLTQ_READ_A(a[i], toff, i);
#pragma unroll
        for (int u = 0; u < 24; ++u) {                // operand u = 2 ks + r: rows 16 r .. + 15, dims 32 ks .. + 31; four MFMAs
            if (u + AD < 24) RR_FLTQ_READ_A(a[(u + AD) % (AD + 1)], toff, u + AD);
            // reads return in order: with the (up to AD) younger ones outstanding, this operand's is in
            if (u + AD < 24) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(AD) : "memory");
            else if (23 - u == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else if (23 - u == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else if (23 - u == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(a[u % (AD + 1)]));
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                constexpr int PB = 48;                  // MFMAs per 32-query block of the PREVIOUS M-tile's epilogue (16 pieces used)
                const int idx = 4 * u + n;              // MFMA of this M-tile: 0 .. 95
                if (u < 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c4[u & 1][n]) : "v"(a[u % (AD + 1)]), "a"(bq[u >> 1][n]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[u & 1][n]) : "v"(a[u % (AD + 1)]), "a"(bq[u >> 1][n]));
                __builtin_amdgcn_sched_barrier(0);      // the piece stands BEHIND its MFMA (in front of it, it would only delay it)
                if (idx % PB < 16) piece(1 - P, idx / PB, idx % PB, mt - 1, have_prev);
                // the wave's LDS-DMA pieces of the M-tile three ahead (its image held M-tile mt - 1: read out, see the barrier)
                if (idx % PB >= 16 && idx % PB < 19) dma_tile(mt + 3, (it + 3) & (NB - 1), 3 * (idx / PB) + idx % PB - 16, 3 * (idx / PB) + idx % PB - 15);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the last MFMAs' results are copied by vector instructions below: the wait states the compiler would count for its own
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c4[0][0]), "+v"(c4[0][1]), "+v"(c4[0][2]), "+v"(c4[0][3]), "+v"(c4[1][0]), "+v"(c4[1][1]),
                     "+v"(c4[1][2]), "+v"(c4[1][3]) :: "memory");
#pragma unroll
        for (int t = 0; t < F; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[P][t][4 * k + i] = c4[k >> 1][2 * t + (k & 1)][i];
        if (have_prev) finish_tile(mt - 1);
    };
    if constexpr (ASM) {
        // ---- wave-uniform operands of the hand-scheduled loop (SGPRs: readfirstlane, the compiler sees tid >> 6 per lane)
        const int wu = __builtin_amdgcn_readfirstlane(w);
        const uint32_t lds_w = __builtin_amdgcn_readfirstlane(ring_lds) + (uint32_t)wu * (PW * 1024);
        const uint32_t dma_voff = (uint32_t)d_r8 * 768u + 16u * (uint32_t)(d_slot ^ (d_r8 >> 1) ^ ((wu & 1) << 2));
        const uint32_t st_voff = h == 0 ? (uint32_t)(qoff + c) * 4u : 0xFFFFF000u;      // (upper half: out of range, dropped)
        const uint32_t inv_step_s = __builtin_amdgcn_readfirstlane(__float_as_uint(inv_step));
        uint32_t al[2], ah[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            al[m] = a_addr[m];
            ah[m] = a_addr[m] + 2u * (TILE_UNITS * 16);
        }
        const uint64_t st_base = (uint64_t)(uintptr_t)gm_out;
        const uint64_t st_bytes = (uint64_t)((G.n_rows + 31) / 32) * (RR_FLT_MAXQ * 4);
        u32x4 st_rsrc;
        st_rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)st_base);
        st_rsrc.y = __builtin_amdgcn_readfirstlane((uint32_t)(st_base >> 32) & 0xFFFFu);
        st_rsrc.z = (uint32_t)(st_bytes < 0xFFFFE000ull ? st_bytes : 0xFFFFE000ull);
        st_rsrc.w = 0x00020000u;
        const int64_t T = m1 - m0;                                      // M-tiles of this run
        const int64_t full = G.n_rows / 32 - m0 < T ? G.n_rows / 32 - m0 : T;   // M-tiles [0, full) are whole
        const int64_t cg2 = 2 * G.tiles_per_group;                     // M-tiles per selection group (a multiple of 4: rr_fltq_geom)
        int64_t it = 0;
        while (it < T) {
            int64_t n_it = 0;
            if (it >= 5 && (it & 3) == 1 && (cg2 & 3) == 0) {
                // bodies it .. it + 4 n - 1: whole M-tiles; only the LAST one (it = 0 mod 4) may be a group's flush body
                const int64_t flush = (it + cg2 - 1) / cg2 * cg2;       // next body whose end closes a group
                int64_t end = flush + 1 < full ? flush + 1 : full;
                if (end > T) end = T;
                n_it = (end - it) / 4;
            }
            if (n_it > 0) {
                // LDS-DMA resource: rows of this wave's row group of M-tile it + 3; num_records = bytes to the matrix's end
                const int64_t first_row = (m0 + it + 3) * 32 + 8 * wu;
                const uint64_t ld_base = (uint64_t)(uintptr_t)mat + (uint64_t)first_row * 768u;
                const int64_t left = (G.n_rows - first_row) * 768;
                u32x4 ld_rsrc;
                ld_rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)ld_base);
                ld_rsrc.y = __builtin_amdgcn_readfirstlane((uint32_t)(ld_base >> 32) & 0xFFFFu);
                ld_rsrc.z = __builtin_amdgcn_readfirstlane((uint32_t)(left <= 0 ? 0 : left < 0x7FFFFFFFll ? left : 0x7FFFFFFFll));
                ld_rsrc.w = 0x00020000u;
                uint32_t st_soff = __builtin_amdgcn_readfirstlane((uint32_t)((m0 + it - 2) * (RR_FLT_MAXQ * 4)));
                uint32_t loops = __builtin_amdgcn_readfirstlane((uint32_t)n_it);
                // (the compiler moves the accumulators into / out of the loop's pinned registers with plain copies and does not
                //  know that MFMAs -- inline asm to it -- wrote them: the matrix pipe must have drained on both sides; the
                //  loop ends with its own s_nops, in front of the copies out)
                asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
#ifdef RR_DEBUG_HARNESS
#include "rr_fltq_loop_abl.inc"
                else
#endif
#include "rr_fltq_loop.inc"
                it += 4 * n_it;
                finish_tile(m0 + it - 2);                               // (flushes only where that M-tile closes a group)
            } else {
                if (it & 1) body(std::integral_constant<int, 1>{}, m0 + it);
                else body(std::integral_constant<int, 0>{}, m0 + it);
                ++it;
            }
        }
    } else {
    for (int64_t mt = m0; mt < m1; mt += 2) {
        body(std::integral_constant<int, 0>{}, mt);
        body(std::integral_constant<int, 1>{}, mt + 1);
    }
    }
    // tail: the last M-tile (set 1) has its epilogue to run
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[1][0]), "+v"(acc[1][1]) :: "memory");
    if ((m1 - 1) * 32 + 32 > G.n_rows) canon_set(1, m1 - 1);
#pragma unroll
    for (int t = 0; t < F; ++t)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) piece(1, t, kk, m1 - 1, true);
    finish_tile(m1 - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (redundant) pieces
    if (ABL != 0 && stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)run * 4 + w) * 4;
        o[0] = 0; o[1] = 0; o[2] = 0; o[3] = __builtin_amdgcn_s_memtime() - dbg_t0;
    }
    if (h == 0) {
        const int cg = (int)G.tiles_per_group;
        for (int k = (int)((t1 - t0 + cg - 1) / cg); k < G.gpw; ++k)
#pragma unroll
            for (int t = 0; t < F; ++t) sm_out[((int64_t)run * G.gpw + k) * RR_FLT_MAXQ + qoff + 32 * t + c] = 0u;
    }
}
#undef RR_FLTQ_READ_A

// ------------------------------------------------------------------ store prefilter: sampled thresholds
// Every `stride`-th 32-row tile of the bf16 matrix / plane is scored against the launch's query planes (the same bf16
// products as the scan, natural A loads: one wave per sampled tile, 24 K-steps x NQ2 MFMAs) and its maximum per query
// kept: samp[tile][query].  The planes are in memory order (what rr_scan_flt16 wants), so lane half h of the
// 16-dim K-step s multiplies the row's dims 16 s + 8 h .. + 7.  (The sums run in another order than the scan's: the
// difference sits inside the 2^-14 term of the bound, and sigma carries a 2.05 eps margin.)
template <int NQ2>
__global__ __launch_bounds__(512, 1) void rr_flt_sample(const u32x4* __restrict__ mat, const u32x4* __restrict__ plane,
                                                        int stride, int n_samp, float* __restrict__ samp) {
    constexpr int QN = 32 * NQ2;
    constexpr int TQ = NQ2 >= 2 ? NQ2 / 2 : 1;          // query tiles per wave: a sampled tile is shared by NQ2 / TQ waves,
    constexpr int SPLIT = NQ2 / TQ;                     // so eight waves (two per SIMD) fit the register file and overlap
    __shared__ u32x4 qs[QN * RR_FLT_QSTRIDE];
    const int tid = threadIdx.x;
    for (int i = tid; i < QN * RR_X3_UNITS; i += 512)
        qs[(i / RR_X3_UNITS) * RR_FLT_QSTRIDE + (i % RR_X3_UNITS)] = plane[i];
    __syncthreads();
    const int lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 8 + (tid >> 6);         // global wave
    const int part = gw % SPLIT, t0 = part * TQ;        // this wave's query tiles t0 .. t0 + TQ - 1
    const int total = gridDim.x * 8 / SPLIT;
    for (int ti = gw / SPLIT; ti < n_samp; ti += total) {
        const int64_t tile = (int64_t)ti * stride + stride / 2;           // a full tile: the host keeps the last one out
        const u32x4* p = mat + (tile * 32 + c) * 48;
        u32x4 a[24];
#pragma unroll
        for (int s = 0; s < 24; ++s) a[s] = p[2 * s + h];            // dims 16 s + 8 h .. + 7: the planes are in memory order
        f32x16 acc[TQ];
#pragma unroll
        for (int t = 0; t < TQ; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 24; ++s)
#pragma unroll
            for (int t = 0; t < TQ; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s]),
                                                                 __builtin_bit_cast(bf16x8, qs[(32 * (t0 + t) + c) * RR_FLT_QSTRIDE + 2 * s + h]),
                                                                 acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            float m = acc[t][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) m = fmaxf(m, acc[t][e]);
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            if (h == 0) samp[(int64_t)ti * QN + 32 * (t0 + t) + c] = m;
        }
    }
}

// sigma[q] = (a lower bound of the m-th largest sampled tile maximum of query q) - 2.05 eps[q]; slots past nq never
// keep a line (+inf); a query without a finite bound keeps everything (-inf).  One 256-thread workgroup per query
// slot: every thread takes the maximum of its strided share of the samples (256 values, each a sampled tile maximum),
// and the m-th largest of those 256 -- found by counting, no bisection -- is at most the m-th largest sample (m
// distinct samples reach it) and sits close to it (m << 256: the top samples rarely share a thread).
// `force` (timing experiments only, RR_FLT_SIGMA_FORCE): +1 = +inf for every query (no line is kept), -1 = -inf.
__global__ __launch_bounds__(256) void rr_flt_sigma(const float* __restrict__ samp, int n_samp, int qn, int nq, int m,
                                                    const float* __restrict__ eps, float* __restrict__ sigma, int force) {
    __shared__ uint32_t tmax[256];
    const int tid = threadIdx.x, q = blockIdx.x;
    if (q >= nq || force > 0) { if (tid == 0) sigma[q] = INFINITY; return; }
    const float e = eps[q];
    if (n_samp < 256 || m > 128 || force < 0 || !(e >= 0.f && e < 3.0e38f)) { if (tid == 0) sigma[q] = -INFINITY; return; }
    float best = -INFINITY;
    for (int i = tid; i < n_samp; i += 256) best = fmaxf(best, samp[(int64_t)i * qn + q]);    // (fmaxf drops NaN)
    const uint32_t mine = rr_f2key(best);
    tmax[tid] = mine;
    __syncthreads();
    int greater = 0, geq = 0;
    for (int i = 0; i < 256; ++i) {
        const uint32_t o = tmax[i];
        greater += o > mine ? 1 : 0;
        geq += o >= mine ? 1 : 0;
    }
    if (greater < m && m <= geq) sigma[q] = rr_key2f(mine) - 2.05f * e;      // (threads with equal keys write the same value)
}

// ------------------------------------------------------------------ exact rescoring
// The listed 8-row M-tiles of a query, rescored with the single-query scans' per-row arithmetic -- lane j of a 16-lane row
// takes 16-byte units j, j + 16, ... of the row, one fmaf chain over its 24 elements in ascending order, rr_row16_sum over
// the row (rr_scan_f32<6,1> / rr_scan_bf16<1>).  sc[query][slot][8].
//
// The kernel is a gather of ~500 scattered M-tiles per query (768 B per plane row): what it costs is how many bytes a
// CU keeps in flight.  A wave therefore takes U M-tiles per pass and issues ALL their row loads before it looks at the
// first (U x 2 x 3 loads of 16 B per lane); the query sits in LDS (no registers held for it).  On a stream masked to a
// few CUs (the pipelined K1: this runs beside the next batch's scan) that is the difference between ~30 and ~90 GB/s
// per CU.
//
// fp32 rows with a bf16 filter plane: a row is first scored on its plane row -- half the bytes -- as sum a~_k q_k, which
// is within eps of its exact score (|sum (a~ - a) q| <= ||a - a~|| ||q||, plus the summation roundings the bound's 2^-14
// term covers); the exact chain over the fp32 row runs only where that can reach the row cut tau (= key of tau~ - 1.02
// eps): one row in eight of an opened M-tile, typically.  The others get -inf: they are below the cut whatever their
// exact score is.  The rows that need the exact chain are taken per 16-lane group from a bit mask, one row per group and
// round, so a round serves up to four rows of four different M-tiles.
__device__ __forceinline__ float rr_bf16x8_chain(const u32x4 x, const f32x4 q0, const f32x4 q1, float acc) {
    acc = __builtin_fmaf(__uint_as_float(x.x << 16), q0.x, acc);
    acc = __builtin_fmaf(__uint_as_float(x.x & 0xFFFF0000u), q0.y, acc);
    acc = __builtin_fmaf(__uint_as_float(x.y << 16), q0.z, acc);
    acc = __builtin_fmaf(__uint_as_float(x.y & 0xFFFF0000u), q0.w, acc);
    acc = __builtin_fmaf(__uint_as_float(x.z << 16), q1.x, acc);
    acc = __builtin_fmaf(__uint_as_float(x.z & 0xFFFF0000u), q1.y, acc);
    acc = __builtin_fmaf(__uint_as_float(x.w << 16), q1.z, acc);
    acc = __builtin_fmaf(__uint_as_float(x.w & 0xFFFF0000u), q1.w, acc);
    return acc;
}

#define RR_RESCORE_U 4            // M-tiles per wave and pass (bf16 rows / plane rows); fp32 rows without a plane: half
template <bool A_BF16>
__global__ __launch_bounds__(256, 3) void rr_rescore_chain(
    const void* __restrict__ mat, int64_t n_rows, const float* __restrict__ queries,   // [nq][384] fp32, padded
    const uint32_t* __restrict__ mtiles, const int32_t* __restrict__ count, const int32_t* __restrict__ fb,
    float* __restrict__ sc, const u32x4* __restrict__ plane_rows,                      // (null: no plane)
    const uint32_t* __restrict__ tau, const float* __restrict__ eps) {
    __shared__ f32x4 qs[96];
    const int q = blockIdx.y;
    if (fb[q]) return;
    const int n = count[q];
    const bool planed = A_BF16 || plane_rows != nullptr;           // rows of 48 16-byte units go through the U-deep pass
    const int per_wave = planed ? RR_RESCORE_U : RR_RESCORE_U / 2;
    if ((int)blockIdx.x * 4 * per_wave >= n) return;               // (the whole workgroup: before the barrier)
    if (threadIdx.x < 96) qs[threadIdx.x] = reinterpret_cast<const f32x4*>(queries + (int64_t)q * 384)[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, grp = lane >> 4;
    mtiles += (int64_t)q * RR_X3_MCAP;
    sc += (int64_t)q * RR_X3_MCAP * 8;
    const int64_t last_row = n_rows - 1;
    if (planed) {
        constexpr int U = RR_RESCORE_U;
        const u32x4* rows16 = A_BF16 ? static_cast<const u32x4*>(mat) : plane_rows;
        float pre_thr = A_BF16 ? -INFINITY : rr_key2f(tau[q]) - 1.01f * eps[q];
        if (!(pre_thr == pre_thr)) pre_thr = -INFINITY;   // (a key that is not a score, a NaN bound: every row takes the exact chain)
        for (int s0 = ((int)blockIdx.x * 4 + wave) * U; s0 < n; s0 += (int)gridDim.x * 4 * U) {
            uint32_t m8[U];
#pragma unroll
            for (int u = 0; u < U; ++u) m8[u] = mtiles[s0 + u < n ? s0 + u : n - 1];      // (a repeated last tile: its stores are masked)
            u32x4 x[U][2][3];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int it = 0; it < 2; ++it) {           // four rows per step: lane (sub, grp) -> row 4 it + grp
                    int64_t row = (int64_t)m8[u] * 8 + 4 * it + grp;
                    row = row < last_row ? row : last_row;
                    const u32x4* pp = rows16 + row * 48 + sub;
#pragma unroll
                    for (int i = 0; i < 3; ++i) x[u][it][i] = pp[16 * i];
                }
            uint32_t need = 0;                             // bit 2 u + it: that row of this 16-lane group takes the exact chain
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    float est = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i)            // lane `sub`: units sub + 16 i = dims 8 (sub + 16 i) .. + 7
                        est = rr_bf16x8_chain(x[u][it][i], qs[2 * (sub + 16 * i)], qs[2 * (sub + 16 * i) + 1], est);
                    est = rr_row16_sum(est);
                    const int64_t row = (int64_t)m8[u] * 8 + 4 * it + grp;
                    const bool live = s0 + u < n;
                    if (A_BF16) {                          // a bf16 index: that WAS the row's chain
                        if (sub == 0 && live) sc[(int64_t)(s0 + u) * 8 + 4 * it + grp] = (row < n_rows && est == est) ? est : -INFINITY;
                    } else {
                        const bool exact = live && row < n_rows && (est >= pre_thr || !(est == est));   // (the 16 lanes of a row agree; NaN: the exact chain decides)
                        need |= exact ? 1u << (2 * u + it) : 0u;
                        if (sub == 0 && live && !exact) sc[(int64_t)(s0 + u) * 8 + 4 * it + grp] = -INFINITY;
                    }
                }
            if (!A_BF16) {
                while (__any(need != 0u)) {
                    if (need != 0u) {
                        const int j = __ffs(need) - 1;
                        need &= need - 1u;
                        const int u = j >> 1, it = j & 1;
                        uint32_t m = m8[0];
#pragma unroll
                        for (int v = 1; v < U; ++v) m = u == v ? m8[v] : m;
                        const int64_t row = (int64_t)m * 8 + 4 * it + grp;
                        const f32x4* p = static_cast<const f32x4*>(mat) + row * 96 + sub;
                        f32x4 y[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) y[i] = p[16 * i];
                        float acc = 0.f;
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            const f32x4 qq = qs[16 * i + sub];
                            acc = __builtin_fmaf(y[i].x, qq.x, acc);
                            acc = __builtin_fmaf(y[i].y, qq.y, acc);
                            acc = __builtin_fmaf(y[i].z, qq.z, acc);
                            acc = __builtin_fmaf(y[i].w, qq.w, acc);
                        }
                        acc = rr_row16_sum(acc);
                        if (sub == 0) sc[(int64_t)(s0 + u) * 8 + 4 * it + grp] = acc == acc ? acc : -INFINITY;   // NaN scores rank last
                    }
                }
            }
        }
    } else {
        // fp32 rows, no plane (RR_NO_SHADOW / no room for it): every row takes the exact chain, two M-tiles' loads in flight
        constexpr int U = RR_RESCORE_U / 2;
        for (int s0 = ((int)blockIdx.x * 4 + wave) * U; s0 < n; s0 += (int)gridDim.x * 4 * U) {
            uint32_t m8[U];
#pragma unroll
            for (int u = 0; u < U; ++u) m8[u] = mtiles[s0 + u < n ? s0 + u : n - 1];
            f32x4 y[U][2][6];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    int64_t row = (int64_t)m8[u] * 8 + 4 * it + grp;
                    row = row < last_row ? row : last_row;
                    const f32x4* p = static_cast<const f32x4*>(mat) + row * 96 + sub;
#pragma unroll
                    for (int i = 0; i < 6; ++i) y[u][it][i] = p[16 * i];
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const f32x4 qq = qs[16 * i + sub];
                        acc = __builtin_fmaf(y[u][it][i].x, qq.x, acc);
                        acc = __builtin_fmaf(y[u][it][i].y, qq.y, acc);
                        acc = __builtin_fmaf(y[u][it][i].z, qq.z, acc);
                        acc = __builtin_fmaf(y[u][it][i].w, qq.w, acc);
                    }
                    acc = rr_row16_sum(acc);
                    const int64_t row = (int64_t)m8[u] * 8 + 4 * it + grp;
                    if (sub == 0 && s0 + u < n) sc[(int64_t)(s0 + u) * 8 + 4 * it + grp] = (row < n_rows && acc == acc) ? acc : -INFINITY;
                }
        }
    }
}

// ------------------------------------------------------------------ host side
static int rr_flt_get_bounds(rr_index* ix, hipStream_t st, rr_flt_bounds* out) {
    if (ix->norm_bound < 0.f) {
        unsigned int* d = reinterpret_cast<unsigned int*>(rr_x3_scratch_of(ix).eps);   // borrowed for two words
        RR_HIP_TRY(hipMemsetAsync(d, 0, 2 * sizeof(unsigned int), st));
        const unsigned blocks = (unsigned)((ix->n_rows + 255) / 256);
        if (ix->dtype == RR_DTYPE_BF16)
            hipLaunchKernelGGL((rr_row_norm_max<true>), dim3(blocks), dim3(256), 0, st, ix->d_matrix, ix->n_rows, ix->dim_pad, d);
        else
            hipLaunchKernelGGL((rr_row_norm_max<false>), dim3(blocks), dim3(256), 0, st, ix->d_matrix, ix->n_rows, ix->dim_pad, d);
        unsigned int bits[2] = {0u, 0u};
        RR_HIP_TRY(hipMemcpyAsync(bits, d, sizeof(bits), hipMemcpyDeviceToHost, st));
        RR_HIP_TRY(hipStreamSynchronize(st));
        float nb[2];
        memcpy(nb, bits, sizeof(nb));
        ix->norm_bound = nb[0] * 1.0001f;
        ix->delta_bound = nb[1] * 1.0001f;
    }
    out->row_norm = ix->norm_bound;
    out->row_delta = ix->delta_bound;
    return RR_OK;
}

// dual: the geometry of a two-set launch (rr_scan_flt16<.., DUAL>): half the resident waves per set, runs twice as long,
// eighth runs as selection groups (the same number and size of groups as a single-set launch)
template <int NQ2, bool A_BF16>
static rr_scan_geom rr_flt_geom(rr_index* ix, bool dual = false) {
    constexpr int THREADS = RR_FLT_THREADS(NQ2);
    static int waves = 0;
    if (!waves) waves = A_BF16 ? rr_resident_waves((const void*)rr_scan_flt16<NQ2>, 512, ix->device)
                               : rr_resident_waves((const void*)rr_scan_flt<NQ2>, THREADS, ix->device);
    rr_scan_geom G = rr_make_geom(ix, dual ? waves / 8 : waves / 4);
    G.qs = 32 * NQ2;
    G.mm_pairs = 3;
    // selection groups = quarter runs: ~4x fewer tile maxima to open per group (at most 8192 groups)
    const int gcap = dual ? 8 : 4;
    G.gpw = RR_MAX_SCAN_WAVES / G.n_waves < gcap ? (RR_MAX_SCAN_WAVES / G.n_waves < 1 ? 1 : RR_MAX_SCAN_WAVES / G.n_waves) : gcap;
    if (G.gpw > G.tiles_per_wave) G.gpw = (int32_t)G.tiles_per_wave;
    G.tiles_per_group = (G.tiles_per_wave + G.gpw - 1) / G.gpw;
    G.gpw = (int32_t)((G.tiles_per_wave + G.tiles_per_group - 1) / G.tiles_per_group);     // no empty trailing groups
    return G;
}

// geometry of rr_scan_fltq: one run per resident workgroup (= CU), 32 selection groups per run
static rr_scan_geom rr_fltq_geom(rr_index* ix) {
    static int runs = 0;
    if (!runs) runs = rr_resident_waves((const void*)rr_scan_fltq<true>, 256, ix->device) / 4;
    rr_scan_geom G = rr_make_geom(ix, runs / 4);      // (rr_make_geom counts 4 "waves" per resident block)
    G.qs = RR_FLT_MAXQ;
    G.mm_pairs = 3;
    G.gpw = RR_MAX_SCAN_WAVES / G.n_waves < 32 ? (RR_MAX_SCAN_WAVES / G.n_waves < 1 ? 1 : RR_MAX_SCAN_WAVES / G.n_waves) : 32;
    if (G.gpw > G.tiles_per_wave) G.gpw = (int32_t)G.tiles_per_wave;
    G.tiles_per_group = (G.tiles_per_wave + G.gpw - 1) / G.gpw;
    // (a group's end takes the wave out of the hand-scheduled loop -- accumulators copied out and in, the matrix pipe drained --:
    //  at least 16 M-tiles per group, which a small shard would otherwise undercut; the selection only needs > pool groups)
    if (G.tiles_per_group < 8 && (int64_t)G.n_waves * ((G.tiles_per_wave + 7) / 8) > 2048) G.tiles_per_group = 8;
    G.tiles_per_group += G.tiles_per_group & 1;       // even: a group closes on a multiple of four M-tiles (the unrolled loop)
    G.gpw = (int32_t)((G.tiles_per_wave + G.tiles_per_group - 1) / G.tiles_per_group);
    return G;
}

// fp32 rows -> the bf16 filter plane (round to nearest even, the rounding rr_row_norm_max<false> bounds)
__global__ __launch_bounds__(256) void rr_shadow_from_f32(const f32x4* __restrict__ src, u32x2* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 x = __builtin_nontemporal_load(src + i);
    const bf16x2 lo = bf16x2{(__bf16)x.x, (__bf16)x.y}, hi = bf16x2{(__bf16)x.z, (__bf16)x.w};
    dst[i] = u32x2{__builtin_bit_cast(unsigned int, lo), __builtin_bit_cast(unsigned int, hi)};
}

static int rr_flt_ensure_shadow(rr_index* ix, hipStream_t st) {
    if (ix->shadow_valid) return RR_OK;
    const size_t bytes = (size_t)ix->n_rows * 384 * 2;
    if (!ix->d_shadow) {
        const hipError_t e = hipMalloc((void**)&ix->d_shadow, bytes);
        if (e != hipSuccess) {          // no room for the plane: scan the fp32 rows (correct, twice the bytes)
            (void)hipGetLastError();
            ix->use_shadow = 0;
            return RR_OK;
        }
    }
    const int64_t n4 = ix->n_rows * 96;
    hipLaunchKernelGGL(rr_shadow_from_f32, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const f32x4*>(ix->d_matrix), reinterpret_cast<u32x2*>(ix->d_shadow), n4);
    RR_HIP_TRY(hipGetLastError());
    ix->shadow_valid = true;
    return RR_OK;
}

// RR_FLT_TUNE: debug-harness switches of the stamped rr_scan_flt16 variants only (32: one wave per SIMD)
static int rr_flt_tune() {
    static const int t = getenv("RR_FLT_TUNE") ? atoi(getenv("RR_FLT_TUNE")) : 0;
    return t;
}

// SCAN_BF16: element type of the matrix the filter scan streams (the index's own bf16 rows, or the bf16 plane of an
// fp32 index); ROWS_BF16: storage of the index, i.e. of the rows the candidates are rescored on.
//
// One scan launch ("set") = query planes + bounds, [store prefilter thresholds], the scan.  Sets 0 and 1 have their own
// planes, eps / sigma entries (at RR_FLT_MAXQ * set) and tile / group maxima (at set * the strides below), so that two
// scan launches can be followed by ONE selection + rescoring sequence over up to RR_SEL_MAXQ queries: the selection
// kernels are latency-bound, one workgroup per query -- 128 of them fill half the CUs.
static int64_t rr_flt_mmax_set_stride(const rr_scan_geom& G) { return (int64_t)2 * G.n_tiles * RR_FLT_MAXQ; }   // 4-byte words
static int64_t rr_flt_smax_set_stride() { return (int64_t)RR_FLT_MAXQ * RR_MAX_SCAN_WAVES; }

template <int NQ2, bool SCAN_BF16>
static int rr_flt_scan_set(rr_index* ix, int set, const rr_scan_geom& G, const void* scan_mat, const float* d_q, int nq,
                           int pool, rr_flt_bounds bounds, hipStream_t st, const float** sigma_out,
                           bool launch_scan = true, bool allow_prefilter = true, int prep_sets = 1) {   // prep_sets: planes + bounds of 0 | this | this and the next set (one launch)
    constexpr int THREADS = RR_FLT_THREADS(NQ2);
    constexpr int QN = 32 * NQ2;
    unsigned short* plane = reinterpret_cast<unsigned short*>(ix->d_qplanes) + (size_t)set * RR_FLT_MAXQ * 384;
    const rr_x3_scratch X = rr_x3_scratch_of(ix);
    float* eps = X.eps + set * RR_FLT_MAXQ;
    float* gmax = ix->d_gmax + set * rr_flt_mmax_set_stride(G);
    uint32_t* smax = ix->d_smax + set * rr_flt_smax_set_stride();
    if (prep_sets && !ix->flt_prep_fresh)         // (rr_flt_pad_prep has done it with the padding, for every slot of the call)
        hipLaunchKernelGGL(rr_flt_prep_queries, dim3(QN * prep_sets), dim3(64), 0, st, d_q, plane, eps, bounds);   // (planes in memory order)
    const dim3 grid((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), block(THREADS);
    // Store prefilter (bf16 stream, >= 2M rows): a 1/64 tile sample gives every query sigma = its m-th largest sampled
    // tile maximum - 2.05 eps.  The m-th largest of a 1/stride sample sits near rank m * stride of all rows; m leaves the
    // probability that the sample alone holds m of the true top-`pool` rows (sigma above the final threshold: that
    // query falls back to the exact pass) below 1e-9 (binomial(pool, 1/stride) tail).
    const float* sigma = nullptr;
    uint32_t* dummy = nullptr;
    static const bool no_prefilter = getenv("RR_NO_PREFILTER") != nullptr;
    const int64_t n_tiles32 = (G.n_rows + 31) / 32;
    if (SCAN_BF16 && !no_prefilter && allow_prefilter && G.n_rows >= 2000000) {
        const int stride = 64;
        int64_t n_samp = (n_tiles32 - 1) / stride;
        if (n_samp > RR_FLT_SAMP_CAP) n_samp = RR_FLT_SAMP_CAP;
        const int m = 16 + (4 * pool + stride - 1) / stride;
        if (!ix->d_flt_samp) {
            RR_HIP_TRY(hipMalloc((void**)&ix->d_flt_samp, sizeof(float) * (size_t)RR_FLT_SAMP_CAP * RR_FLT_MAXQ));
            RR_HIP_TRY(hipMalloc((void**)&ix->d_flt_sigma, sizeof(float) * 2 * RR_FLT_MAXQ));
        }
        float* sg = ix->d_flt_sigma + set * RR_FLT_MAXQ;
        hipLaunchKernelGGL((rr_flt_sample<NQ2>), dim3(256), dim3(512), 0, st, reinterpret_cast<const u32x4*>(scan_mat),
                           reinterpret_cast<const u32x4*>(plane), stride, (int)n_samp, ix->d_flt_samp);
        static const int force = getenv("RR_FLT_SIGMA_FORCE") ? atoi(getenv("RR_FLT_SIGMA_FORCE")) : 0;
        hipLaunchKernelGGL(rr_flt_sigma, dim3(QN), dim3(256), 0, st, ix->d_flt_samp, (int)n_samp, QN, nq, m, eps, sg, force);
        sigma = sg;
        // one line per scan wave behind the two sets of tile words (rr_ensure_scratch)
        dummy = reinterpret_cast<uint32_t*>(ix->d_gmax) + (size_t)4 * G.n_tiles * RR_FLT_MAXQ;
    }
    *sigma_out = sigma ? ix->d_flt_sigma : nullptr;        // (set 0's base: the selection adds set * RR_FLT_MAXQ itself)
    if (!launch_scan) {                                    // the caller scans both sets in one launch
        RR_HIP_TRY(hipGetLastError());
        return RR_OK;
    }
    const int slot = rr_scan_events_begin(ix, st);
    rr_scan_note(ix, 5, NQ2, nq, 1, SCAN_BF16 ? 2 : 4);
    if (SCAN_BF16)
        hipLaunchKernelGGL((rr_scan_flt16<NQ2>), grid, block, 0, st, reinterpret_cast<const u32x4*>(scan_mat), G,
                           reinterpret_cast<const u32x4*>(plane), gmax, smax, eps, nq, sigma, dummy, 0, (int64_t)0,
                           (uint32_t*)nullptr, 0u, rr_flt_tune());
    else
        hipLaunchKernelGGL((rr_scan_flt<NQ2>), grid, block, 0, st, reinterpret_cast<const u32x4*>(scan_mat), G,
                           reinterpret_cast<const u32x4*>(plane), gmax, smax, eps, nq, sigma, dummy);
    rr_scan_events_end(ix, slot, st);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

// selection + rescoring + ordering + per-64 fallback over nq_a (+ nq_b) queries of one (two) scan launches
template <bool ROWS_BF16>
static int rr_flt_finish(rr_index* ix, const rr_scan_geom& G, const float* d_q, int nq_a, int nq_b, int pool,
                         int64_t* d_rows, float* d_scores, const float* sigma, hipStream_t st, const float* floor = nullptr,
                         int parts = 7) {
    const rr_x3_scratch X = rr_x3_scratch_of(ix);
    const int nq = nq_a + nq_b;
    if (parts & 1)
        rr_launch_select_mtiles(ix, G, nq_a, pool, st, X.eps, sigma, nq_b, rr_flt_mmax_set_stride(G), rr_flt_smax_set_stride(), floor);
    // (the plane pre-scoring of the rescored rows: fp32 storage with a valid plane.  Under a corpus-wide floor the row cut is
    //  the floor: rows below it come back as -inf and only fill the shard's list up -- the merge never takes them)
    static const bool no_pre = getenv("RR_NO_RESCORE_PLANE") != nullptr;
    const u32x4* plane_rows = (!ROWS_BF16 && ix->shadow_valid && ix->d_shadow && !no_pre)
                                  ? reinterpret_cast<const u32x4*>(ix->d_shadow) : nullptr;
    // (16 M-tiles per workgroup and pass, ~500-650 listed per query: up to 3 passes; under a corpus-wide floor a shard opens
    //  ~pool / 8 M-tiles per query: a quarter of the workgroups.  Workgroups with nothing to do return before their barrier.)
    if (parts & 2)
        hipLaunchKernelGGL((rr_rescore_chain<ROWS_BF16>), dim3(floor ? 4 : 16, nq), dim3(256), 0, st, ix->d_matrix, G.n_rows, d_q,
                           X.mtiles, X.count, X.fb, X.sc, plane_rows, X.tau, X.eps);
    RR_HIP_TRY(hipGetLastError());
    if (!(parts & 4)) return RR_OK;
    rr_launch_select_rescored(ix, G, nq, pool, d_rows, d_scores, st, floor != nullptr);
    RR_HIP_TRY(hipGetLastError());
    // Flagged queries: at most eight in the call -> the single-query chain (bit for bit a batch of one); more -> the stored-score
    // pass of the split-operand scan, 64 queries per slice of ONE pair of launches; every launch returns at once when no
    // flag (of its queries) is up.
    if (!ROWS_BF16) {
        const int rc = rr_dense_listed_fallback(ix, d_q, nq, pool, d_rows, d_scores, X.fb, st);
        if (rc != RR_OK) return rc;
    }
    return rr_dense_x3w_fallback_all(ix, d_q, nq, pool, d_rows, d_scores, X.fb, st);
}

// What phase 1 of a two-phase call (row shards: scan, exchange a bound, select) leaves for phase 2.
struct rr_flt_pending {
    bool valid;
    rr_scan_geom G;
    const float* d_q;
    int nq_a, nq_b, pool;
    const float* sigma;
    bool rows_bf16;
};
struct rr_flt_phase {            // how a chunk function was called
    int phase = 0;               // 0 both, 1 scan only (+ bound), 2 selection only
    int kth = 0;
    float* d_bound = nullptr;
    const float* d_floor = nullptr;
};
static rr_flt_pending* rr_flt_pending_of(rr_index* ix) {
    if (!ix->flt_pending) ix->flt_pending = calloc(1, sizeof(rr_flt_pending));
    return static_cast<rr_flt_pending*>(ix->flt_pending);
}
void rr_flt_drop_pending(rr_index* ix) {
    if (ix->flt_pending) static_cast<rr_flt_pending*>(ix->flt_pending)->valid = false;
}
// the tail of every scan path: straight on to the selection, or park the state for phase 2
template <bool ROWS_BF16>
static int rr_flt_after_scan(rr_index* ix, const rr_scan_geom& G, const float* d_q, int nq_a, int nq_b, int pool,
                             int64_t* d_rows, float* d_scores, const float* sigma, hipStream_t st, const rr_flt_phase& ph) {
    if (ph.phase != 1) return rr_flt_finish<ROWS_BF16>(ix, G, d_q, nq_a, nq_b, pool, d_rows, d_scores, sigma, st);
    rr_flt_pending* p = rr_flt_pending_of(ix);
    if (!p) return RR_E_HIP;
    *p = rr_flt_pending{true, G, d_q, nq_a, nq_b, pool, sigma, ROWS_BF16};
    if (ph.kth > 0 && ph.d_bound)     // (kth = 0: the pipelined single-GPU K1 wants the split, not a bound)
        rr_launch_group_kth(ix, G, nq_a, nq_b, ph.kth, rr_x3_scratch_of(ix).eps, ph.d_bound, rr_flt_smax_set_stride(), st);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

template <int NQ2, bool SCAN_BF16, bool ROWS_BF16>
static int rr_dense_chunk_flt_t(rr_index* ix, const void* scan_mat, const float* d_q, int nq, int pool, int64_t* d_rows,
                                float* d_scores, rr_flt_bounds bounds, hipStream_t st, const rr_flt_phase& ph = rr_flt_phase()) {
    const rr_scan_geom G = rr_flt_geom<NQ2, SCAN_BF16>(ix);
    const float* sigma = nullptr;
    const int rc = rr_flt_scan_set<NQ2, SCAN_BF16>(ix, 0, G, scan_mat, d_q, nq, pool, bounds, st, &sigma);
    if (rc != RR_OK) return rc;
    return rr_flt_after_scan<ROWS_BF16>(ix, G, d_q, nq, 0, pool, d_rows, d_scores, sigma, st, ph);
}

// 129 .. 256 queries whose second part still fills a 128-slot launch (> 64 queries): two scan launches, one selection
template <bool SCAN_BF16, bool ROWS_BF16>
static int rr_dense_pair_flt_t(rr_index* ix, const void* scan_mat, const float* d_q, int nq_a, int nq_b, int pool,
                               int64_t* d_rows, float* d_scores, rr_flt_bounds bounds, hipStream_t st,
                               const rr_flt_phase& ph = rr_flt_phase()) {
    // bf16 stream: both sets in ONE launch, the matrix leaves HBM once for the 256 queries (rr_scan_flt16<.., DUAL>).
    // RR_NO_DUAL=1: two launches back to back (A/B).  An fp32 stream (no plane) keeps the two launches: its ring is
    // two segments per M-tile and the pair's lock-step window would be twice as wide.
    static const bool no_dual = getenv("RR_NO_DUAL") != nullptr;
    const bool dual = SCAN_BF16 && !no_dual;
    // the two-set launch runs the query-stationary kernel (rows through LDS once per CU); RR_NO_FLTQ=1: rr_scan_flt16<.., DUAL>
    static const bool use_q = getenv("RR_NO_FLTQ") == nullptr;
    const bool fltq = dual && use_q;
    const rr_scan_geom G = fltq ? rr_fltq_geom(ix) : rr_flt_geom<4, SCAN_BF16>(ix, dual);
    const float *sg0 = nullptr, *sg1 = nullptr;
    // The store prefilter pays where the scan is HBM-bound (one set per launch: 1.41 -> 1.31 ms for 0.06 ms of sample
    // and sigma).  The two-set launch is paced by the matrix side: there the skipped stores save 0.08 ms per launch and
    // the two samples cost 0.125 (r02, same box: 2.74 -> 2.67 ms per step without them), so it runs without.
    static const bool dual_prefilter = getenv("RR_DUAL_PREFILTER") != nullptr;
    const bool pre = !dual || dual_prefilter;
    // (nq_a = RR_FLT_MAXQ: the two sets' query slots, planes and bounds are contiguous -- one preparation launch for both
    //  when nothing sits between them)
    const bool prep_both = dual && !pre && nq_a == RR_FLT_MAXQ;
    int rc = rr_flt_scan_set<4, SCAN_BF16>(ix, 0, G, scan_mat, d_q, nq_a, pool, bounds, st, &sg0, !dual, pre, prep_both ? 2 : 1);
    if (rc != RR_OK) return rc;
    rc = rr_flt_scan_set<4, SCAN_BF16>(ix, 1, G, scan_mat, d_q + (int64_t)nq_a * ix->dim_pad, nq_b, pool, bounds, st, &sg1, !dual, pre,
                                       prep_both ? 0 : 1);
    if (rc != RR_OK) return rc;
    if (fltq) {
        const rr_x3_scratch X = rr_x3_scratch_of(ix);
        const int slot = rr_scan_events_begin(ix, st);
        rr_scan_note(ix, 5, 9, nq_a + nq_b, 1, 2);
        static const bool noasm = getenv("RR_FLTQ_NOASM") != nullptr;    // the C++ bodies everywhere (A/B)
        if (noasm)
            hipLaunchKernelGGL((rr_scan_fltq<false>), dim3(G.n_waves), dim3(256), 0, st, reinterpret_cast<const u32x4*>(scan_mat), G,
                               reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, X.eps, nq_a, nq_b,
                               rr_flt_mmax_set_stride(G), (unsigned long long*)nullptr);
        else
            hipLaunchKernelGGL((rr_scan_fltq<true>), dim3(G.n_waves), dim3(256), 0, st, reinterpret_cast<const u32x4*>(scan_mat), G,
                               reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, X.eps, nq_a, nq_b,
                               rr_flt_mmax_set_stride(G), (unsigned long long*)nullptr);
        rr_scan_events_end(ix, slot, st);
        RR_HIP_TRY(hipGetLastError());
    } else if (dual) {
        constexpr int THREADS = RR_FLT_THREADS(4);
        const int nb = (G.n_waves + THREADS / 64 - 1) / (THREADS / 64);        // workgroups per set
        const dim3 grid(((nb + 7) / 8) * 16), block(THREADS);
        const rr_x3_scratch X = rr_x3_scratch_of(ix);
        uint32_t* dummy = sg0 ? reinterpret_cast<uint32_t*>(ix->d_gmax) + (size_t)4 * G.n_tiles * RR_FLT_MAXQ : nullptr;
        const int slot = rr_scan_events_begin(ix, st);
        rr_scan_note(ix, 5, 8, nq_a + nq_b, 1, 2);
        static const bool no_couple = getenv("RR_NO_COUPLE") != nullptr;
        if (!ix->d_flt_prog) {
            RR_HIP_TRY(hipMalloc((void**)&ix->d_flt_prog, sizeof(uint32_t) * 2 * RR_MAX_SCAN_WAVES));
            RR_HIP_TRY(hipMemsetAsync(ix->d_flt_prog, 0, sizeof(uint32_t) * 2 * RR_MAX_SCAN_WAVES, st));
        }
        ix->flt_seq = (ix->flt_seq + 1) & 0x7FFFu;
        hipLaunchKernelGGL((rr_scan_flt16<4, 0, true>), grid, block, 0, st, reinterpret_cast<const u32x4*>(scan_mat), G,
                           reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, X.eps, nq_a, sg0, dummy,
                           nq_b, rr_flt_mmax_set_stride(G), no_couple ? (uint32_t*)nullptr : ix->d_flt_prog,
                           (uint32_t)(ix->flt_seq + 1) << 16, rr_flt_tune());
        rr_scan_events_end(ix, slot, st);
        RR_HIP_TRY(hipGetLastError());
    }
    return rr_flt_after_scan<ROWS_BF16>(ix, G, d_q, nq_a, nq_b, pool, d_rows, d_scores, sg0, st, ph);
}

int rr_flt_pad_prep(rr_index* ix, const float* d_queries, int nq, int slots, hipStream_t st) {
    ix->flt_prep_fresh = false;
    rr_flt_bounds nb;
    const int rc = rr_flt_get_bounds(ix, st, &nb);
    if (rc != RR_OK) return rc;
    if (!(nb.row_norm < 3.0e18f) || slots > RR_SEL_MAXQ || ix->dim_pad != 384) return RR_FLT_NO_BOUND;
    const rr_x3_scratch X = rr_x3_scratch_of(ix);
    hipLaunchKernelGGL(rr_flt_pad_prep_queries, dim3(slots), dim3(64), 0, st, d_queries, nq, ix->dim, ix->d_q,
                       reinterpret_cast<unsigned short*>(ix->d_qplanes), X.eps, nb);
    RR_HIP_TRY(hipGetLastError());
    ix->flt_prep_fresh = true;
    return RR_OK;
}

// (every exit of a filter call -- served or declined -- leaves the planes "not fresh": they belong to ONE call)
struct rr_flt_fresh_guard {
    rr_index* ix;
    ~rr_flt_fresh_guard() { ix->flt_prep_fresh = false; }
};

int rr_dense_chunk_flt(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                       float* d_scores, hipStream_t st, int phase, int kth, float* d_bound, const float* d_floor, int parts) {
    rr_flt_fresh_guard fresh_guard{ix};
    if (d_q != ix->d_q) ix->flt_prep_fresh = false;       // a later chunk of a long call: its planes are not the prepared ones
    if (phase == 2) {
        // selection of the scan a phase-1 call left behind (same queries, same pool), with the exchanged floor
        rr_flt_pending* p = rr_flt_pending_of(ix);
        if (!p || !p->valid || p->nq_a + p->nq_b != nq || p->pool != pool || p->d_q != d_q) return RR_FLT_SMALL;   // (caller: plain call)
        if (parts & 4) p->valid = false;
        return p->rows_bf16 ? rr_flt_finish<true>(ix, p->G, d_q, p->nq_a, p->nq_b, pool, d_rows, d_scores, p->sigma, st, d_floor, parts)
                            : rr_flt_finish<false>(ix, p->G, d_q, p->nq_a, p->nq_b, pool, d_rows, d_scores, p->sigma, st, d_floor, parts);
    }
    rr_flt_phase ph;
    ph.phase = phase;
    ph.kth = kth;
    ph.d_bound = d_bound;
    if (ix->flt_pending) static_cast<rr_flt_pending*>(ix->flt_pending)->valid = false;
    // A small matrix has too few tile groups for the threshold to mean anything (every query would be
    // flagged): the caller runs the per-row-chain VALU scans instead, which cost microseconds there and
    // keep the answer bitwise equal to what a large index (or the unsharded one) gives through rescoring.
    if ((ix->n_rows + 63) / 64 < 8 * (int64_t)pool) return RR_FLT_SMALL;
    rr_flt_bounds nb;
    int rc = rr_flt_get_bounds(ix, st, &nb);
    if (rc != RR_OK) return rc;
    if (!(nb.row_norm < 3.0e18f))     // no usable bound (inf / NaN rows, or squares that overflow): exact scans, 64 at a time
        return RR_FLT_NO_BOUND;
    const bool b = ix->dtype == RR_DTYPE_BF16;
    static const bool no_shadow = getenv("RR_NO_SHADOW") != nullptr;
    if (nq > RR_FLT_MAXQ) {
        // a pair of launches (the caller offers > 128 queries only when the second part is > 64: same kernel, same geometry)
        const int nq_a = RR_FLT_MAXQ, nq_b = nq - RR_FLT_MAXQ;
        if (!b && ix->use_shadow && !no_shadow) {
            rc = rr_flt_ensure_shadow(ix, st);
            if (rc != RR_OK) return rc;
            if (ix->shadow_valid)
                return rr_dense_pair_flt_t<true, false>(ix, ix->d_shadow, d_q, nq_a, nq_b, pool, d_rows, d_scores, nb, st, ph);
        }
        return b ? rr_dense_pair_flt_t<true, true>(ix, ix->d_matrix, d_q, nq_a, nq_b, pool, d_rows, d_scores, nb, st, ph)
                 : rr_dense_pair_flt_t<false, false>(ix, ix->d_matrix, d_q, nq_a, nq_b, pool, d_rows, d_scores, nb, st, ph);
    }
    if (!b && ix->use_shadow && !no_shadow) {
        // fp32 storage: the scan streams the bf16 filter plane (half the bytes per launch; the approximate scores
        // and their bound are those of the on-the-fly rounding), the candidates are rescored on the fp32 rows
        rc = rr_flt_ensure_shadow(ix, st);
        if (rc != RR_OK) return rc;
        if (ix->shadow_valid) {
            const void* sm = ix->d_shadow;
            if (nq <= 32) return rr_dense_chunk_flt_t<1, true, false>(ix, sm, d_q, nq, pool, d_rows, d_scores, nb, st, ph);
            if (nq <= 64) return rr_dense_chunk_flt_t<2, true, false>(ix, sm, d_q, nq, pool, d_rows, d_scores, nb, st, ph);
            return rr_dense_chunk_flt_t<4, true, false>(ix, sm, d_q, nq, pool, d_rows, d_scores, nb, st, ph);
        }
    }
    const void* m = ix->d_matrix;
    if (nq <= 32)
        return b ? rr_dense_chunk_flt_t<1, true, true>(ix, m, d_q, nq, pool, d_rows, d_scores, nb, st, ph)
                 : rr_dense_chunk_flt_t<1, false, false>(ix, m, d_q, nq, pool, d_rows, d_scores, nb, st, ph);
    if (nq <= 64)
        return b ? rr_dense_chunk_flt_t<2, true, true>(ix, m, d_q, nq, pool, d_rows, d_scores, nb, st, ph)
                 : rr_dense_chunk_flt_t<2, false, false>(ix, m, d_q, nq, pool, d_rows, d_scores, nb, st, ph);
    return b ? rr_dense_chunk_flt_t<4, true, true>(ix, m, d_q, nq, pool, d_rows, d_scores, nb, st, ph)
             : rr_dense_chunk_flt_t<4, false, false>(ix, m, d_q, nq, pool, d_rows, d_scores, nb, st, ph);
}

#ifdef RR_DEBUG_HARNESS
#include "rr_debug.h"
// Timing-only ablations of the 128-query fp32 filter scan (tools/flt_ablate.py).  Garbage in the scratch afterwards.
template <int DBG, bool PLANE = false>
static float rr_debug_time_flt(rr_index* ix, hipStream_t st, int reps) {
    constexpr int THREADS = RR_FLT_THREADS(4);
    const rr_scan_geom G = rr_flt_geom<4, PLANE>(ix);
    const dim3 grid((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), block(THREADS);
    unsigned long long* d_stamps = nullptr;
    if (DBG & 128) {
        if (hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 6 * RR_MAX_SCAN_WAVES) != hipSuccess) return -1.f;
        hipMemset(d_stamps, 0, sizeof(unsigned long long) * 6 * RR_MAX_SCAN_WAVES);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float total = 0.f;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(e0, st);
        if constexpr (PLANE)
            hipLaunchKernelGGL((rr_scan_flt16<4, DBG>), grid, block, 0, st, reinterpret_cast<const u32x4*>(ix->d_shadow), G,
                               reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, rr_x3_scratch_of(ix).eps, 128,
                               (const float*)nullptr, (uint32_t*)nullptr, 0, (int64_t)0, reinterpret_cast<uint32_t*>(d_stamps), 0u,
                               rr_flt_tune());
        else
            hipLaunchKernelGGL((rr_scan_flt<4, DBG>), grid, block, 0, st, reinterpret_cast<const u32x4*>(ix->d_matrix), G,
                               reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, rr_x3_scratch_of(ix).eps, 128,
                               (const float*)nullptr, (uint32_t*)nullptr);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (r) total += ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (DBG & 128) {        // per-wave shader cycles of the last launch: K-loops, epilogues, ring waits, whole scan
        std::vector<unsigned long long> h((size_t)6 * G.n_waves);
        hipMemcpy(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipFree(d_stamps);
        double sum[6] = {0, 0, 0, 0, 0, 0};
        int live = 0;
        for (int w = 0; w < G.n_waves; ++w) {
            if (h[(size_t)6 * w + 3]) ++live;
            for (int k = 0; k < 6; ++k) sum[k] += (double)h[(size_t)6 * w + k];
        }
        // (with RR_FLT_TUNE=32 only every other wave runs: the per-M-tile figures are then per TWO M-tiles of a running wave)
        const double tiles = (double)((G.n_rows + 31) / 32);
        fprintf(stderr, "[flt stamps] per M-tile and wave, shader cycles: K-loop %.0f, between epilogue and K-loop (or ring waits) %.0f, "
                        "epilogue %.0f; tile loop per wave %.0f cycles = %.1f us (100 MHz clock), kernel entry to the loop %.1f us, "
                        "%d waves wrote\n", sum[0] / tiles, sum[2] / tiles, sum[1] / tiles, sum[3] / (live ? live : 1),
                sum[4] / (live ? live : 1) / 100.0, sum[5] / (live ? live : 1) / 100.0, live);
    }
    return total / reps;
}

template <int ABL>
static float rr_debug_time_fltq_asm(rr_index* ix, hipStream_t st, int reps) {
    if (!ix->shadow_valid) return -1.f;
    const rr_scan_geom G = rr_fltq_geom(ix);
    unsigned long long* d_st = nullptr;
    if (hipMalloc((void**)&d_st, sizeof(unsigned long long) * 16 * 1024) != hipSuccess) return -1.f;
    hipMemset(d_st, 0, sizeof(unsigned long long) * 16 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float total = 0.f;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(e0, st);
        hipLaunchKernelGGL((rr_scan_fltq<true, ABL>), dim3(G.n_waves), dim3(256), 0, st, reinterpret_cast<const u32x4*>(ix->d_shadow), G,
                           reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, rr_x3_scratch_of(ix).eps, 128, 128,
                           rr_flt_mmax_set_stride(G), ABL ? d_st : (unsigned long long*)nullptr);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (r) total += ms;
    }
    if (ABL) {
        std::vector<unsigned long long> hst((size_t)16 * G.n_waves);
        hipMemcpy(hst.data(), d_st, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double cyc = 0, cmax = 0;
        for (size_t k = 3; k < (size_t)16 * G.n_waves; k += 4) {
            cyc += (double)hst[k];
            if ((double)hst[k] > cmax) cmax = (double)hst[k];
        }
        cyc /= 4.0 * G.n_waves;
        const double mt = (double)((G.n_rows + 31) / 32) / G.n_waves;
        fprintf(stderr, "[fltq asm loop, ablation %d] %.0f cycles per wave (slowest wave %.0f) = %.0f per M-tile; %.3f ms per launch => %.2f GHz\n",
                ABL, cyc, cmax, cyc / mt, total / reps, cyc / (total / reps) * 1e-6);
    }
    hipFree(d_st); hipEventDestroy(e0); hipEventDestroy(e1);
    return total / reps;
}

// The hand-scheduled loop against the C++ bodies: both kernels scan the plane with the planes / bounds the last 256-query
// search left in the scratch; every tile word and every group maximum of both query sets must come out bit for bit the
// same.  out[0] = tile words that differ, out[1] = group maxima that differ, out[2] = words compared, out[3] = groups.
extern "C" int rr_debug_fltq_compare(rr_index* ix, int64_t* out) {
    RR_REQUIRE(ix && out && ix->shadow_valid && ix->scratch_q >= 64, "run a 256-query search on an fp32 index first");
    std::lock_guard<std::mutex> lock(ix->mu);
    hipStream_t st = nullptr;
    const rr_scan_geom G = rr_fltq_geom(ix);
    const size_t words = (size_t)2 * rr_flt_mmax_set_stride(G), groups = (size_t)2 * rr_flt_smax_set_stride();
    std::vector<uint32_t> w[2], g[2];
    for (int v = 0; v < 2; ++v) {
        RR_HIP_TRY(hipMemsetAsync(ix->d_gmax, 0xA5, words * 4, st));
        RR_HIP_TRY(hipMemsetAsync(ix->d_smax, 0xA5, groups * 4, st));
        if (v == 0)
            hipLaunchKernelGGL((rr_scan_fltq<false>), dim3(G.n_waves), dim3(256), 0, st, reinterpret_cast<const u32x4*>(ix->d_shadow), G,
                               reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, rr_x3_scratch_of(ix).eps, 128, 128,
                               rr_flt_mmax_set_stride(G), (unsigned long long*)nullptr);
        else
            hipLaunchKernelGGL((rr_scan_fltq<true>), dim3(G.n_waves), dim3(256), 0, st, reinterpret_cast<const u32x4*>(ix->d_shadow), G,
                               reinterpret_cast<const u32x4*>(ix->d_qplanes), ix->d_gmax, ix->d_smax, rr_x3_scratch_of(ix).eps, 128, 128,
                               rr_flt_mmax_set_stride(G), (unsigned long long*)nullptr);
        RR_HIP_TRY(hipGetLastError());
        w[v].resize(words);
        g[v].resize(groups);
        RR_HIP_TRY(hipMemcpy(w[v].data(), ix->d_gmax, words * 4, hipMemcpyDeviceToHost));
        RR_HIP_TRY(hipMemcpy(g[v].data(), ix->d_smax, groups * 4, hipMemcpyDeviceToHost));
    }
    int64_t dw = 0, dg = 0, first = -1;
    for (size_t i = 0; i < words; ++i)
        if (w[0][i] != w[1][i]) {
            if (first < 0) first = (int64_t)i;
            ++dw;
        }
    for (size_t i = 0; i < groups; ++i) dg += g[0][i] != g[1][i];
    out[0] = dw; out[1] = dg; out[2] = (int64_t)words; out[3] = (int64_t)groups; out[4] = first;
    out[5] = first >= 0 ? w[0][first] : 0; out[6] = first >= 0 ? w[1][first] : 0;
    if (dw && getenv("RR_DEBUG_FLTQ_DIFFS")) {          // where the words differ: by body of the unrolled loop, query, bit field
        int64_t by_body[4] = {0, 0, 0, 0}, by_nib[8] = {0}, by_q32[4] = {0}, shown = 0;
        const size_t per_set = words / 2;
        for (size_t i = 0; i < words; ++i) {
            if (w[0][i] == w[1][i]) continue;
            const size_t j = i % per_set, tile = j / RR_FLT_MAXQ, q = j % RR_FLT_MAXQ;
            const int64_t it = (int64_t)(tile % (2 * G.tiles_per_wave));
            by_body[it & 3]++;
            by_q32[q / 32]++;
            const uint32_t x = w[0][i] ^ w[1][i];
            for (int n = 0; n < 8; ++n) by_nib[n] += ((x >> (4 * n)) & 15u) != 0;
            if (shown++ < 24) fprintf(stderr, "  set %d tile %zu (in run %lld) query %zu: %08x vs %08x\n", (int)(i / per_set), tile, (long long)it, q, w[0][i], w[1][i]);
        }
        fprintf(stderr, "  by (tile in run) & 3: %lld %lld %lld %lld; by query / 32: %lld %lld %lld %lld; by nibble 0..7: %lld %lld %lld %lld %lld %lld %lld %lld\n",
                (long long)by_body[0], (long long)by_body[1], (long long)by_body[2], (long long)by_body[3], (long long)by_q32[0], (long long)by_q32[1],
                (long long)by_q32[2], (long long)by_q32[3], (long long)by_nib[0], (long long)by_nib[1], (long long)by_nib[2], (long long)by_nib[3],
                (long long)by_nib[4], (long long)by_nib[5], (long long)by_nib[6], (long long)by_nib[7]);
    }
    return RR_OK;
}

extern "C" int rr_debug_scan_flt(rr_index* ix, int32_t dbg, int32_t reps, float* out_ms) {
    RR_REQUIRE(ix && out_ms && ix->dtype == RR_DTYPE_F32 && ix->dim_pad == 384, "fp32 index of dim 384 expected");
    RR_REQUIRE(ix->scratch_q >= 64, "run a batched search first (allocates the scratch)");
    std::lock_guard<std::mutex> lock(ix->mu);
    hipStream_t st = nullptr;
    switch (dbg) {
        case 0: *out_ms = rr_debug_time_flt<0>(ix, st, reps); break;
        case 1: *out_ms = rr_debug_time_flt<1>(ix, st, reps); break;
        case 2: *out_ms = rr_debug_time_flt<2>(ix, st, reps); break;
        case 4: *out_ms = rr_debug_time_flt<4>(ix, st, reps); break;
        case 8: *out_ms = rr_debug_time_flt<8>(ix, st, reps); break;
        case 3: *out_ms = rr_debug_time_flt<3>(ix, st, reps); break;
        case 7: *out_ms = rr_debug_time_flt<7>(ix, st, reps); break;
        // (15 "ring loads + maxima stores only" and 31 "ring loads only" are gone: with nothing reading the ring registers the
        //  compiler let a v_permlane16_swap land on v2 / v6 while global_load_dwordx4 v[2:5] / v[6:9] were still in flight --
        //  tools/check_ring_hazards.py over the -DRR_DEBUG_HARNESS assembly, 8 violations each; the class of the round-2 fault.
        //  tests/test_ring_register_contract.py now walks the debug build too.)
        case 16: *out_ms = rr_debug_time_flt<16>(ix, st, reps); break;
        case 32: *out_ms = rr_debug_time_flt<32>(ix, st, reps); break;
        // 64: the full kernel over the bf16 filter plane (a batched search must have built it).  The ablated
        // variants are NOT built for the plane: `no epilogue` over the plane faulted on the GPU (r02: the ablations
        // change register liveness around the asynchronous ring loads; only the variants above were ever validated)
        case 64:
            RR_REQUIRE(ix->shadow_valid, "no bf16 filter plane yet: run a batched search first");
            *out_ms = rr_debug_time_flt<0, true>(ix, st, reps);
            break;
        // 128: the full kernel over the plane with every ring load served from cache (same code, same registers: only
        // the addresses differ) = the pace of everything but HBM
        case 128:
            RR_REQUIRE(ix->shadow_valid, "no bf16 filter plane yet: run a batched search first");
            *out_ms = rr_debug_time_flt<64, true>(ix, st, reps);
            break;
        // 256 / 320: in-kernel s_memtime stamps (K-loop / ring waits / epilogue per wave, printed to stderr), over HBM / from cache
        case 256:
            RR_REQUIRE(ix->shadow_valid, "no bf16 filter plane yet: run a batched search first");
            *out_ms = rr_debug_time_flt<128, true>(ix, st, reps);
            break;
        case 320:
            RR_REQUIRE(ix->shadow_valid, "no bf16 filter plane yet: run a batched search first");
            *out_ms = rr_debug_time_flt<192, true>(ix, st, reps);
            break;
        // 400 (16x16x32 kernel only; wrong results): stamped, cached, without the B-fragment reads
        // 3000: the hand-scheduled loop (the product's default 256-query kernel), timed like the variants above;
        // 3000 + bits: its generated timing ablations with the whole-loop cycle count (=> the shader clock of the launch)
#define RR_FLTQA_CASE(b) case 3000 + (b): *out_ms = rr_debug_time_fltq_asm<(b)>(ix, st, reps); break;
        RR_FLTQA_CASE(0) RR_FLTQA_CASE(128) RR_FLTQA_CASE(1) RR_FLTQA_CASE(2) RR_FLTQA_CASE(3) RR_FLTQA_CASE(4) RR_FLTQA_CASE(8)
        RR_FLTQA_CASE(64) RR_FLTQA_CASE(66) RR_FLTQA_CASE(1025) RR_FLTQA_CASE(1026) RR_FLTQA_CASE(1027) RR_FLTQA_CASE(1028) RR_FLTQA_CASE(1029) RR_FLTQA_CASE(1030)
#undef RR_FLTQA_CASE
        case 400:
            RR_REQUIRE(ix->shadow_valid, "no bf16 filter plane yet: run a batched search first");
            *out_ms = rr_debug_time_flt<192 | 2, true>(ix, st, reps);
            break;
        // (401 / 402 -- without the ring re-loads / without both -- produced r02_flt16_stamps_ablations_10M.txt and are gone: a
        //  variant that leaves asynchronously written registers unread lets the compiler hand them out again while the write
        //  is still on its way; one such build faulted on the GPU.  The same goes for MFMA results nobody reads.)
        default: RR_REQUIRE(false, "unknown ablation %d", dbg);
    }
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}
#endif  // RR_DEBUG_HARNESS
