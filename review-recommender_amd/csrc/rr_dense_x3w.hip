// rr_dense_x3w.hip -- K1 batched scan, 17..64 queries: split operands on 32x32x16 bf16 MFMA tiles.
//
// Same arithmetic idea as rr_dense_x3.hip (fp32 = exact sum of three bf16 terms, products exact,
// fp32 accumulation in the matrix core), on the wide tile: 32 matrix rows x 32 queries per MFMA.
// Why the wide tile: with 32..64 queries the 16x16x32 kernel is paced by the CU, not by HBM --
//   * a B fragment (ds_read_b128, 1 KiB per wave) feeds 8192 MACs there and 16384 here: half the
//     LDS traffic per row, and LDS time was level with MFMA time;
//   * a 16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16 cycles, a 32x32x16 for 8 of
//     its 32: the operand split (~5.5 VALU ops per element) now fits beside the matrix pipe;
//   * no ds_bpermute: a load instruction reads 16 rows x 64 B (lane l -> row l & 15, 16-B piece
//     l >> 4, the shape tools/membench measures at full HBM rate: 16 distinct lines per
//     instruction), two of them (rows 0-15 = X, rows 16-31 = Y) are turned into the MFMA A layout
//     (lane l -> row l & 31, k half l >> 5) by ONE v_permlane16_swap per dword pair:
//         X' = [X.row0 Y.row0 X.row2 Y.row2] = rows 0..31, pieces 0 | 2
//         Y' = [X.row1 Y.row1 X.row3 Y.row3] = rows 0..31, pieces 1 | 3      (row = 16 lanes)
//     fp32 matrix: (X', Y') are dims [8h, 8h+4), [8h+4, 8h+8) of a 16-dim K-step: natural order;
//     bf16 matrix: X' and Y' are the A operands of two K-steps over one 32-dim group, K-step v
//     slot (h, j) = dim 16h + 8v + j -- the query planes are written in that order.
// Stream structure: contiguous per-wave tile runs, a 24-unit VGPR ring of inline-asm loads with
// counted waits, refilled by halves in bursts of 12 (fp32: the ring is half a 32-row M-tile,
// the accumulators live across two ring segments; bf16: a whole M-tile), M-tile maxima only
// (STORE = false) + rescoring, or every score (STORE = true: the fallback / diagnostic pass).
#include <vector>
#include "rr_x3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RR_X3W_QSTRIDE 49   // 16-B units per query plane in LDS: 48 + 1 pad -> eight consecutive queries
                            // land in eight different 16-B bank groups (784 B stride = 16 mod 256)

// term i of the split product for one K-step: fp32 matrix a1q3 a3q1 a2q1 a1q1 a2q2 a1q2, bf16 matrix
// aq3 aq1 aq2.  One accumulator sees its terms in this order, K-steps ascending.  The order groups
// the uses of each query plane (q3 | q1 | q2) so that a plane's registers can be re-loaded for the
// next K-step right after its last term, with no second buffer, and still land many MFMAs before
// their first use there; the six products are exact, so the order only moves fp32 roundings.
#define RR_X3W_A_OF(i, f) ((i) == 1 ? (f).a3 : ((i) == 2 || (i) == 4) ? (f).a2 : (f).a1)       /* fp32 matrix */
#define RR_X3W_PL_OF(i, bf) ((bf) ? ((i) == 0 ? 2 : (i) == 1 ? 0 : 1) : ((i) == 0 ? 2 : (i) <= 3 ? 0 : 1))
template <bool A_BF16>
__device__ __forceinline__ void rr_x3w_term(int i, const rr_x3_afrag& f, const bf16x8 (&q)[3], f32x16& acc) {
    const bf16x8 a = A_BF16 ? f.a1 : RR_X3W_A_OF(i, f);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q[RR_X3W_PL_OF(i, A_BF16)], acc, 0, 0, 0);
}

// rows 0-15 (x) and rows 16-31 (y) in load order -> (lo, hi) units in the MFMA A layout
__device__ __forceinline__ void rr_x3w_to_mfma_lanes(u32x4 x, u32x4 y, u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const auto r = __builtin_amdgcn_permlane16_swap(x[d], y[d], false, false);
        lo[d] = r[0];
        hi[d] = r[1];
    }
}

// DBG != 0: timing-only ablations for tools/x3w_ablate.py (wrong results): bit 0 drops the operand
// split, bit 1 the B-fragment LDS reads, bit 2 the MFMAs, bit 3 the lane swaps; bit 6 skips the epilogue, bit 7 the LDS fill; bit 4 stamps the shader clock around
// the ring waits and the epilogue and leaves per-wave {total, waits, epilogue} cycles in `sims`.
template <int NQ2, bool A_BF16, bool STORE, int DBG = 0>
__global__ __launch_bounds__(((NQ2 == 2 && !(DBG & 32)) ? 512 : 256), 2) void rr_scan_x3w(
    const u32x4* __restrict__ mat, rr_scan_geom G, const u32x4* __restrict__ planes,  // [3][32*NQ2][48] units
    float* __restrict__ sims, float* __restrict__ gmax, uint32_t* __restrict__ smax,
    const int32_t* __restrict__ fallback, int n_flags, const float* __restrict__ raw_q = nullptr,
    int64_t sims_slice = 0, int64_t gmax_slice = 0, int64_t smax_slice = 0) {
    constexpr int THREADS = (NQ2 == 2 && !(DBG & 32)) ? 512 : 256;   // (DBG bit 5: one wave per SIMD)
    if (STORE && fallback && gridDim.y > 1) {
        // ONE launch for the fallback of up to four 64-query blocks (rr_dense_x3w_fallback_all): slice y = blockIdx.y serves
        // queries 64 y .. of the call with its own flags, query vectors and score / maxima scratch
        const int y = blockIdx.y;
        fallback += 64 * y;
        n_flags = n_flags - 64 * y < 64 ? n_flags - 64 * y : 64;
        raw_q += (int64_t)y * 64 * 384;
        sims += y * sims_slice; gmax += y * gmax_slice; smax += y * smax_slice;
    }
    if (STORE && fallback) {
        // (n_flags <= 64: one flag per lane, one load; every wave of the workgroup sees the same answer)
        const int li = threadIdx.x & 63;
        const int mine = li < n_flags ? fallback[li] : 0;
        if (!__any(mine)) return;                          // uniform: nobody needs the stored scores
    }
    constexpr int QN = 32 * NQ2;
    constexpr int ROWU = A_BF16 ? 48 : 96;            // 16-byte units per matrix row
    constexpr int SEGS = A_BF16 ? 1 : 2;              // ring segments (24 units per lane) per 32-row M-tile
    constexpr int NTERM = A_BF16 ? 3 : 6;
    __shared__ u32x4 qs[3 * QN * RR_X3W_QSTRIDE];
    const int tid = threadIdx.x;
    if (raw_q) {
        // fallback launches: split the fp32 queries here (what rr_split_queries would have left in `planes`,
        // bit for bit) -- one kernel launch fewer per fallback, which is enqueued behind every filter launch
        for (int i = tid; i < QN * RR_X3_UNITS; i += THREADS) {
            const int qn = i / RR_X3_UNITS, u = i % RR_X3_UNITS;
            u32x4 pl[3];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                float x1[2], x2[2], x3[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pos = u * 8 + 2 * e2 + k;                  // position in plane order
                    int src = pos;
                    if (A_BF16) {                                        // RR_X3_ORDER_WIDE_BF16
                        const int e = pos & 31, v = e >> 4, hh = (e >> 3) & 1, j = e & 7;
                        src = (pos & ~31) + 16 * hh + 8 * v + j;
                    }
                    const float x = raw_q[qn * 384 + src];
                    x1[k] = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
                    const float r1 = x - x1[k];
                    x2[k] = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
                    x3[k] = r1 - x2[k];
                }
                pl[0][e2] = rr_pack_hi(x1[0], x1[1]);
                pl[1][e2] = rr_pack_hi(x2[0], x2[1]);
                pl[2][e2] = rr_pack_hi(x3[0], x3[1]);
            }
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3) qs[(p3 * QN + qn) * RR_X3W_QSTRIDE + u] = pl[p3];
        }
    } else if (!(DBG & 128)) {
        for (int i = tid; i < 3 * QN * RR_X3_UNITS; i += THREADS)
            qs[(i / RR_X3_UNITS) * RR_X3W_QSTRIDE + (i % RR_X3_UNITS)] = planes[i];
    }
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
    const u32x4* px;                                  // current segment, rows 0-15 / 16-31 of the M-tile
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
    // ring unit j: pair j / 2 (the 64-byte group), j & 1 = 0: rows 0-15 (x), 1: rows 16-31 (y)
#define RR_X3W_LOAD(dst, j) \
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(((j) & 1) ? py : px), "n"(64 * ((j) / 2)) : "memory")
    u32x4 a[24];
    seg_ptrs(0);
#pragma unroll
    for (int j = 0; j < 24; ++j) RR_X3W_LOAD(a[j], j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // B-fragment addresses: one base register per query plane, every (t, K-step) an immediate offset
    // (ds_read offsets reach 64 KB; one base for all three planes made hipcc add a literal per read)
    int qlane[3];                                         // 16-byte unit index into qs
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        qlane[pl] = (pl * QN + c) * RR_X3W_QSTRIDE + h;           // + 32 t * QSTRIDE + 2 * kk
        asm volatile("" : "+v"(qlane[pl]));                       // (opaque: keeps the three bases apart)
    }
    float gm[NQ2];
    float2 pend[NQ2];                                     // !STORE: maxima of the M-tile just finished, not yet stored
    float tmax[NQ2];                                      // STORE: running maximum of the 64-row tile
#pragma unroll
    for (int t = 0; t < NQ2; ++t) {
        gm[t] = tmax[t] = -INFINITY;
        pend[t] = float2{-INFINITY, -INFINITY};
    }
    long long dbg_t0 = 0, dbg_wait = 0, dbg_epi = 0;   // (dbg_wait: unused since the ring became a steady flow)
    if (DBG & 16) dbg_t0 = clock64();

    // ---- software pipeline.  While the matrix pipe runs the MFMAs of K-step s (one per "slot"),
    // the wave's vector issue prepares K-step s + 1 in the gaps: lane swap + operand split of the
    // next A fragment (48 VALU ops for fp32, cut into equal shares per slot) and the ds_reads of the
    // next B fragments.  Left to itself the compiler emits [split][12 MFMAs] per K-step, and the
    // counters show the two running one after the other (MFMA busy 56 %, co-execution 19 %);
    // __builtin_amdgcn_sched_barrier(0) after every slot pins the interleaving written here.
    constexpr int STEPS = A_BF16 ? 24 : 12;               // K-steps per ring segment
    constexpr int NSLOT = NTERM * NQ2;                    // MFMAs per K-step
    constexpr int NV = A_BF16 ? 4 : 48;                   // VALU ops that prepare one A fragment pair
    constexpr int VPS = (NV + NSLOT - 1) / NSLOT;         // ... per slot
    constexpr unsigned HI16 = 0xFFFF0000u;
    u32x4 lo, hi;                  // the pair being prepared, in MFMA lanes
    float tmp[6];                  // split temporaries of the element pair in flight
    u32x4 pk[3];                   // the three bf16 planes of the fragment being built
    rr_x3_afrag af;                // operand of the current K-step
    bf16x8 qf[3][NQ2];                                    // B fragments by plane (q1, q2, q3)
    auto valu_op = [&](int k, const u32x4& x, const u32x4& y) {
        if (k < 4) {
            if (DBG & 8) {
                lo[k] = x[k];
                hi[k] = y[k];
                return;
            }
            const auto r = __builtin_amdgcn_permlane16_swap(x[k], y[k], false, false);
            lo[k] = r[0];
            hi[k] = r[1];
            return;
        }
        if (A_BF16 || k >= NV) return;
        if (DBG & 1) {
            if (k < 8) pk[0][k - 4] = lo[k - 4] ^ hi[k - 4], pk[1][k - 4] = lo[k - 4], pk[2][k - 4] = hi[k - 4];
            return;
        }
        const int e = (k - 4) / 11, o = (k - 4) % 11;     // element pair e = elements 2e, 2e + 1 of {lo, hi}
        const unsigned xa = (2 * e < 4) ? lo[(2 * e) & 3] : hi[(2 * e) & 3];
        const unsigned xb = (2 * e + 1 < 4) ? lo[(2 * e + 1) & 3] : hi[(2 * e + 1) & 3];
        switch (o) {   // same operations as rr_x3_split
            case 0: tmp[0] = __uint_as_float(xa & HI16); break;                          // h1a
            case 1: tmp[1] = __uint_as_float(xa) - tmp[0]; break;                        // r1a
            case 2: tmp[2] = __uint_as_float(__float_as_uint(tmp[1]) & HI16); break;     // h2a
            case 3: tmp[1] = tmp[1] - tmp[2]; break;                                     // h3a
            case 4: tmp[3] = __uint_as_float(xb & HI16); break;                          // h1b
            case 5: tmp[4] = __uint_as_float(xb) - tmp[3]; break;                        // r1b
            case 6: tmp[5] = __uint_as_float(__float_as_uint(tmp[4]) & HI16); break;     // h2b
            case 7: tmp[4] = tmp[4] - tmp[5]; break;                                     // h3b
            case 8: pk[0][e] = rr_pack_hi(tmp[0], tmp[3]); break;
            case 9: pk[1][e] = rr_pack_hi(tmp[2], tmp[5]); break;
            default: pk[2][e] = rr_pack_hi(tmp[1], tmp[4]); break;
        }
    };
    auto read_q = [&](int pl, int t, int kk) {
        if (DBG & 2) kk = 0, pl = 0;
        return __builtin_bit_cast(bf16x8, qs[qlane[pl] + 32 * t * RR_X3W_QSTRIDE + 2 * kk]);
    };
    {   // prologue: K-step 0 of the first segment
        rr_x3w_to_mfma_lanes(a[0], a[1], lo, hi);
        af = A_BF16 ? rr_x3_split<true>(lo, u32x4{}) : rr_x3_split<false>(lo, hi);
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            qf[0][t] = read_q(0, t, 0);
            qf[1][t] = read_q(1, t, 0);
            qf[2][t] = read_q(2, t, 0);
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
                const int s2 = (s + 1) % STEPS;                           // the K-step being prepared
                const int kk2 = A_BF16 ? s2 : (12 * p + s + 1) % 24;      // ... as K-step of the row
                const bool swap = A_BF16 ? (s2 % 2 == 0) : true;          // it starts a new pair of the ring
                const int np = A_BF16 ? s2 / 2 : s2;
                constexpr int REFILL_SLOT = 3 / VPS + 1 < NSLOT ? 3 / VPS + 1 : NSLOT - 1;   // first slot after the lane swaps
                // Ring refill by halves, each as ONE burst of 12 loads right after the lane swap of the
                // half's last pair: a burst asks for 384 contiguous bytes of each of the 32 rows at once,
                // which keeps the DRAM pages open.  Re-loading pair by pair (a steady 22-24 loads in
                // flight, one pair per K-step) measured 4.7 TB/s for the bare load/wait skeleton; so did
                // spreading a half's loads over the MFMA slots of a K-step.
                // The wait: the half holding pair np must have landed (np = 6: second half of this
                // segment, np = 0: first half of the next).  Younger in the queue: the other half's burst
                // (12) and, in an M-tile's first segment, the NQ2 maxima stores of the previous M-tile,
                // which are issued right behind this segment's first burst (below) precisely so that
                // they are YOUNGER than what either wait of the segment needs: vmcnt retires in order,
                // a store is acknowledged microseconds after issue under this read load, and a wait
                // that covers a fresh store stalls the wave for that long (0.34 ms of a 2.8 ms scan
                // when the stores sat in front of the next wait).  By the next wait that does cover
                // them they are 13+ K-steps old.  (STORE: the fallback pass stores at once and waits
                // conservatively.)
                if (swap && (np == 6 || np == 0)) {
                    if (!STORE && p == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(12 + NQ2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 12; ++j) asm volatile("" : "+v"(a[(np / 6) * 12 + j]));   // uses stay below the wait
                }
#pragma unroll
                for (int j = 0; j < NSLOT; ++j) {
                    const int i = j / NQ2, t = j % NQ2;
                    if (DBG & 4) {
                        asm volatile("" :: "v"(af.a1), "v"(af.a2), "v"(af.a3));   // (no MFMA: keep the operands alive)
                    } else {
                        const bf16x8 am = A_BF16 ? af.a1 : RR_X3W_A_OF(i, af);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qf[RR_X3W_PL_OF(i, A_BF16)][t], acc[t], 0, 0, 0);
                    }
                    if (swap) {
#pragma unroll
                        for (int k = VPS * j; k < VPS * j + VPS; ++k) valu_op(k, a[2 * np], a[2 * np + 1]);
                    }
                    if (swap && (np == 5 || np == 11) && j == REFILL_SLOT) {
#pragma unroll
                        for (int u = 0; u < 12; u += 2) RR_X3W_LOAD(a[(np / 6) * 12 + u], (np / 6) * 12 + u);          // rows 0-15: 384 B each
#pragma unroll
                        for (int u = 1; u < 12; u += 2) RR_X3W_LOAD(a[(np / 6) * 12 + u], (np / 6) * 12 + u);          // rows 16-31
                        if (!STORE && p == 0 && np == 5) {
                            // maxima of the previous M-tile ([32-row tile][query][2]: 256 contiguous bytes per
                            // store).  First M-tile of the wave: nothing pending yet -- the same count of stores
                            // goes to its own slot, overwritten one M-tile later by this wave.
                            const int64_t mprev = mt > m0 ? mt - 1 : mt;
#pragma unroll
                            for (int t = 0; t < NQ2; ++t)
                                if (h == 0) *reinterpret_cast<float2*>(gmax + ((mprev * QN + 32 * t + c) << 1)) = pend[t];
                        }
                    }
                    // B fragments of the next K-step, in place, each plane right after its last term here:
                    // term index after the last use of q3 | q1 | q2
                    constexpr int FREE3 = 1, FREE1 = A_BF16 ? 2 : 4, FREE2 = A_BF16 ? 3 : 6;
                    if ((DBG & 2) && (mt != m0 || p != 0 || s != 0)) { __builtin_amdgcn_sched_barrier(0); continue; }
                    if (j >= FREE3 * NQ2 && j < FREE3 * NQ2 + NQ2) qf[2][j - FREE3 * NQ2] = read_q(2, j - FREE3 * NQ2, kk2);
                    if (j >= FREE1 * NQ2 && j < FREE1 * NQ2 + NQ2) qf[0][j - FREE1 * NQ2] = read_q(0, j - FREE1 * NQ2, kk2);
                    if (FREE2 * NQ2 < NSLOT && j >= FREE2 * NQ2 && j < FREE2 * NQ2 + NQ2) qf[1][j - FREE2 * NQ2] = read_q(1, j - FREE2 * NQ2, kk2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(DBG & 2)) {
#pragma unroll
                    for (int t = 0; t < NQ2; ++t) qf[1][t] = read_q(1, t, kk2);   // q2 closes the K-step: re-loaded here,
                }                                                                //  first needed NQ2 * (A_BF16 ? 2 : 4) MFMAs on
                if (A_BF16) {
                    af.a1 = __builtin_bit_cast(bf16x8, (s2 % 2 == 0) ? lo : hi);
                } else {
                    af.a1 = __builtin_bit_cast(bf16x8, pk[0]);
                    af.a2 = __builtin_bit_cast(bf16x8, pk[1]);
                    af.a3 = __builtin_bit_cast(bf16x8, pk[2]);
                }
            }
        }
        if (DBG & 64) {                                   // (ablation: no epilogue, the accumulators stay alive)
#pragma unroll
            for (int e = 0; e < 16; ++e) asm volatile("" :: "v"(acc[0][e]), "v"(acc[NQ2 - 1][e]));
            continue;
        }
        // lane (c, h), register 4g + i: row 8g + 4h + i of the M-tile, query 32t + c
        long long e0 = 0;
        if (DBG & 16) e0 = clock64();
        const int64_t rbase = mt * 32 + 4 * h;
        const bool full = mt * 32 + 32 <= G.n_rows;
        const bool tile_end = (mt & 1) == 1;
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            float m16[2] = {-INFINITY, -INFINITY};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                if (STORE || !full) v = rr_x3_canon(v, rbase + 8 * g, G.n_rows);   // (fmaxf drops a NaN by itself)
                m16[g >> 1] = fmaxf(m16[g >> 1], fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                if (STORE)
                    *reinterpret_cast<f32x4*>(sims + (((mt * 2 + (g >> 1)) * QN + 32 * t + c) * 16 + 8 * (g & 1) + 4 * h)) = v;
            }
            {   // the other k half's rows: lanes l and l ^ 32 (one v_permlane32_swap each, no LDS round trip)
                const auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m16[0]), __float_as_uint(m16[0]), false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m16[1]), __float_as_uint(m16[1]), false, false);
                m16[0] = fmaxf(__uint_as_float(r0[0]), __uint_as_float(r0[1]));
                m16[1] = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
            }
            gm[t] = fmaxf(gm[t], fmaxf(m16[0], m16[1]));
            if (STORE) {             // rr_select's layout: one maximum per 64-row tile
                tmax[t] = fmaxf(tmax[t], fmaxf(m16[0], m16[1]));
                if (tile_end) {
                    if (h == 0) gmax[(mt >> 1) * QN + 32 * t + c] = tmax[t];
                    tmax[t] = -INFINITY;
                }
            } else {                 // rr_select_mtiles' layout (mm_pairs); stored during the next M-tile
                pend[t] = float2{m16[0], m16[1]};
            }
        }
        if (DBG & 16) dbg_epi += clock64() - e0;
    }
    if ((DBG & 16) && lane == 0) {
        long long* o = reinterpret_cast<long long*>(sims) + wave * 4;
        o[0] = clock64() - dbg_t0;
        o[1] = dbg_wait;
        o[2] = dbg_epi;
        o[3] = m1 - m0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (redundant) loads
    if (h == 0) {
#pragma unroll
        for (int t = 0; t < NQ2; ++t) {
            if (!STORE) *reinterpret_cast<float2*>(gmax + (((m1 - 1) * QN + 32 * t + c) << 1)) = pend[t];
            smax[wave * QN + 32 * t + c] = rr_f2key(gm[t]);
        }
    }
}

// Recomputes the 16 scores of every 16-row M-tile rr_select_mtiles listed, one wave per (query,
// M-tile): the tile's rows sit in their own half of the 32-row MFMA tile (the other half is
// zeros), the query's planes fill every B column, and split, term order and K-step order are the
// scan's, so each score is the scan's bit for bit.  Output: sc[query][slot][16].
template <bool A_BF16>
__global__ __launch_bounds__(256, 4) void rr_rescore_x3w(
    const u32x4* __restrict__ mat, int64_t n_rows, const u32x4* __restrict__ planes, int QN,
    const uint32_t* __restrict__ mtiles, const int32_t* __restrict__ count, const int32_t* __restrict__ fb,
    float* __restrict__ sc) {
    constexpr int ROWU = A_BF16 ? 48 : 96;
    constexpr int SEGS = A_BF16 ? 1 : 2;
    constexpr int NTERM = A_BF16 ? 3 : 6;
    const int q = blockIdx.y;
    if (fb[q]) return;
    const int n = count[q];
    if ((int)blockIdx.x * 4 >= n) return;                  // (whole workgroup: no barrier is skipped by a part of it)
    __shared__ u32x4 qs[3 * RR_X3_UNITS];                  // the query's three planes
    if (threadIdx.x < 3 * RR_X3_UNITS)
        qs[threadIdx.x] = planes[((threadIdx.x / RR_X3_UNITS) * QN + q) * RR_X3_UNITS + threadIdx.x % RR_X3_UNITS];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int lrow = lane & 15, lpc = lane >> 4;
    for (int slot = blockIdx.x * 4 + (threadIdx.x >> 6); slot < n; slot += gridDim.x * 4) {
        const int64_t m16 = mtiles[(int64_t)q * RR_X3_MCAP + slot];
        const int half = (int)(m16 & 1);                   // which 16 rows of the 32-row M-tile
        int64_t row = m16 * 16 + lrow;
        row = row < n_rows ? row : n_rows - 1;
        const u32x4* p = mat + row * ROWU + lpc;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 1
        for (int seg = 0; seg < 2 * SEGS; ++seg) {          // six 64-byte groups of the row at a time
            u32x4 u[6];
#pragma unroll
            for (int pair = 0; pair < 6; ++pair) u[pair] = p[seg * 24 + 4 * pair];
#pragma unroll
            for (int pair = 0; pair < 6; ++pair) {
                const u32x4 zero = {0u, 0u, 0u, 0u};
                u32x4 lo, hi;
                rr_x3w_to_mfma_lanes(half ? zero : u[pair], half ? u[pair] : zero, lo, hi);
                constexpr int KSTEPS = A_BF16 ? 2 : 1;
#pragma unroll
                for (int v = 0; v < KSTEPS; ++v) {
                    const int kk = (A_BF16 ? 2 : 1) * (6 * seg + pair) + v;      // K-step of the row, 0..23
                    bf16x8 qf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) qf[pl] = __builtin_bit_cast(bf16x8, qs[pl * RR_X3_UNITS + 2 * kk + h]);
                    const rr_x3_afrag af = A_BF16 ? rr_x3_split<true>(v == 0 ? lo : hi, u32x4{})
                                                  : rr_x3_split<false>(lo, hi);
#pragma unroll
                    for (int i = 0; i < NTERM; ++i) rr_x3w_term<A_BF16>(i, af, qf, acc);
                }
            }
        }
        if ((lane & 31) == 0) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                // register 4g + i with g = 2 * half + g2: row 8 * g2 + 4h + i of the 16-row M-tile
                f32x4 v;
                if (half) v = f32x4{acc[8 + 4 * g2], acc[9 + 4 * g2], acc[10 + 4 * g2], acc[11 + 4 * g2]};
                else v = f32x4{acc[4 * g2], acc[1 + 4 * g2], acc[2 + 4 * g2], acc[3 + 4 * g2]};
                v = rr_x3_canon(v, m16 * 16 + 8 * g2 + 4 * h, n_rows);
                *reinterpret_cast<f32x4*>(sc + ((int64_t)q * RR_X3_MCAP + slot) * 16 + 8 * g2 + 4 * h) = v;
            }
        }
    }
}

template <int NQ2, bool A_BF16>
static int rr_dense_chunk_x3w_t(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                                float* d_scores, hipStream_t st) {
    constexpr int THREADS = NQ2 == 2 ? 512 : 256;
    constexpr int QN = 32 * NQ2;
    static int waves = 0;
    if (!waves) waves = rr_resident_waves((const void*)rr_scan_x3w<NQ2, A_BF16, false>, THREADS, ix->device);
    rr_scan_geom G = rr_make_geom(ix, waves / 4);
    G.qs = QN;
    G.mm_pairs = 1;
    unsigned short* planes = reinterpret_cast<unsigned short*>(ix->d_qplanes);
    const u32x4* mat = reinterpret_cast<const u32x4*>(ix->d_matrix);
    const u32x4* pl4 = reinterpret_cast<const u32x4*>(planes);
    const dim3 grid((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), block(THREADS);
    const rr_x3_scratch X = rr_x3_scratch_of(ix);
    rr_launch_split_queries(d_q, planes, QN, A_BF16 ? RR_X3_ORDER_WIDE_BF16 : RR_X3_ORDER_NATURAL, st);
    if (rr_x3_stored_path(ix)) {
        const int slot = rr_scan_events_begin(ix, st);
    rr_scan_note(ix, 4, NQ2, nq, A_BF16 ? 3 : 6);
        hipLaunchKernelGGL((rr_scan_x3w<NQ2, A_BF16, true>), grid, block, 0, st, mat, G, pl4, ix->d_sims,
                           ix->d_gmax, ix->d_smax, (const int32_t*)nullptr, 0);
        rr_scan_events_end(ix, slot, st);
        rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st);
        RR_HIP_TRY(hipGetLastError());
        return RR_OK;
    }
    const int slot = rr_scan_events_begin(ix, st);
    rr_scan_note(ix, 4, NQ2, nq, A_BF16 ? 3 : 6);
    hipLaunchKernelGGL((rr_scan_x3w<NQ2, A_BF16, false>), grid, block, 0, st, mat, G, pl4, ix->d_sims,
                       ix->d_gmax, ix->d_smax, (const int32_t*)nullptr, 0);
    rr_scan_events_end(ix, slot, st);
    rr_launch_select_mtiles(ix, G, nq, pool, st);
    hipLaunchKernelGGL((rr_rescore_x3w<A_BF16>), dim3(64, nq), dim3(256), 0, st, mat, G.n_rows, pl4, QN,
                       X.mtiles, X.count, X.fb, X.sc);
    rr_launch_select_rescored(ix, G, nq, pool, d_rows, d_scores, st);
    // Fallback for the queries that raised their flag: both launches return at once otherwise.
    hipLaunchKernelGGL((rr_scan_x3w<NQ2, A_BF16, true>), grid, block, 0, st, mat, G, pl4, ix->d_sims,
                       ix->d_gmax, ix->d_smax, (const int32_t*)X.fb, nq);
    rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st, X.fb);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

template <int NQ2, bool A_BF16>
static int rr_x3w_fallback_t(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                             float* d_scores, const int32_t* flags, hipStream_t st) {
    constexpr int THREADS = NQ2 == 2 ? 512 : 256;
    constexpr int QN = 32 * NQ2;
    static int waves = 0;
    if (!waves) waves = rr_resident_waves((const void*)rr_scan_x3w<NQ2, A_BF16, true>, THREADS, ix->device);
    rr_scan_geom G = rr_make_geom(ix, waves / 4);
    G.qs = QN;
    unsigned short* planes = reinterpret_cast<unsigned short*>(ix->d_qplanes);
    const dim3 grid((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), block(THREADS);
    hipLaunchKernelGGL((rr_scan_x3w<NQ2, A_BF16, true>), grid, block, 0, st, reinterpret_cast<const u32x4*>(ix->d_matrix), G,
                       reinterpret_cast<const u32x4*>(planes), ix->d_sims, ix->d_gmax, ix->d_smax, flags, nq, d_q);
    rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st, flags);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

// The fallback of a whole filter call (<= 256 queries) as TWO launches instead of two per 64-query block: slice y of the
// grids serves queries 64 y .. 64 y + 63 with its own score / maxima scratch (rr_ensure_scratch sizes d_sims for the call;
// d_gmax / d_smax hold four slices as they are).  Every slice returns at once when none of its flags is up.
int rr_dense_x3w_fallback_all(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows, float* d_scores,
                              const int32_t* flags, hipStream_t st) {
    const bool b = ix->dtype == RR_DTYPE_BF16;
    const int slices = (nq + RR_MFMA_MAXQ - 1) / RR_MFMA_MAXQ;
    RR_REQUIRE(slices >= 1 && slices <= 4, "rr_dense_x3w_fallback_all: %d queries", nq);
    if (ix->scratch_q < slices * RR_MFMA_MAXQ) {
        // one score slice only (no room for more: rr_ensure_scratch): the blocks of 64 queries one after the other
        for (int q0 = 0; q0 < nq; q0 += RR_MFMA_MAXQ) {
            const int n = nq - q0 < RR_MFMA_MAXQ ? nq - q0 : RR_MFMA_MAXQ;
            const int rc = rr_dense_chunk_x3w_fallback(ix, d_q + (int64_t)q0 * ix->dim_pad, n, pool, d_rows + (int64_t)q0 * pool,
                                                       d_scores + (int64_t)q0 * pool, flags + q0, st);
            if (rc != RR_OK) return rc;
        }
        return RR_OK;
    }
    static int waves[2] = {0, 0};
    if (!waves[b]) waves[b] = b ? rr_resident_waves((const void*)rr_scan_x3w<2, true, true>, 512, ix->device)
                                : rr_resident_waves((const void*)rr_scan_x3w<2, false, true>, 512, ix->device);
    rr_scan_geom G = rr_make_geom(ix, waves[b] / 4);
    G.qs = 64;
    const int64_t sims_slice = (int64_t)64 * G.n_pad, gmax_slice = (int64_t)64 * G.n_tiles, smax_slice = (int64_t)64 * RR_MAX_SCAN_WAVES;
    const dim3 grid((G.n_waves + 7) / 8, slices), block(512);
    const u32x4* planes = reinterpret_cast<const u32x4*>(ix->d_qplanes);          // (unused: the queries are split in the kernel)
    if (b)
        hipLaunchKernelGGL((rr_scan_x3w<2, true, true>), grid, block, 0, st, reinterpret_cast<const u32x4*>(ix->d_matrix), G, planes,
                           ix->d_sims, ix->d_gmax, ix->d_smax, flags, nq, d_q, sims_slice, gmax_slice, smax_slice);
    else
        hipLaunchKernelGGL((rr_scan_x3w<2, false, true>), grid, block, 0, st, reinterpret_cast<const u32x4*>(ix->d_matrix), G, planes,
                           ix->d_sims, ix->d_gmax, ix->d_smax, flags, nq, d_q, sims_slice, gmax_slice, smax_slice);
    rr_launch_select(ix, G, nq, pool, d_rows, d_scores, st, flags, slices, sims_slice, gmax_slice, smax_slice);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

int rr_dense_chunk_x3w_fallback(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                                float* d_scores, const int32_t* flags, hipStream_t st) {
    const bool b = ix->dtype == RR_DTYPE_BF16;
    if (nq <= 32)
        return b ? rr_x3w_fallback_t<1, true>(ix, d_q, nq, pool, d_rows, d_scores, flags, st)
                 : rr_x3w_fallback_t<1, false>(ix, d_q, nq, pool, d_rows, d_scores, flags, st);
    return b ? rr_x3w_fallback_t<2, true>(ix, d_q, nq, pool, d_rows, d_scores, flags, st)
             : rr_x3w_fallback_t<2, false>(ix, d_q, nq, pool, d_rows, d_scores, flags, st);
}

int rr_dense_chunk_x3w(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                       float* d_scores, hipStream_t st) {
    const bool b = ix->dtype == RR_DTYPE_BF16;
    if (nq <= 32)
        return b ? rr_dense_chunk_x3w_t<1, true>(ix, d_q, nq, pool, d_rows, d_scores, st)
                 : rr_dense_chunk_x3w_t<1, false>(ix, d_q, nq, pool, d_rows, d_scores, st);
    return b ? rr_dense_chunk_x3w_t<2, true>(ix, d_q, nq, pool, d_rows, d_scores, st)
             : rr_dense_chunk_x3w_t<2, false>(ix, d_q, nq, pool, d_rows, d_scores, st);
}

#ifdef RR_DEBUG_HARNESS
#include "rr_debug.h"
// Timing-only ablations of the fp32 wide scan at 64 queries (tools/x3w_ablate.py): runs variant `dbg`
// `reps` times and returns the mean kernel time.  The scores it leaves behind are garbage.
template <int DBG>
static float rr_debug_time_x3w(rr_index* ix, hipStream_t st, int reps) {
    constexpr int THREADS = (DBG & 32) ? 256 : 512, QN = 64;
    const int waves = rr_resident_waves((const void*)rr_scan_x3w<2, false, false, DBG>, THREADS, ix->device);
    rr_scan_geom G = rr_make_geom(ix, waves / 4);
    G.qs = QN;
    G.mm_pairs = 1;
    const dim3 grid((G.n_waves + THREADS / 64 - 1) / (THREADS / 64)), block(THREADS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float total = 0.f;
    for (int r = 0; r < reps + 1; ++r) {
        hipEventRecord(e0, st);
        hipLaunchKernelGGL((rr_scan_x3w<2, false, false, DBG>), grid, block, 0, st,
                           reinterpret_cast<const u32x4*>(ix->d_matrix), G, reinterpret_cast<const u32x4*>(ix->d_qplanes),
                           ix->d_sims, ix->d_gmax, ix->d_smax, (const int32_t*)nullptr, 0);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (r) total += ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (DBG & 16) {
        std::vector<long long> hst((size_t)G.n_waves * 4);
        hipMemcpy(hst.data(), ix->d_sims, hst.size() * 8, hipMemcpyDeviceToHost);
        double tot = 0, wt = 0, ep = 0, mtl = 0, mx = 0;
        for (int w = 0; w < G.n_waves; ++w) {
            tot += hst[4 * w]; wt += hst[4 * w + 1]; ep += hst[4 * w + 2]; mtl += hst[4 * w + 3];
            if (hst[4 * w] > mx) mx = (double)hst[4 * w];
        }
        printf("stamps: %d waves, mean wave cycles %.0f (max %.0f), in ring waits %.1f %%, in epilogues %.1f %%, "
               "cycles per 32-row M-tile %.0f\n", G.n_waves, tot / G.n_waves, mx, 100 * wt / tot, 100 * ep / tot, tot / mtl);
        fflush(stdout);
    }
    return total / reps;
}

extern "C" int rr_debug_scan_x3w(rr_index* ix, int32_t dbg, int32_t reps, float* out_ms) {
    RR_REQUIRE(ix && out_ms && ix->dtype == RR_DTYPE_F32 && ix->dim_pad == 384, "fp32 index of dim 384 expected");
    RR_REQUIRE(ix->scratch_q >= 64, "run a 64-query search first (allocates the scratch)");
    std::lock_guard<std::mutex> lock(ix->mu);
    hipStream_t st = nullptr;
    switch (dbg) {
        case 0: *out_ms = rr_debug_time_x3w<0>(ix, st, reps); break;
        case 1: *out_ms = rr_debug_time_x3w<1>(ix, st, reps); break;
        case 2: *out_ms = rr_debug_time_x3w<2>(ix, st, reps); break;
        case 3: *out_ms = rr_debug_time_x3w<3>(ix, st, reps); break;
        case 9: *out_ms = rr_debug_time_x3w<9>(ix, st, reps); break;
        case 11: *out_ms = rr_debug_time_x3w<11>(ix, st, reps); break;
        case 16: *out_ms = rr_debug_time_x3w<16>(ix, st, reps); break;
        case 15: *out_ms = rr_debug_time_x3w<15>(ix, st, reps); break;
        case 47: *out_ms = rr_debug_time_x3w<47>(ix, st, reps); break;
        case 79: *out_ms = rr_debug_time_x3w<79>(ix, st, reps); break;
        case 143: *out_ms = rr_debug_time_x3w<143>(ix, st, reps); break;
        case 207: *out_ms = rr_debug_time_x3w<207>(ix, st, reps); break;
        case 64: *out_ms = rr_debug_time_x3w<64>(ix, st, reps); break;
        case 32: *out_ms = rr_debug_time_x3w<32>(ix, st, reps); break;
        case 75: *out_ms = rr_debug_time_x3w<75>(ix, st, reps); break;
        case 65: *out_ms = rr_debug_time_x3w<65>(ix, st, reps); break;
        case 66: *out_ms = rr_debug_time_x3w<66>(ix, st, reps); break;
        case 48: *out_ms = rr_debug_time_x3w<48>(ix, st, reps); break;
        case 59: *out_ms = rr_debug_time_x3w<59>(ix, st, reps); break;
        default: RR_REQUIRE(false, "unknown ablation %d", dbg);
    }
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}
#endif  // RR_DEBUG_HARNESS
