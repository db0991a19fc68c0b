// rr_dense.hip -- K1: dense dot-product scan + exact top-pool selection (gfx950).
//
// Replaces cosine_similarity_search (utils.py:111-124), _cosine_pool
// (app/app_product_search.py:192-195) and cosine_search (app/test.py:125-132):
//   sims = M @ q ; argpartition(-sims, pool-1)[:pool] ; argsort.
//
// Data layout in HBM
//   M      n_rows x dim_pad fp32, row-major, dim_pad % 64 == 0 (zero padded)
//   sims   [query][n_pad]   fp32, n_pad = 64 * n_tiles         (4 B / row / query)
//   gmax   [query][n_tiles] fp32, max of each 64-row tile       (1/16 B / row / query)
//
// rr_scan_f32: HBM-bound streaming kernel.  A 16-lane DPP row owns one matrix row:
// lane j reads float4 j, j+16, ... of the row (NF = dim_pad/64 loads, each wave
// instruction covers four rows x 256 contiguous bytes) and runs one fp32 fmaf
// chain over its 4*NF elements; the 16 partials are added by rr_row16_sum.
// Every row therefore has the same summation order wherever it sits, so a
// score does not depend on sharding, batch size or launch geometry.
// A wave walks a 64-row tile in 16 steps of 4 rows, double-buffered in
// registers (2 x NF loads of 16 B per lane in flight), keeps row (4*it+grp) in
// lane (it + 16*grp), and ends the tile with one coalesced 256-B store of the
// 64 scores plus their maximum.
//
// rr_select: one 1024-thread workgroup per query.  The pool-th largest tile
// maximum is a lower bound for the pool-th largest score, so only tiles whose
// maximum reaches it are opened (~pool tiles x 256 B); the survivors are
// ordered by the 64-bit key (score desc, row asc) in LDS.  Degenerate inputs
// (massive ties, clustered rows) fall back to an 8-pass radix select over the
// same key, still exact.
#include "rr_common.h"

#define RR_SCAN_THREADS 256
#define RR_SEL_THREADS 1024
#define RR_SEL_GCAP 4096   // tiles kept in LDS
#define RR_SEL_CCAP 8192   // candidate rows kept in LDS

// ------------------------------------------------------------------ scan
template <int NF>
__device__ __forceinline__ void rr_load_rows(f32x4 (&dst)[NF], const f32x4* __restrict__ mat,
                                             int64_t row, int64_t n_rows, int sub) {
    row = row < n_rows ? row : n_rows - 1;  // tail rows re-read the last row; masked later
    const f32x4* p = mat + row * (int64_t)(NF * 16) + sub;
#pragma unroll
    for (int i = 0; i < NF; ++i) dst[i] = __builtin_nontemporal_load(p + 16 * i);
}

template <int NF, int NB>
__global__ __launch_bounds__(RR_SCAN_THREADS, (NB <= 1 ? 4 : 2)) void rr_scan_f32(
    const f32x4* __restrict__ mat, int64_t n_rows, int64_t n_tiles,
    const float* __restrict__ queries,  // NB x (NF*64)
    float* __restrict__ sims, int64_t sims_stride, float* __restrict__ gmax, int64_t gmax_stride) {
    __shared__ f32x4 qs[NB][NF * 16];
    const int tid = threadIdx.x;
    for (int i = tid; i < NB * NF * 16; i += RR_SCAN_THREADS)
        qs[i / (NF * 16)][i % (NF * 16)] = reinterpret_cast<const f32x4*>(queries)[i];
    __syncthreads();

    const int lane = tid & 63;
    const int sub = lane & 15;
    const int grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (RR_SCAN_THREADS / 64) + (tid >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (RR_SCAN_THREADS / 64);

    // NB == 1 keeps the query slice in registers; larger tiles re-read LDS
    // (conflict-free: the four rows of a wave broadcast the same 16 float4).
    f32x4 qreg[NB == 1 ? NF : 1];
    if (NB == 1) {
#pragma unroll
        for (int i = 0; i < NF; ++i) qreg[i] = qs[0][16 * i + sub];
    }

    f32x4 bufA[NF], bufB[NF];
    int64_t tile = wave;
    if (tile < n_tiles) rr_load_rows<NF>(bufA, mat, tile * 64 + grp, n_rows, sub);

    while (tile < n_tiles) {
        const int64_t next = tile + n_waves;
        const int64_t row0 = tile * 64;
        float mine[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) mine[b] = 0.f;

#pragma unroll(NB == 1 ? 8 : 1)
        for (int it = 0; it < 16; it += 2) {
            rr_load_rows<NF>(bufB, mat, row0 + 4 * (it + 1) + grp, n_rows, sub);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (NB > 1) asm volatile("" ::: "memory");  // keep one query slice live, re-read LDS
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const f32x4 q = (NB == 1) ? qreg[i] : qs[b][16 * i + sub];
                    acc = __builtin_fmaf(bufA[i].x, q.x, acc);
                    acc = __builtin_fmaf(bufA[i].y, q.y, acc);
                    acc = __builtin_fmaf(bufA[i].z, q.z, acc);
                    acc = __builtin_fmaf(bufA[i].w, q.w, acc);
                }
                acc = rr_row16_sum(acc);
                mine[b] = (sub == it) ? acc : mine[b];
            }
            if (it + 2 < 16)
                rr_load_rows<NF>(bufA, mat, row0 + 4 * (it + 2) + grp, n_rows, sub);
            else
                rr_load_rows<NF>(bufA, mat, (next < n_tiles ? next : tile) * 64 + grp, n_rows, sub);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (NB > 1) asm volatile("" ::: "memory");  // keep one query slice live, re-read LDS
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const f32x4 q = (NB == 1) ? qreg[i] : qs[b][16 * i + sub];
                    acc = __builtin_fmaf(bufB[i].x, q.x, acc);
                    acc = __builtin_fmaf(bufB[i].y, q.y, acc);
                    acc = __builtin_fmaf(bufB[i].z, q.z, acc);
                    acc = __builtin_fmaf(bufB[i].w, q.w, acc);
                }
                acc = rr_row16_sum(acc);
                mine[b] = (sub == it + 1) ? acc : mine[b];
            }
        }
        // lane (sub, grp) now holds row row0 + 4*sub + grp for every query.
        const int64_t my_row = row0 + 4 * sub + grp;
        const bool valid = my_row < n_rows;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v = mine[b];
            v = (valid && v == v) ? v : -INFINITY;  // NaN scores and pad rows rank last
            sims[(int64_t)b * sims_stride + my_row] = v;
            const float m = rr_wave_max(v);
            if (lane == 0) gmax[(int64_t)b * gmax_stride + tile] = m;
        }
        tile = next;
    }
}

// Generic-dimension variant (runtime NF); same per-row summation order.
template <int NB>
__global__ __launch_bounds__(RR_SCAN_THREADS) void rr_scan_f32_generic(
    const f32x4* __restrict__ mat, int64_t n_rows, int64_t n_tiles, int nf,
    const float* __restrict__ queries, float* __restrict__ sims, int64_t sims_stride,
    float* __restrict__ gmax, int64_t gmax_stride) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = lane & 15;
    const int grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (RR_SCAN_THREADS / 64) + (tid >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (RR_SCAN_THREADS / 64);
    const f32x4* q4 = reinterpret_cast<const f32x4*>(queries);
    for (int64_t tile = wave; tile < n_tiles; tile += n_waves) {
        const int64_t row0 = tile * 64;
        float mine[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) mine[b] = 0.f;
        for (int it = 0; it < 16; ++it) {
            int64_t row = row0 + 4 * it + grp;
            row = row < n_rows ? row : n_rows - 1;
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
                const float s = rr_row16_sum(acc[b]);
                mine[b] = (sub == it) ? s : mine[b];
            }
        }
        const int64_t my_row = row0 + 4 * sub + grp;
        const bool valid = my_row < n_rows;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v = mine[b];
            v = (valid && v == v) ? v : -INFINITY;
            sims[(int64_t)b * sims_stride + my_row] = v;
            const float m = rr_wave_max(v);
            if (lane == 0) gmax[(int64_t)b * gmax_stride + tile] = m;
        }
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

__global__ __launch_bounds__(RR_SEL_THREADS) void rr_select(
    const float* __restrict__ sims, int64_t sims_stride, const float* __restrict__ gmax,
    int64_t gmax_stride, int64_t n_rows, int64_t n_tiles, int pool, int64_t row_offset,
    int64_t* __restrict__ out_rows, float* __restrict__ out_scores) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t sel[2];
    __shared__ uint32_t counters[2];
    __shared__ uint32_t glist[RR_SEL_GCAP];
    __shared__ uint64_t cand[RR_SEL_CCAP];

    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const float* s = sims + (int64_t)q * sims_stride;
    const float* g = gmax + (int64_t)q * gmax_stride;

    // ---- 1. tau = pool-th largest tile maximum (or "everything" if few tiles)
    uint32_t tau_key = 0;
    if (n_tiles > pool) {
        uint32_t prefix = 0, mask = 0, k = (uint32_t)pool;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int64_t i = tid; i < n_tiles; i += RR_SEL_THREADS) {
                const uint32_t key = rr_f2key(g[i]);
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
            if (rr_f2key(g[i]) >= tau_key) {
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
            const uint32_t key = rr_f2key(s[row]);
            if (key >= tau_key) {
                const uint32_t slot = atomicAdd(&counters[1], 1u);
                if (slot < RR_SEL_CCAP)
                    cand[slot] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)row);
            }
        }
    }
    __syncthreads();
    uint32_t n_cand = counters[1];

    if (n_cand > RR_SEL_CCAP) {
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
                    const uint64_t key = ((uint64_t)rr_f2key(s[row]) << 32) |
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
                const uint64_t key = ((uint64_t)rr_f2key(s[row]) << 32) |
                                     (uint64_t)(0xFFFFFFFFu - (uint32_t)row);
                if (key >= prefix) cand[atomicAdd(&counters[1], 1u)] = key;
            }
        }
        __syncthreads();
        n_cand = counters[1];  // == pool
    }

    // ---- 4. order the survivors (score desc, row asc) and emit the first pool
    int n_sort = 1;
    while (n_sort < (int)n_cand) n_sort <<= 1;
    for (int i = tid; i < n_sort; i += RR_SEL_THREADS)
        if (i >= (int)n_cand) cand[i] = 0;
    rr_bitonic_desc(cand, n_sort);
    for (int i = tid; i < pool; i += RR_SEL_THREADS) {
        const uint64_t key = cand[i];
        const uint32_t row = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
        out_rows[(int64_t)q * pool + i] = (int64_t)row + row_offset;
        out_scores[(int64_t)q * pool + i] = rr_key2f((uint32_t)(key >> 32));
    }
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
static int rr_ensure_scratch(rr_index* ix, int nq) {
    if (ix->scratch_q >= nq) return RR_OK;
    const int64_t n_tiles = rr_round_up(ix->n_rows, 64) / 64;
    if (ix->d_sims) hipFree(ix->d_sims);
    if (ix->d_gmax) hipFree(ix->d_gmax);
    ix->d_sims = nullptr;
    ix->d_gmax = nullptr;
    ix->scratch_q = 0;
    RR_HIP_TRY(hipMalloc(&ix->d_sims, sizeof(float) * (size_t)nq * n_tiles * 64));
    RR_HIP_TRY(hipMalloc(&ix->d_gmax, sizeof(float) * (size_t)nq * n_tiles));
    ix->scratch_q = nq;
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
    return per_cu * cus;
}

template <int NB>
static void rr_launch_scan(rr_index* ix, const float* d_q, hipStream_t st, int grid) {
    const int64_t n_tiles = rr_round_up(ix->n_rows, 64) / 64;
    const int64_t n_pad = n_tiles * 64;
    const f32x4* mat = reinterpret_cast<const f32x4*>(ix->d_matrix);
    const int nf = ix->dim_pad / 64;
    static int cap6 = 0, capg = 0;   // per template instance
    if (nf == 6) {
        if (!cap6) cap6 = rr_resident_grid(rr_scan_f32<6, NB>, ix->device);
        if (grid > cap6) grid = cap6;
    } else {
        if (!capg) capg = rr_resident_grid(rr_scan_f32_generic<NB>, ix->device);
        if (grid > capg) grid = capg;
    }
    if (nf == 6)
        hipLaunchKernelGGL((rr_scan_f32<6, NB>), dim3(grid), dim3(RR_SCAN_THREADS), 0, st, mat,
                           ix->n_rows, n_tiles, d_q, ix->d_sims, n_pad, ix->d_gmax, n_tiles);
    else
        hipLaunchKernelGGL((rr_scan_f32_generic<NB>), dim3(grid), dim3(RR_SCAN_THREADS), 0, st, mat,
                           ix->n_rows, n_tiles, nf, d_q, ix->d_sims, n_pad, ix->d_gmax, n_tiles);
}

// Scan + select for up to 8 queries already on the device (padded to dim_pad).
static int rr_dense_chunk(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                          float* d_scores, hipStream_t st, bool time_it) {
    const int64_t n_tiles = rr_round_up(ix->n_rows, 64) / 64;
    const int64_t n_pad = n_tiles * 64;
    int grid = (int)((n_tiles + 3) / 4);  // capped to the resident grid in rr_launch_scan
    if (time_it) hipEventRecord(ix->ev0, st);
    const int slot = (int)(ix->ring_head % rr_index::kRing);
    hipEventRecord(ix->ring0[slot], st);
    switch (nq) {
        case 1: rr_launch_scan<1>(ix, d_q, st, grid); break;
        case 2: rr_launch_scan<2>(ix, d_q, st, grid); break;
        case 3: case 4: rr_launch_scan<4>(ix, d_q, st, grid); break;
        default: rr_launch_scan<8>(ix, d_q, st, grid); break;
    }
    hipEventRecord(ix->ring1[slot], st);
    ix->ring_head++;
    if (ix->ring_head - ix->ring_tail > rr_index::kRing) ix->ring_tail = ix->ring_head - rr_index::kRing;
    if (time_it) {
        hipEventRecord(ix->ev1, st);
        ix->timing_valid = true;
    }
    hipLaunchKernelGGL(rr_select, dim3(nq), dim3(RR_SEL_THREADS), 0, st, ix->d_sims, n_pad,
                       ix->d_gmax, n_tiles, ix->n_rows, n_tiles, pool, ix->row_offset, d_rows,
                       d_scores);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

static int rr_dense_topk_impl(rr_index* ix, const float* d_q_padded, int nq, int pool,
                              int64_t* d_rows, float* d_scores, hipStream_t st) {
    RR_HIP_TRY(hipSetDevice(ix->device));
    int rc = rr_ensure_scratch(ix, 8);
    if (rc) return rc;
    for (int q0 = 0; q0 < nq; q0 += 8) {
        int n = nq - q0 < 8 ? nq - q0 : 8;
        // the scan kernels read NB = 1/2/4/8 query slots; slots past n hold zeros
        rc = rr_dense_chunk(ix, d_q_padded + (int64_t)q0 * ix->dim_pad, n, pool,
                            d_rows + (int64_t)q0 * pool, d_scores + (int64_t)q0 * pool, st,
                            q0 == 0);
        if (rc) return rc;
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

extern "C" int rr_dense_topk_dev(rr_index* ix, const float* d_queries, int32_t n_queries,
                                 int32_t pool, int64_t* d_out_rows, float* d_out_scores,
                                 void* stream) {
    RR_REQUIRE(ix && d_queries && d_out_rows && d_out_scores, "rr_dense_topk_dev: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH, "rr_dense_topk_dev: n_queries %d out of [1,%d]",
               n_queries, RR_MAX_BATCH);
    RR_REQUIRE(pool >= 1 && pool <= RR_MAX_POOL && pool <= ix->n_rows,
               "rr_dense_topk_dev: pool %d out of [1,min(%d,n_rows=%lld)]", pool, RR_MAX_POOL,
               (long long)ix->n_rows);
    RR_REQUIRE(ix->d_matrix, "rr_dense_topk_dev: index has no matrix");
    RR_REQUIRE(ix->dtype == RR_DTYPE_F32, "rr_dense_topk_dev: only fp32 storage is built");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream
    const int slots = (int)rr_round_up(n_queries, 8);
    const int64_t total = (int64_t)slots * ix->dim_pad;
    hipLaunchKernelGGL(rr_pad_queries, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       d_queries, ix->d_q, n_queries, ix->dim, ix->dim_pad, slots);
    return rr_dense_topk_impl(ix, ix->d_q, n_queries, pool, d_out_rows, d_out_scores, st);
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
    RR_REQUIRE(ix->dtype == RR_DTYPE_F32, "rr_dense_topk: only fp32 storage is built");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = ix->stream;
    const int slots = (int)rr_round_up(n_queries, 8);
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
    return RR_OK;
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

extern "C" int rr_index_l2_normalize(rr_index* ix, float eps) {
    RR_REQUIRE(ix && ix->d_matrix, "rr_index_l2_normalize: index has no matrix");
    RR_REQUIRE(ix->dtype == RR_DTYPE_F32, "rr_index_l2_normalize: only fp32 storage is built");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    const int rows_per_block = 4;
    const unsigned grid = (unsigned)((ix->n_rows + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(rr_l2norm_f32, dim3(grid), dim3(64 * rows_per_block), 0, ix->stream,
                       (float*)ix->d_matrix, ix->n_rows, ix->dim_pad, eps);
    RR_HIP_TRY(hipGetLastError());
    RR_HIP_TRY(hipStreamSynchronize(ix->stream));
    return RR_OK;
}
