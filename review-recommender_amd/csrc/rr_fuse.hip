// rr_fuse.hip -- K3: shard merge + min-max / priors / trust / blend / gate + top-k (gfx950).
//
// Replaces everything run_search does once the candidate pool exists
// (app/app_product_search.py:256-312) and the CLI's variant (app/test.py:250-309).
// Latency-bound: one 256-thread workgroup per query, all columns in LDS.
//
// The kernel reproduces numpy's evaluation order and dtypes, not just its maths
// (built with -ffp-contract=off; SURVEY section 7 "Mixed precision in the blend"):
//   _minmax            lo/hi as Python floats; float32 input is scaled in float32 with the
//                      divisor float32(hi - lo + 1e-12), float64 input in float64; any
//                      non-finite lo/hi or hi - lo < 1e-12 gives zeros       (:182-187)
//   _bayes_prior       float64, g = nanmean over the POOL (numpy pairwise sum)   (:197-199)
//   prior_volume       log1p(n) / (max(log1p(n)) + 1e-9), float64                (:267)
//   _prior             float64(float32(minmax * float32(0.7))) + 0.3 * volume    (:268)
//   _trust             float32(0.6*clip(n/max(min_reviews,1),0,1) + 0.4*min(1, log1p(n)/log1p(sat)))
//   final              float32 products w*col for float32 columns, float32 sum of the dense
//                      and bm25 terms, float64 from the first float64 term on, one cast to
//                      float32, then * trust * gate in float32                  (:306-309)
//   order              final desc, ties by pool position (the reference's quicksort leaves
//                      tie order unspecified), NaN last; first k                (:312)
#include "rr_common.h"

#define RR_FUSE_THREADS 256
#define RR_FUSE_MAXCAND 4096

struct rr_fuse_dev_params {
    rr_fuse_params p;
    double log1p_sat;     // np.log1p(max(trust_sat, 1)) computed on the host
    float w_dense32, w_bm2532, w_rerank32, w_best32;
    int64_t row_offset, n_rows;
    int32_t key_cap, col_cap;   // LDS capacities of THIS launch: sort keys (power of two >= n_candidates and pool), pool columns
};

// numpy's pairwise float64 sum (np.add.reduce on a contiguous array).
__device__ double rr_np_pairwise(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
            r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
            r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
        }
        double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return rr_np_pairwise(a, n2) + rr_np_pairwise(a + n2, n - n2);
}

__device__ __forceinline__ bool rr_finite(double x) { return (x - x) == 0.0; }

// Candidate i of query q.  Contiguous input: [query][candidate].  Gathered shard payloads
// (one RCCL all-gather of per-rank blocks): [rank][query][per_rank], ranks `stride` bytes apart.
struct rr_cand_addr {
    int q, ncand, per_rank;
    int64_t stride_bytes;
    template <typename T>
    __device__ __forceinline__ T get(const T* p, int i) const {
        if (per_rank == 0) return p[(int64_t)q * ncand + i];
        const int r = i / per_rank, j = i - r * per_rank;
        return p[(int64_t)r * (stride_bytes / (int64_t)sizeof(T)) + (int64_t)q * per_rank + j];
    }
};

// Block-wide min / max / NaN flag over LDS columns (n <= RR_MAX_POOL).
template <typename T>
__device__ void rr_block_minmax(const T* x, int n, double* red, double& lo, double& hi, bool& bad) {
    const int tid = threadIdx.x;
    double mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    for (int i = tid; i < n; i += RR_FUSE_THREADS) {
        const double v = (double)x[i];
        if (v != v) nan = 1;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    red[tid] = mn;
    red[RR_FUSE_THREADS + tid] = mx;
    red[2 * RR_FUSE_THREADS + tid] = (double)nan;
    __syncthreads();
    for (int s = RR_FUSE_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] = red[tid] < red[tid + s] ? red[tid] : red[tid + s];
            red[RR_FUSE_THREADS + tid] = red[RR_FUSE_THREADS + tid] > red[RR_FUSE_THREADS + tid + s]
                                             ? red[RR_FUSE_THREADS + tid]
                                             : red[RR_FUSE_THREADS + tid + s];
            red[2 * RR_FUSE_THREADS + tid] += red[2 * RR_FUSE_THREADS + tid + s];
        }
        __syncthreads();
    }
    lo = red[0];
    hi = red[RR_FUSE_THREADS];
    bad = red[2 * RR_FUSE_THREADS] != 0.0;  // np.min / np.max propagate NaN
    __syncthreads();
}

// _minmax on a float32 column, in place.
__device__ void rr_minmax_f32(float* x, int n, double* red) {
    if (n == 0) return;
    double lo, hi;
    bool bad;
    rr_block_minmax(x, n, red, lo, hi, bad);
    const bool zero = bad || !rr_finite(lo) || !rr_finite(hi) || (hi - lo) < 1e-12;
    const float lo32 = (float)lo;
    const float den32 = (float)(hi - lo + 1e-12);
    for (int i = threadIdx.x; i < n; i += RR_FUSE_THREADS) x[i] = zero ? 0.f : (x[i] - lo32) / den32;
    __syncthreads();
}

// Bitonic sort, descending by key, carrying a 32-bit payload.
__device__ void rr_bitonic_desc_kv(uint64_t* keys, int32_t* vals, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (n >> 1); i += RR_FUSE_THREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b; keys[hi] = a;
                    const int32_t t = vals[lo]; vals[lo] = vals[hi]; vals[hi] = t;
                }
            }
        }
    }
    __syncthreads();
}

__device__ void rr_bitonic_desc_pairs(uint64_t* keys, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (n >> 1); i += RR_FUSE_THREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

// LDS plan: sort keys + 6 K reduction scratch + four float64 columns + six float32 columns + the slot map, each sized for
// THIS launch's pool and candidate count (rr_fuse_lds_bytes): 20 KB at pool 150 / 150 candidates, 38 KB when eight ranks'
// lists are merged -- four to six workgroups share a CU (the kernel is latency-bound: on the few CUs a masked stream gives
// it, or with more queries than CUs, that is the difference between one round and several) -- up to 158 KB at pool 2048.
__global__ __launch_bounds__(RR_FUSE_THREADS) void rr_fuse(
    rr_fuse_dev_params fp, const int64_t* __restrict__ g_rows, const float* __restrict__ g_dense,
    const float* __restrict__ g_bm25, const double* __restrict__ g_n, const double* __restrict__ g_avg,
    const double* __restrict__ g_l1p, const float* __restrict__ g_rerank, const float* __restrict__ g_best,
    const float* __restrict__ g_gate, const double* __restrict__ ix_n, const double* __restrict__ ix_avg,
    const double* __restrict__ ix_l1p, int64_t* __restrict__ out_rows, double* __restrict__ out_cols,
    int32_t* __restrict__ out_order) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int ncand = fp.p.n_candidates;
    const int pool = fp.p.pool;
    const int k = fp.p.k;

    // carve-up
    const int KC = fp.key_cap, PC = fp.col_cap;                            // (rr_fuse_lds_bytes: the same two numbers)
    uint64_t* keys = reinterpret_cast<uint64_t*>(lds);                    // KC
    double* red = reinterpret_cast<double*>(keys + KC);                    // 3 * threads
    double* c_n = red + 3 * RR_FUSE_THREADS;                               // PC each, float64 (the merge's slot map, KC int32, sits here first)
    double* c_avg = c_n + PC;
    double* c_l1p = c_avg + PC;
    double* c_prior = c_l1p + PC;
    float* c_dense = reinterpret_cast<float*>(c_prior + PC);              // PC each, float32
    float* c_bm25 = c_dense + PC;
    float* c_rr = c_bm25 + PC;
    float* c_best = c_rr + PC;
    float* c_trust = c_best + PC;
    float* c_final = c_trust + PC;
    int32_t* c_src = reinterpret_cast<int32_t*>(c_final + PC);             // candidate slot of pool pos
    double* s_scalar = reinterpret_cast<double*>(c_src + PC);              // 4 scalars

    const rr_cand_addr at{q, ncand, fp.p.cand_per_rank, fp.p.cand_rank_stride_bytes};

    // ---- 0. shard merge: best `pool` candidates by (dense desc, row asc)
    if (ncand > pool) {
        int n_sort = 1;
        while (n_sort < ncand) n_sort <<= 1;
        int32_t* slot = reinterpret_cast<int32_t*>(c_n);  // column area is free until step 1
        int32_t* unsorted = reinterpret_cast<int32_t*>(s_scalar);
        if (tid == 0) *unsorted = 0;
        for (int i = tid; i < n_sort; i += RR_FUSE_THREADS) {
            uint64_t key = 0;
            if (i < ncand) {
                float d = at.get(g_dense, i);
                d = (d == d) ? d : -INFINITY;
                // score key in the high word, row (rows < 2^32) in the low word: ties go to
                // the smaller global row whichever shard supplied it
                key = ((uint64_t)rr_f2key(d) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)at.get(g_rows, i));
            }
            keys[i] = key;
            slot[i] = i < ncand ? i : 0;
        }
        __syncthreads();
        const int cpr = fp.p.cand_per_rank;
        if (cpr > 0) {      // (are the ranks' lists in key order, as K1 leaves them?)
            for (int i = tid; i < ncand; i += RR_FUSE_THREADS)
                if (i % cpr != 0 && keys[i - 1] < keys[i]) atomicOr(unsorted, 1);
        }
        __syncthreads();
        if (cpr > 0 && !*unsorted) {
            // Every rank's list is already ordered (K1: score desc, row asc = key desc): a candidate's place in the merged
            // order is its place in its own list + the number of larger keys in every other list (a binary search each;
            // equal keys -- the same row offered twice -- go to the earlier list) -- no sort, no barrier per stage: 73 -> ~30 us
            // for 8 x 150 candidates.  The places 0 .. pool - 1 are each taken exactly once.
            const int nl = ncand / cpr;
            int top = 1;
            while (top * 2 <= cpr) top *= 2;           // largest power of two <= cpr
            for (int i = tid; i < ncand; i += RR_FUSE_THREADS) {
                const uint64_t key = keys[i];
                const int r = i / cpr;
                int place = i - r * cpr;
                // branch-free searches, eight lists at a time: their LDS reads are independent and overlap
                for (int r0 = 0; r0 < nl; r0 += 8) {
                    int lo[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) lo[u] = 0;
                    for (int step = top; step >= 1; step >>= 1) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int r2 = r0 + u < nl ? r0 + u : nl - 1;
                            const int mid = lo[u] + step;            // lo = how many keys of the list come before this one
                            const uint64_t v = keys[r2 * cpr + (mid <= cpr ? mid - 1 : cpr - 1)];
                            const bool before = r2 < r ? v >= key : v > key;
                            lo[u] = (mid <= cpr && before) ? mid : lo[u];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) place += (r0 + u < nl && r0 + u != r) ? lo[u] : 0;
                }
                if (place < pool) c_src[place] = i;
            }
        } else {
            rr_bitonic_desc_kv(keys, slot, n_sort);
            for (int i = tid; i < pool; i += RR_FUSE_THREADS) c_src[i] = slot[i];
        }
    } else {
        for (int i = tid; i < pool; i += RR_FUSE_THREADS) c_src[i] = i;
    }
    __syncthreads();

    // ---- 1. load the pool columns
    for (int i = tid; i < pool; i += RR_FUSE_THREADS) {
        const int s = c_src[i];
        const int64_t row = at.get(g_rows, s);
        c_dense[i] = at.get(g_dense, s);
        c_bm25[i] = g_bm25 ? at.get(g_bm25, s) : 0.f;
        c_rr[i] = (g_rerank && fp.p.rerank_active && i < fp.p.rerank_k) ? g_rerank[(int64_t)q * pool + i] : 0.f;
        c_best[i] = g_best ? g_best[(int64_t)q * pool + i] : 0.f;
        if (g_n) {
            c_n[i] = at.get(g_n, s); c_avg[i] = at.get(g_avg, s); c_l1p[i] = at.get(g_l1p, s);
        } else {
            const int64_t local = row - fp.row_offset;
            c_n[i] = ix_n[local]; c_avg[i] = ix_avg[local]; c_l1p[i] = ix_l1p[local];
        }
        out_rows[(int64_t)q * pool + i] = row;
    }
    __syncthreads();

    // ---- 2. _dense, _bm25 (float32 min-max)
    rr_minmax_f32(c_dense, pool, red);
    rr_minmax_f32(c_bm25, pool, red);

    // ---- 3. priors (float64)
    // g = nanmean(avg): NaNs replaced by 0, pairwise sum, divided by the non-NaN count
    for (int i = tid; i < pool; i += RR_FUSE_THREADS) c_prior[i] = (c_avg[i] == c_avg[i]) ? c_avg[i] : 0.0;
    __syncthreads();
    if (tid == 0) {
        int cnt = 0;
        for (int i = 0; i < pool; ++i) cnt += (c_avg[i] == c_avg[i]) ? 1 : 0;
        const double tot = rr_np_pairwise(c_prior, pool);
        s_scalar[0] = tot / (double)cnt;  // 0/0 = NaN for an all-NaN pool, like np.nanmean
    }
    __syncthreads();
    const double gmean = s_scalar[0];
    for (int i = tid; i < pool; i += RR_FUSE_THREADS)
        c_prior[i] = ((c_avg[i] * c_n[i]) + (gmean * fp.p.prior_C)) / (c_n[i] + fp.p.prior_C + 1e-9);
    __syncthreads();
    {
        double lo, hi, l1lo, l1hi;
        bool bad, l1bad;
        rr_block_minmax(c_prior, pool, red, lo, hi, bad);
        rr_block_minmax(c_l1p, pool, red, l1lo, l1hi, l1bad);
        const bool zero = bad || !rr_finite(lo) || !rr_finite(hi) || (hi - lo) < 1e-12;
        const double vden = (l1bad ? (double)NAN : l1hi) + 1e-9;
        for (int i = tid; i < pool; i += RR_FUSE_THREADS) {
            const float mm = zero ? 0.f : (float)((c_prior[i] - lo) / (hi - lo + 1e-12));
            const double vol = c_l1p[i] / vden;
            c_prior[i] = (double)(mm * 0.7f) + 0.3 * vol;
        }
        __syncthreads();
    }

    // ---- 4. _rerank: min-max over the first rr_k rows only; _best over the pool when present
    if (fp.p.rerank_active) {
        int rr_k = fp.p.rerank_k < pool ? fp.p.rerank_k : pool;
        if (g_rerank) rr_minmax_f32(c_rr, rr_k, red);
    }
    if (g_best) rr_minmax_f32(c_best, pool, red);

    // ---- 5. trust, blend, gate
    const double ramp_den = (double)(fp.p.min_reviews > 1 ? fp.p.min_reviews : 1);
    for (int i = tid; i < pool; i += RR_FUSE_THREADS) {
        double ramp = c_n[i] / ramp_den;
        ramp = ramp < 0.0 ? 0.0 : (ramp > 1.0 ? 1.0 : ramp);   // np.clip keeps NaN as NaN
        double sat = c_l1p[i] / fp.log1p_sat;
        sat = (sat != sat) ? sat : (1.0 < sat ? 1.0 : sat);     // np.minimum propagates NaN
        const float trust = fp.p.apply_trust ? (float)(0.6 * ramp + 0.4 * sat) : 1.0f;
        c_trust[i] = trust;
        const float gate = g_gate ? g_gate[(int64_t)q * pool + i] : 1.0f;

        const float t_dense = fp.w_dense32 * c_dense[i];
        const float t_bm25 = fp.w_bm2532 * c_bm25[i];
        float s32 = t_dense + t_bm25;
        double s64;
        if (fp.p.bm25_f64) {
            // CLI without a BM25 artefact: `cand["_bm25"] = 0.0` is a float64 column (app/test.py:252), so the
            // sum is float64 from its second term on and a float32 rerank product is widened before it is added
            s64 = (double)t_dense + fp.p.w_bm25 * 0.0;
            s64 = fp.p.rerank_active ? s64 + (double)(fp.w_rerank32 * c_rr[i]) : s64 + fp.p.w_rerank * 0.0;
        } else if (fp.p.rerank_active) {
            s32 = s32 + fp.w_rerank32 * c_rr[i];
            s64 = (double)s32;
        } else {
            s64 = (double)s32 + fp.p.w_rerank * 0.0;
        }
        s64 = s64 + fp.p.w_prior * c_prior[i];
        s64 = s64 + (double)(fp.w_best32 * c_best[i]);
        float fin = (float)s64;
        if (fp.p.apply_trust) fin = fin * trust;
        fin = fin * gate;
        c_final[i] = fin;

        double* oc = out_cols + ((int64_t)q * 8) * pool;
        oc[0 * pool + i] = (double)c_dense[i];
        oc[1 * pool + i] = (double)c_bm25[i];
        oc[2 * pool + i] = c_prior[i];
        oc[3 * pool + i] = (double)c_rr[i];
        oc[4 * pool + i] = (double)c_best[i];
        oc[5 * pool + i] = (double)gate;
        oc[6 * pool + i] = (double)trust;
        oc[7 * pool + i] = (double)fin;
    }
    __syncthreads();

    // ---- 6. stable order by final desc (NaN last), first k
    {
        int n_sort = 1;
        while (n_sort < pool) n_sort <<= 1;
        for (int i = tid; i < n_sort; i += RR_FUSE_THREADS) {
            uint64_t key = 0;
            if (i < pool) {
                const float f = c_final[i];
                // NaN sorts below every number (pandas na_position='last'); slot breaks ties
                const uint32_t fk = (f == f) ? rr_f2key(f) : 0u;
                key = ((uint64_t)fk << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
            }
            keys[i] = key;
        }
        rr_bitonic_desc_pairs(keys, n_sort);
        for (int i = tid; i < k; i += RR_FUSE_THREADS)
            out_order[(int64_t)q * k + i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFu));
    }
}

__global__ void rr_gather_meta(const int64_t* __restrict__ rows, int64_t n, int64_t row_offset,
                               int64_t n_rows, const double* __restrict__ ix_n,
                               const double* __restrict__ ix_avg, const double* __restrict__ ix_l1p,
                               double* __restrict__ o_n, double* __restrict__ o_avg,
                               double* __restrict__ o_l1p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t local = rows[i] - row_offset;
    const bool ok = local >= 0 && local < n_rows;
    o_n[i] = ok ? ix_n[local] : 0.0;
    o_avg[i] = ok ? ix_avg[local] : (double)NAN;
    o_l1p[i] = ok ? ix_l1p[local] : 0.0;
}

// ------------------------------------------------------------------ host side
// LDS capacities of a launch: keys = the power of two the merge / the final sort pad to; columns = pool rounded up to 64,
// and at least KC / 8 doubles per float64 column area so that the merge's slot map (KC int32 in the four float64 columns'
// space) fits: 4 columns x PC x 8 B >= KC x 4 B.
static void rr_fuse_caps(const rr_fuse_params* p, int32_t* key_cap, int32_t* col_cap) {
    int kc = 64;
    while (kc < p->n_candidates || kc < p->pool) kc <<= 1;
    int pc = (p->pool + 63) / 64 * 64;
    if (pc * 8 < kc) pc = kc / 8;
    *key_cap = kc;
    *col_cap = pc;
}
static size_t rr_fuse_lds_bytes(int key_cap, int col_cap) {
    return sizeof(uint64_t) * (size_t)key_cap + sizeof(double) * 3 * RR_FUSE_THREADS +
           sizeof(double) * 4 * (size_t)col_cap + sizeof(float) * 6 * (size_t)col_cap +
           sizeof(int32_t) * (size_t)col_cap + sizeof(double) * 4;
}

static int rr_fuse_check(const rr_index* ix, const rr_fuse_params* p, int32_t nq, const void* rows,
                         const void* dense, const void* n, const void* avg, const void* l1p) {
    RR_REQUIRE(ix && p && rows && dense, "rr_fuse_topk: NULL argument");
    RR_REQUIRE(nq >= 1 && nq <= RR_MAX_BATCH, "rr_fuse_topk: n_queries %d out of [1,%d]", nq, RR_MAX_BATCH);
    RR_REQUIRE(p->pool >= 1 && p->pool <= RR_MAX_POOL, "rr_fuse_topk: pool %d out of [1,%d]", p->pool, RR_MAX_POOL);
    RR_REQUIRE(p->n_candidates >= p->pool && p->n_candidates <= RR_FUSE_MAXCAND,
               "rr_fuse_topk: n_candidates %d out of [pool=%d,%d]", p->n_candidates, p->pool, RR_FUSE_MAXCAND);
    RR_REQUIRE(p->k >= 1 && p->k <= p->pool, "rr_fuse_topk: k %d out of [1,pool=%d]", p->k, p->pool);
    RR_REQUIRE((n && avg && l1p) || (!n && !avg && !l1p), "rr_fuse_topk: pass all three meta columns or none");
    RR_REQUIRE(p->cand_per_rank >= 0 && p->cand_rank_stride_bytes >= 0 && p->cand_rank_stride_bytes % 8 == 0,
               "rr_fuse_topk: bad shard payload geometry");
    RR_REQUIRE(p->cand_per_rank == 0 || p->n_candidates % p->cand_per_rank == 0,
               "rr_fuse_topk: n_candidates %d is not a multiple of cand_per_rank %d", p->n_candidates, p->cand_per_rank);
    if (!n) {
        if (!ix->has_meta) {
            rr_set_error("rr_fuse_topk: index has no metadata (call rr_index_set_meta) and none was passed");
            return RR_E_STATE;
        }
        RR_REQUIRE(p->n_candidates == p->pool,
                   "rr_fuse_topk: merging shards (n_candidates > pool) needs the meta columns in the payload");
    }
    return RR_OK;
}

extern "C" int rr_fuse_topk_dev(rr_index* ix, const rr_fuse_params* p, int32_t n_queries,
                                const int64_t* d_rows, const float* d_dense, const float* d_bm25,
                                const double* d_n_reviews, const double* d_avg_stars,
                                const double* d_log1p_n, const float* d_rerank, const float* d_best,
                                const float* d_gate, int64_t* d_out_rows, double* d_out_cols,
                                int32_t* d_out_order, void* stream) {
    int rc = rr_fuse_check(ix, p, n_queries, d_rows, d_dense, d_n_reviews, d_avg_stars, d_log1p_n);
    if (rc) return rc;
    RR_REQUIRE(d_out_rows && d_out_cols && d_out_order, "rr_fuse_topk_dev: NULL output");
    RR_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream
    rr_fuse_dev_params fp;
    fp.p = *p;
    const int sat = p->trust_sat > 1 ? p->trust_sat : 1;
    fp.log1p_sat = log1p((double)sat);
    fp.w_dense32 = (float)p->w_dense;
    fp.w_bm2532 = (float)p->w_bm25;
    fp.w_rerank32 = (float)p->w_rerank;
    fp.w_best32 = (float)p->w_best;
    fp.row_offset = ix->row_offset;
    fp.n_rows = ix->n_rows;
    rr_fuse_caps(p, &fp.key_cap, &fp.col_cap);
    const size_t lds = rr_fuse_lds_bytes(fp.key_cap, fp.col_cap);
    static std::mutex attr_mu;
    static bool attr_done[64] = {false};
    {   // (the opt-in to more than 64 KB of dynamic LDS is per device, once: the largest launch)
        std::lock_guard<std::mutex> lk(attr_mu);
        if (ix->device >= 0 && ix->device < 64 && !attr_done[ix->device]) {
            RR_HIP_TRY(hipFuncSetAttribute((const void*)rr_fuse, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)rr_fuse_lds_bytes(RR_FUSE_MAXCAND, RR_MAX_POOL)));
            attr_done[ix->device] = true;
        }
    }
    hipLaunchKernelGGL(rr_fuse, dim3(n_queries), dim3(RR_FUSE_THREADS), lds, st, fp, d_rows, d_dense,
                       d_bm25, d_n_reviews, d_avg_stars, d_log1p_n, d_rerank, d_best, d_gate,
                       ix->d_n_reviews, ix->d_avg_stars, ix->d_log1p_n, d_out_rows, d_out_cols,
                       d_out_order);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

namespace {
struct DevBuf {
    void* p = nullptr;
    hipError_t put(const void* h, size_t bytes, hipStream_t st) {
        if (!h) return hipSuccess;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(p, h, bytes, hipMemcpyHostToDevice, st);
        return e;
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    ~DevBuf() { if (p) hipFree(p); }
};
}  // namespace

extern "C" int rr_fuse_topk(rr_index* ix, const rr_fuse_params* p, int32_t n_queries,
                            const int64_t* h_rows, const float* h_dense, const float* h_bm25,
                            const double* h_n_reviews, const double* h_avg_stars,
                            const double* h_log1p_n, const float* h_rerank, const float* h_best,
                            const float* h_gate, int64_t* h_out_rows, double* h_out_cols,
                            int32_t* h_out_order) {
    int rc = rr_fuse_check(ix, p, n_queries, h_rows, h_dense, h_n_reviews, h_avg_stars, h_log1p_n);
    if (rc) return rc;
    RR_REQUIRE(h_out_rows && h_out_cols && h_out_order, "rr_fuse_topk: NULL output");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = ix->stream;
    const size_t nc = (size_t)n_queries * p->n_candidates, np_ = (size_t)n_queries * p->pool;
    DevBuf rows, dense, bm25, n, avg, l1p, rr, best, gate, orow, ocol, oord;
    hipError_t e = rows.put(h_rows, nc * 8, st);
    if (e == hipSuccess) e = dense.put(h_dense, nc * 4, st);
    if (e == hipSuccess) e = bm25.put(h_bm25, nc * 4, st);
    if (e == hipSuccess) e = n.put(h_n_reviews, nc * 8, st);
    if (e == hipSuccess) e = avg.put(h_avg_stars, nc * 8, st);
    if (e == hipSuccess) e = l1p.put(h_log1p_n, nc * 8, st);
    if (e == hipSuccess) e = rr.put(h_rerank, np_ * 4, st);
    if (e == hipSuccess) e = best.put(h_best, np_ * 4, st);
    if (e == hipSuccess) e = gate.put(h_gate, np_ * 4, st);
    if (e == hipSuccess) e = orow.alloc(np_ * 8);
    if (e == hipSuccess) e = ocol.alloc(np_ * 8 * 8);
    if (e == hipSuccess) e = oord.alloc((size_t)n_queries * p->k * 4);
    if (e != hipSuccess) { rr_set_error("rr_fuse_topk: %s", hipGetErrorString(e)); return RR_E_HIP; }
    rc = rr_fuse_topk_dev(ix, p, n_queries, (const int64_t*)rows.p, (const float*)dense.p,
                          (const float*)bm25.p, (const double*)n.p, (const double*)avg.p,
                          (const double*)l1p.p, (const float*)rr.p, (const float*)best.p,
                          (const float*)gate.p, (int64_t*)orow.p, (double*)ocol.p, (int32_t*)oord.p, st);
    if (rc) return rc;
    RR_HIP_TRY(hipMemcpyAsync(h_out_rows, orow.p, np_ * 8, hipMemcpyDeviceToHost, st));
    RR_HIP_TRY(hipMemcpyAsync(h_out_cols, ocol.p, np_ * 8 * 8, hipMemcpyDeviceToHost, st));
    RR_HIP_TRY(hipMemcpyAsync(h_out_order, oord.p, (size_t)n_queries * p->k * 4, hipMemcpyDeviceToHost, st));
    RR_HIP_TRY(hipStreamSynchronize(st));
    return RR_OK;
}

extern "C" int rr_index_gather_meta_dev(rr_index* ix, const int64_t* d_rows, int64_t n,
                                        double* d_n_reviews, double* d_avg_stars, double* d_log1p_n,
                                        void* stream) {
    RR_REQUIRE(ix && d_rows && d_n_reviews && d_avg_stars && d_log1p_n, "rr_index_gather_meta_dev: NULL argument");
    RR_REQUIRE(n >= 0, "rr_index_gather_meta_dev: negative n");
    if (!ix->has_meta) { rr_set_error("rr_index_gather_meta_dev: index has no metadata"); return RR_E_STATE; }
    if (n == 0) return RR_OK;
    RR_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream
    hipLaunchKernelGGL(rr_gather_meta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_rows, n,
                       ix->row_offset, ix->n_rows, ix->d_n_reviews, ix->d_avg_stars, ix->d_log1p_n,
                       d_n_reviews, d_avg_stars, d_log1p_n);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}
