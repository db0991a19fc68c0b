// rr_reviews.hip -- best review per candidate product.
//
// Replaces _best_snippets (app/app_product_search.py:320-370) / best_review_snippets
// (app/test.py:181-215): among the reviews of each candidate sku, the one whose embedding has
// the largest dot product with the query (`En @ qvec`, per-sku argmax = first maximum in file
// order).  The reference re-reads the whole reviews parquet per query; here the normalised review
// embeddings stay in HBM, grouped by product through a CSR (product row -> ascending review
// ids), so a query touches only the reviews of its <= pool candidates.
//
// Latency-bound: one 256-thread workgroup per (query, candidate); a 16-lane DPP row scores one
// review with the summation order of rr_scan_f32 (one fmaf chain per lane, rr_row16_sum).
#include "rr_common.h"

struct rr_reviews {
    int device = 0;
    int64_t n_reviews = 0, n_products = 0;
    int32_t dim = 0, dim_pad = 0;
    float* d_emb = nullptr;          // n_reviews x dim_pad, rows l2-normalised
    int64_t* d_indptr = nullptr;     // n_products + 1
    int32_t* d_ids = nullptr;        // review ids, ascending per product
    int32_t* d_cut = nullptr;        // [RR_MAX_BATCH] per-query review-id cut of rr_reviews_best_cut_dev
    std::mutex mu;
};


__global__ __launch_bounds__(256) void rr_best_review(
    const f32x4* __restrict__ emb, int nf, const int64_t* __restrict__ indptr, const int32_t* __restrict__ ids,
    int64_t n_products, const float* __restrict__ queries, int dim, const int64_t* __restrict__ rows, int pool,
    int64_t row_offset, int32_t max_review_id_all, const int32_t* __restrict__ cuts,
    float* __restrict__ best_score, int32_t* __restrict__ best_id) {
    __shared__ uint64_t wbest[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int sub = lane & 15, grp = lane >> 4;
    const int q = blockIdx.y, c = blockIdx.x;
    const int32_t max_review_id = cuts ? cuts[q] : max_review_id_all;
    const int64_t prod = rows[(int64_t)q * pool + c] - row_offset;
    uint64_t best = 0;                                    // (score key << 32) | ~review id ; 0 = none
    if (prod >= 0 && prod < n_products) {
        const int64_t s = indptr[prod], e = indptr[prod + 1];
        const float* qv = queries + (int64_t)q * dim;
        for (int64_t base = s + 4 * w; base < e; base += 16) {   // 4 waves x 4 reviews per step
            const int64_t j = base + grp;
            const bool have = j < e;
            const int32_t rid = have ? ids[j] : 0;
            const f32x4* p = emb + (int64_t)rid * (nf * 16) + sub;
            float acc = 0.f;
            for (int i = 0; i < nf; ++i) {
                const f32x4 x = p[16 * i];
                const int k = 4 * (sub + 16 * i);
                const float q0 = k + 0 < dim ? qv[k + 0] : 0.f, q1 = k + 1 < dim ? qv[k + 1] : 0.f;
                const float q2 = k + 2 < dim ? qv[k + 2] : 0.f, q3 = k + 3 < dim ? qv[k + 3] : 0.f;
                acc = __builtin_fmaf(x.x, q0, acc);
                acc = __builtin_fmaf(x.y, q1, acc);
                acc = __builtin_fmaf(x.z, q2, acc);
                acc = __builtin_fmaf(x.w, q3, acc);
            }
            acc = rr_row16_sum(acc);
            if (have && rid <= max_review_id) {
                const float v = acc == acc ? acc : -INFINITY;   // np.argmax treats NaN as the maximum;
                                                                // a NaN review embedding is not expected
                const uint64_t key = ((uint64_t)rr_f2key(v) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)rid);
                best = key > best ? key : best;                 // larger score, then smaller review id
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t o = __shfl_xor(best, m, 64);
        best = o > best ? o : best;
    }
    if (lane == 0) wbest[w] = best;
    __syncthreads();
    if (tid == 0) {
        uint64_t b = wbest[0];
        for (int i = 1; i < 4; ++i) b = wbest[i] > b ? wbest[i] : b;
        const int64_t o = (int64_t)q * pool + c;
        best_score[o] = b ? rr_key2f((uint32_t)(b >> 32)) : 0.f;
        best_id[o] = b ? (int32_t)(0xFFFFFFFFu - (uint32_t)(b & 0xFFFFFFFFu)) : -1;
    }
}

// The reference scores only the first `max_rows` reviews, in file order, of all reviews whose sku is among
// the query's candidates (`sub_meta.iloc[:max_rows]`, app/app_product_search.py:342-345, app/test.py:200-203).
// File order = review id, so the cut is a review-id threshold: the max_rows-th smallest id in the union of
// the candidates' (ascending) lists; n_reviews (= keep all) when the union is not longer than max_rows.
// One workgroup per query: bisection on the id, counts by binary search in every candidate's list.
__global__ __launch_bounds__(256) void rr_reviews_cut(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ ids, int64_t n_products,
    const int64_t* __restrict__ rows, int pool, int64_t row_offset, int64_t max_rows, int64_t n_reviews,
    int32_t* __restrict__ cut) {
    __shared__ long long wsum[4];
    const int tid = threadIdx.x, q = blockIdx.x;
    auto count_le = [&](int64_t T) -> long long {      // reviews with id <= T among the candidates (block-wide)
        long long c = 0;
        for (int i = tid; i < pool; i += 256) {
            const int64_t prod = rows[(int64_t)q * pool + i] - row_offset;
            if (prod < 0 || prod >= n_products) continue;
            int64_t lo = indptr[prod], hi = indptr[prod + 1];
            const int64_t s = lo;
            while (lo < hi) {                             // first entry > T
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)ids[mid] <= T) lo = mid + 1; else hi = mid;
            }
            c += lo - s;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
        __syncthreads();
        if ((tid & 63) == 0) wsum[tid >> 6] = c;
        __syncthreads();
        return wsum[0] + wsum[1] + wsum[2] + wsum[3];
    };
    int32_t out;
    if (max_rows <= 0) out = -1;
    else if (count_le(n_reviews) <= max_rows) out = (int32_t)n_reviews;
    else {
        int64_t lo = 0, hi = n_reviews - 1;               // smallest T with count_le(T) >= max_rows
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (count_le(mid) >= max_rows) hi = mid; else lo = mid + 1;
        }
        out = (int32_t)lo;
    }
    if (tid == 0) cut[q] = out;
}

__global__ void rr_reviews_l2norm(float* __restrict__ mat, int64_t n_rows, int dim_pad, float eps) {
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

extern "C" int rr_reviews_destroy(rr_reviews* rv) {
    if (!rv) return RR_OK;
    hipSetDevice(rv->device);
    hipFree(rv->d_emb); hipFree(rv->d_indptr); hipFree(rv->d_ids); hipFree(rv->d_cut);
    delete rv;
    return RR_OK;
}

extern "C" int rr_reviews_create(const float* h_emb, int64_t n_reviews, int32_t dim, int64_t n_products,
                                 const int64_t* h_indptr, const int32_t* h_ids, int32_t device,
                                 float normalize_eps, rr_reviews** out) {
    RR_REQUIRE(out, "rr_reviews_create: NULL out");
    *out = nullptr;
    RR_REQUIRE(h_emb && h_indptr && n_reviews >= 1 && n_reviews < (1ll << 31) && n_products >= 1 && dim >= 1,
               "rr_reviews_create: bad argument");
    const int64_t nnz = h_indptr[n_products];
    RR_REQUIRE(nnz >= 0 && nnz <= n_reviews && (nnz == 0 || h_ids), "rr_reviews_create: bad CSR");
    RR_HIP_TRY(hipSetDevice(device));
    rr_reviews* rv = new rr_reviews();
    rv->device = device; rv->n_reviews = n_reviews; rv->n_products = n_products;
    rv->dim = dim; rv->dim_pad = (int32_t)rr_round_up(dim, 64);
    hipError_t e = hipMalloc((void**)&rv->d_emb, sizeof(float) * (size_t)n_reviews * rv->dim_pad);
    if (e == hipSuccess) e = hipMalloc((void**)&rv->d_indptr, sizeof(int64_t) * (size_t)(n_products + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&rv->d_ids, sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    if (e == hipSuccess) e = hipMalloc((void**)&rv->d_cut, sizeof(int32_t) * RR_MAX_BATCH);
    if (e == hipSuccess && rv->dim_pad != dim) e = hipMemset(rv->d_emb, 0, sizeof(float) * (size_t)n_reviews * rv->dim_pad);
    if (e == hipSuccess)
        e = hipMemcpy2D(rv->d_emb, sizeof(float) * rv->dim_pad, h_emb, sizeof(float) * dim, sizeof(float) * dim,
                        (size_t)n_reviews, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(rv->d_indptr, h_indptr, sizeof(int64_t) * (size_t)(n_products + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(rv->d_ids, h_ids, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice);
    if (e == hipSuccess && normalize_eps > 0.f) {
        hipLaunchKernelGGL(rr_reviews_l2norm, dim3((unsigned)((n_reviews + 3) / 4)), dim3(256), 0, nullptr, rv->d_emb,
                           n_reviews, rv->dim_pad, normalize_eps);
        e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) {
        rr_set_error("rr_reviews_create: %s", hipGetErrorString(e));
        rr_reviews_destroy(rv);
        return RR_E_HIP;
    }
    *out = rv;
    return RR_OK;
}

extern "C" int rr_reviews_best_dev(rr_reviews* rv, const float* d_queries, int32_t n_queries,
                                   const int64_t* d_rows, int32_t pool, int64_t row_offset,
                                   int32_t max_review_id, float* d_best_score, int32_t* d_best_id, void* stream) {
    RR_REQUIRE(rv && d_queries && d_rows && d_best_score && d_best_id, "rr_reviews_best_dev: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH && pool >= 1 && pool <= RR_MAX_POOL,
               "rr_reviews_best_dev: n_queries %d / pool %d out of range", n_queries, pool);
    RR_HIP_TRY(hipSetDevice(rv->device));
    hipLaunchKernelGGL(rr_best_review, dim3((unsigned)pool, (unsigned)n_queries), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(rv->d_emb), rv->dim_pad / 64, rv->d_indptr, rv->d_ids,
                       rv->n_products, d_queries, rv->dim, d_rows, pool, row_offset, max_review_id,
                       (const int32_t*)nullptr, d_best_score, d_best_id);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

extern "C" int rr_reviews_best_cut_dev(rr_reviews* rv, const float* d_queries, int32_t n_queries,
                                       const int64_t* d_rows, int32_t pool, int64_t row_offset,
                                       int64_t max_rows, float* d_best_score, int32_t* d_best_id, void* stream) {
    RR_REQUIRE(rv && d_queries && d_rows && d_best_score && d_best_id, "rr_reviews_best_cut_dev: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH && pool >= 1 && pool <= RR_MAX_POOL,
               "rr_reviews_best_cut_dev: n_queries %d / pool %d out of range", n_queries, pool);
    std::lock_guard<std::mutex> lk(rv->mu);
    RR_HIP_TRY(hipSetDevice(rv->device));
    hipLaunchKernelGGL(rr_reviews_cut, dim3((unsigned)n_queries), dim3(256), 0, (hipStream_t)stream, rv->d_indptr,
                       rv->d_ids, rv->n_products, d_rows, pool, row_offset, max_rows, rv->n_reviews, rv->d_cut);
    hipLaunchKernelGGL(rr_best_review, dim3((unsigned)pool, (unsigned)n_queries), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(rv->d_emb), rv->dim_pad / 64, rv->d_indptr, rv->d_ids,
                       rv->n_products, d_queries, rv->dim, d_rows, pool, row_offset, 0, (const int32_t*)rv->d_cut,
                       d_best_score, d_best_id);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}
