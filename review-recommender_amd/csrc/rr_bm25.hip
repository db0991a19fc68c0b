// rr_bm25.hip -- K2: Okapi BM25 over CSR postings (gfx950).
//
// Replaces rank_bm25's BM25Okapi.get_scores as the reference calls it
// (app/app_product_search.py:206, app/test.py:170) and the candidate gather that
// follows (app/app_product_search.py:207-208, app/test.py:171-173).
//
// Arithmetic (float64, one rounding per operation, no contraction -- this file is
// built with -ffp-contract=off):
//   per query token t, in query order, for every document d with tf(t,d) > 0
//     score[d] += idf[t] * ( tf*(k1+1) / ( tf + k1*((1-b) + b*dl[d]/avgdl) ) )
// which is what numpy evaluates for `score += idf * (q_freq*(k1+1) / (q_freq + k1*(1 - b +
// b*doc_len/avgdl)))`; documents with tf == 0 receive +0.0 there.
//
// Data layout in HBM (int32 ids, 8 B per posting in each orientation)
//   postings  post_indptr[n_terms+1] ; post_docs[nnz] ascending per term ; post_tf[nnz]
//   forward   doc_indptr[n_docs+1]   ; doc_terms[nnz] ascending per doc  ; doc_tf[nnz]
//   doc_len[n_docs] int32 ; idf[n_terms] float64 (epsilon floor applied on the host)
//
// rr_bm25_slices (get_scores): HBM/L2-bound.  A workgroup owns RR_SLICE consecutive
// documents and keeps their float64 scores in LDS; for each query token it finds the
// block of the term's postings that falls in its document range (two binary searches),
// streams that block and adds into LDS -- no global scatter, no atomics, and the token
// order of the additions is the reference's.  One coalesced 8-B store per document ends it.
//
// rr_bm25_at (hybrid path): latency-bound.  One thread per (candidate, token) pair binary-
// searches the candidate's forward list (mode 0, ~6 probes) or the term's postings list
// (mode 1, ~log2 df probes) for tf; contributions are summed per candidate in token order.
#include "rr_common.h"

#define RR_SLICE 4096
#define RR_AT_CANDS 64

struct rr_bm25_view {
    const int64_t* post_indptr;
    const int32_t* post_docs;
    const int32_t* post_tf;
    const int64_t* doc_indptr;
    const int32_t* doc_terms;
    const int32_t* doc_tf;
    const int32_t* doc_len;
    const double* idf;
    double avgdl, k1, b;
    int64_t n_docs, n_terms, row_offset;
};

__device__ __forceinline__ double rr_bm25_term(double idf, int tf, int dl, double avgdl, double k1,
                                               double b) {
    const double f = (double)tf;
    const double one_minus_b = 1.0 - b;
    const double norm = one_minus_b + (b * (double)dl) / avgdl;
    return idf * ((f * (k1 + 1.0)) / (f + k1 * norm));
}

// first index in [lo, hi) with a[i] >= key
__device__ __forceinline__ int64_t rr_lower_bound(const int32_t* __restrict__ a, int64_t lo,
                                                  int64_t hi, int32_t key) {
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// first index in [lo, hi) with a[i] >= key, found by a whole wave: 64 probes per step (log64 steps instead of log2: a
// workgroup's first use of a long postings list is a chain of DEPENDENT loads -- 4 instead of 23 at df = 10M)
__device__ __forceinline__ int64_t rr_lower_bound_wave(const int32_t* __restrict__ a, int64_t lo, int64_t hi, int32_t key) {
    const int lane = threadIdx.x & 63;
    while (hi - lo > 64) {
        const int64_t stride = (hi - lo + 63) >> 6;
        const int64_t at = lo + lane * stride;
        const bool below = at < hi && a[at] < key;          // (sorted: the lanes that answer "below" are a prefix)
        const int c = __popcll(__ballot(below));
        if (c == 0) return lo;                               // a[lo] >= key
        const int64_t nlo = lo + (int64_t)(c - 1) * stride + 1, nhi = lo + (int64_t)c * stride;
        lo = nlo;
        hi = nhi < hi ? nhi : hi;
    }
    const bool below = lo + lane < hi && a[lo + lane] < key;
    return lo + __popcll(__ballot(below));
}

#define RR_SLICE_TERMS 16     // query tokens whose posting blocks a workgroup locates in one go
// get_scores: see the file header.  `dirty[slice]` (in/out): whether the slice of `out` holds anything but zeros -- `out`
// is all zeros when the handle is created, a call writes a slice only if one of its tokens has a posting in it or the
// previous call left something there (rare-term queries touch a few slices of the 80 MB at 10M documents, not all).
__global__ __launch_bounds__(256) void rr_bm25_slices(rr_bm25_view v, const int32_t* __restrict__ terms,
                                                      int n_terms_q, double* __restrict__ out, unsigned char* __restrict__ dirty) {
    __shared__ double acc[RR_SLICE];
    __shared__ int64_t range[RR_SLICE_TERMS][2];
    __shared__ int touched;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t d0 = (int64_t)blockIdx.x * RR_SLICE;
    const int64_t d1 = d0 + RR_SLICE < v.n_docs ? d0 + RR_SLICE : v.n_docs;
    for (int i = tid; i < RR_SLICE; i += 256) acc[i] = 0.0;
    if (tid == 0) touched = 0;
    for (int j0 = 0; j0 < n_terms_q; j0 += RR_SLICE_TERMS) {
        const int nj = n_terms_q - j0 < RR_SLICE_TERMS ? n_terms_q - j0 : RR_SLICE_TERMS;
        __syncthreads();                                     // (the previous chunk's ranges have been used)
        // the block of every token's postings that falls into [d0, d1): 2 nj searches, one per wave at a time
        for (int sidx = wave; sidx < 2 * nj; sidx += 4) {
            const int32_t t = terms[j0 + (sidx >> 1)];
            int64_t r = 0;
            if (t >= 0 && t < v.n_terms)                     // (idf.get(t) -> 0 otherwise: an empty block, contributes +0.0)
                r = rr_lower_bound_wave(v.post_docs, v.post_indptr[t], v.post_indptr[t + 1], (int32_t)((sidx & 1) ? d1 : d0));
            if ((tid & 63) == 0) range[sidx >> 1][sidx & 1] = r;
        }
        __syncthreads();
        // tokens in query order (a document's additions are the reference's); the first 256 postings of the NEXT token's
        // block are fetched while this one's are added
        int32_t nd = 0, ntf = 0, ndl = 0;
        bool nhave = false;
        {
            const int64_t p = range[0][0] + tid;
            nhave = p < range[0][1];
            if (nhave) { nd = v.post_docs[p]; ntf = v.post_tf[p]; ndl = v.doc_len[nd]; }
        }
        for (int j = 0; j < nj; ++j) {
            const int64_t lo = range[j][0], hi = range[j][1];
            const int32_t t = terms[j0 + j];
            const bool have = nhave;
            const int32_t d = nd, tf = ntf, dl = ndl;
            if (j + 1 < nj) {
                const int64_t p = range[j + 1][0] + tid;
                nhave = p < range[j + 1][1];
                if (nhave) { nd = v.post_docs[p]; ntf = v.post_tf[p]; ndl = v.doc_len[nd]; }
            }
            if (hi > lo) {
                const double idf = v.idf[t];
                if (tid == 0) touched = 1;
                if (have) acc[d - d0] += rr_bm25_term(idf, tf, dl, v.avgdl, v.k1, v.b);
                for (int64_t p = lo + 256 + tid; p < hi; p += 256) {
                    const int32_t dd = v.post_docs[p];
                    acc[dd - d0] += rr_bm25_term(idf, v.post_tf[p], v.doc_len[dd], v.avgdl, v.k1, v.b);
                }
            }
            __syncthreads();                                 // a document takes one addition per token: tokens in order
        }
    }
    __syncthreads();
    const bool write = touched || dirty[blockIdx.x];
    if (write)
        for (int64_t i = tid; i < d1 - d0; i += 256) out[d0 + i] = acc[i];
    if (tid == 0) dirty[blockIdx.x] = (unsigned char)touched;
}

template <int MODE>
__global__ __launch_bounds__(256) void rr_bm25_at(rr_bm25_view v, const int32_t* __restrict__ term_ids,
                                                  const int32_t* __restrict__ term_off,
                                                  const int64_t* __restrict__ rows, int pool,
                                                  float* __restrict__ out) {
    __shared__ double contrib[RR_AT_CANDS][RR_MAX_QTERMS + 1];
    const int tid = threadIdx.x;
    const int q = blockIdx.y;
    const int c0 = blockIdx.x * RR_AT_CANDS;
    const int t0 = term_off[q];
    const int nt = term_off[q + 1] - t0;
    const int nc = pool - c0 < RR_AT_CANDS ? pool - c0 : RR_AT_CANDS;
    double sum = 0.0;                                 // candidate tid's running score (threads < nc)

    // tokens in chunks of RR_MAX_QTERMS: get_scores has no token limit (a pasted paragraph is a legal query);
    // the running sum keeps the additions in token order across chunks
    for (int jb = 0; jb < nt; jb += RR_MAX_QTERMS) {
        const int nj = nt - jb < RR_MAX_QTERMS ? nt - jb : RR_MAX_QTERMS;
        for (int i = tid; i < nc * nj; i += 256) {
            const int c = i / nj, j = i % nj;
            const int64_t d = rows[(int64_t)q * pool + c0 + c] - v.row_offset;
            const int32_t t = term_ids[t0 + jb + j];
            double val = 0.0;
            if (d >= 0 && d < v.n_docs && t >= 0 && t < v.n_terms) {
                int tf = 0;
                if (MODE == 0) {
                    const int64_t s = v.doc_indptr[d], e = v.doc_indptr[d + 1];
                    const int64_t p = rr_lower_bound(v.doc_terms, s, e, t);
                    if (p < e && v.doc_terms[p] == t) tf = v.doc_tf[p];
                } else {
                    const int64_t s = v.post_indptr[t], e = v.post_indptr[t + 1];
                    const int64_t p = rr_lower_bound(v.post_docs, s, e, (int32_t)d);
                    if (p < e && v.post_docs[p] == (int32_t)d) tf = v.post_tf[p];
                }
                if (tf > 0) val = rr_bm25_term(v.idf[t], tf, v.doc_len[d], v.avgdl, v.k1, v.b);
            }
            contrib[c][j] = val;
        }
        __syncthreads();
        if (tid < nc)
            for (int j = 0; j < nj; ++j) sum += contrib[tid][j];   // token order, like get_scores
        __syncthreads();
    }
    if (tid < nc) out[(int64_t)q * pool + c0 + tid] = (float)sum;  // np.array(..., dtype=np.float32)
}

// ------------------------------------------------------------------ host side
template <typename T>
static int rr_upload(T** dst, const T* src, size_t n) {
    *dst = nullptr;
    RR_HIP_TRY(hipMalloc((void**)dst, sizeof(T) * (n ? n : 1)));
    if (n) RR_HIP_TRY(hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
    return RR_OK;
}

static rr_bm25_view rr_view(const rr_bm25* bm) {
    rr_bm25_view v;
    v.post_indptr = bm->d_post_indptr; v.post_docs = bm->d_post_docs; v.post_tf = bm->d_post_tf;
    v.doc_indptr = bm->d_doc_indptr; v.doc_terms = bm->d_doc_terms; v.doc_tf = bm->d_doc_tf;
    v.doc_len = bm->d_doc_len; v.idf = bm->d_idf;
    v.avgdl = bm->avgdl; v.k1 = bm->k1; v.b = bm->b;
    v.n_docs = bm->n_docs; v.n_terms = bm->n_terms; v.row_offset = bm->row_offset;
    return v;
}

extern "C" int rr_bm25_create(int32_t device, int64_t n_docs, int64_t n_terms, int64_t nnz,
                              const int64_t* post_indptr, const int32_t* post_docs,
                              const int32_t* post_tf, const int64_t* doc_indptr,
                              const int32_t* doc_terms, const int32_t* doc_tf,
                              const int32_t* doc_len, const double* idf, double avgdl, double k1,
                              double b, int64_t row_offset, rr_bm25** out) {
    RR_REQUIRE(out, "rr_bm25_create: NULL out");
    *out = nullptr;
    RR_REQUIRE(n_docs >= 1 && n_docs < (1ll << 31), "rr_bm25_create: n_docs %lld out of range",
               (long long)n_docs);
    RR_REQUIRE(n_terms >= 0 && n_terms < (1ll << 31) && nnz >= 0, "rr_bm25_create: bad sizes");
    RR_REQUIRE(post_indptr && doc_indptr && doc_len && (nnz == 0 || (post_docs && post_tf && doc_terms && doc_tf)) &&
                   (n_terms == 0 || idf),
               "rr_bm25_create: NULL array");
    RR_REQUIRE(post_indptr[n_terms] == nnz && doc_indptr[n_docs] == nnz,
               "rr_bm25_create: indptr tails (%lld, %lld) do not match nnz %lld",
               (long long)post_indptr[n_terms], (long long)doc_indptr[n_docs], (long long)nnz);
    RR_REQUIRE(avgdl > 0.0, "rr_bm25_create: avgdl must be positive");
    RR_HIP_TRY(hipSetDevice(device));
    rr_bm25* bm = new rr_bm25();
    bm->device = device; bm->n_docs = n_docs; bm->n_terms = n_terms; bm->nnz = nnz;
    bm->avgdl = avgdl; bm->k1 = k1; bm->b = b; bm->row_offset = row_offset;
    int rc = RR_OK;
    if (!rc) rc = rr_upload(&bm->d_post_indptr, post_indptr, (size_t)n_terms + 1);
    if (!rc) rc = rr_upload(&bm->d_post_docs, post_docs, (size_t)nnz);
    if (!rc) rc = rr_upload(&bm->d_post_tf, post_tf, (size_t)nnz);
    if (!rc) rc = rr_upload(&bm->d_doc_indptr, doc_indptr, (size_t)n_docs + 1);
    if (!rc) rc = rr_upload(&bm->d_doc_terms, doc_terms, (size_t)nnz);
    if (!rc) rc = rr_upload(&bm->d_doc_tf, doc_tf, (size_t)nnz);
    if (!rc) rc = rr_upload(&bm->d_doc_len, doc_len, (size_t)n_docs);
    if (!rc) rc = rr_upload(&bm->d_idf, idf, (size_t)n_terms);
    if (!rc && hipStreamCreateWithFlags(&bm->stream, hipStreamNonBlocking) != hipSuccess) {
        rr_set_error("rr_bm25_create: hipStreamCreate failed");
        rc = RR_E_HIP;
    }
    if (rc) { rr_bm25_destroy(bm); return rc; }
    *out = bm;
    return RR_OK;
}

extern "C" int rr_bm25_create_dev(int32_t device, int64_t n_docs, int64_t n_terms, int64_t nnz,
                                  const int64_t* d_post_indptr, const int32_t* d_post_docs,
                                  const int32_t* d_post_tf, const int64_t* d_doc_indptr,
                                  const int32_t* d_doc_terms, const int32_t* d_doc_tf,
                                  const int32_t* d_doc_len, const double* d_idf, double avgdl,
                                  double k1, double b, int64_t row_offset, rr_bm25** out) {
    RR_REQUIRE(out, "rr_bm25_create_dev: NULL out");
    *out = nullptr;
    RR_REQUIRE(n_docs >= 1 && n_docs < (1ll << 31) && n_terms >= 1 && n_terms < (1ll << 31) && nnz >= 1,
               "rr_bm25_create_dev: bad sizes");
    RR_REQUIRE(d_post_indptr && d_post_docs && d_post_tf && d_doc_indptr && d_doc_terms && d_doc_tf &&
                   d_doc_len && d_idf, "rr_bm25_create_dev: NULL array");
    RR_REQUIRE(avgdl > 0.0, "rr_bm25_create_dev: avgdl must be positive");
    RR_HIP_TRY(hipSetDevice(device));
    int64_t tails[2] = {-1, -1};
    RR_HIP_TRY(hipMemcpy(&tails[0], d_post_indptr + n_terms, sizeof(int64_t), hipMemcpyDeviceToHost));
    RR_HIP_TRY(hipMemcpy(&tails[1], d_doc_indptr + n_docs, sizeof(int64_t), hipMemcpyDeviceToHost));
    RR_REQUIRE(tails[0] == nnz && tails[1] == nnz, "rr_bm25_create_dev: indptr tails (%lld, %lld) != nnz %lld",
               (long long)tails[0], (long long)tails[1], (long long)nnz);
    rr_bm25* bm = new rr_bm25();
    bm->device = device; bm->n_docs = n_docs; bm->n_terms = n_terms; bm->nnz = nnz;
    bm->avgdl = avgdl; bm->k1 = k1; bm->b = b; bm->row_offset = row_offset;
    bm->owns_arrays = false;
    bm->d_post_indptr = const_cast<int64_t*>(d_post_indptr);
    bm->d_post_docs = const_cast<int32_t*>(d_post_docs);
    bm->d_post_tf = const_cast<int32_t*>(d_post_tf);
    bm->d_doc_indptr = const_cast<int64_t*>(d_doc_indptr);
    bm->d_doc_terms = const_cast<int32_t*>(d_doc_terms);
    bm->d_doc_tf = const_cast<int32_t*>(d_doc_tf);
    bm->d_doc_len = const_cast<int32_t*>(d_doc_len);
    bm->d_idf = const_cast<double*>(d_idf);
    if (hipStreamCreateWithFlags(&bm->stream, hipStreamNonBlocking) != hipSuccess) {
        rr_set_error("rr_bm25_create_dev: hipStreamCreate failed");
        delete bm;
        return RR_E_HIP;
    }
    *out = bm;
    return RR_OK;
}

extern "C" int rr_bm25_destroy(rr_bm25* bm) {
    if (!bm) return RR_OK;
    hipSetDevice(bm->device);
    if (bm->owns_arrays) {
        hipFree(bm->d_post_indptr); hipFree(bm->d_post_docs); hipFree(bm->d_post_tf);
        hipFree(bm->d_doc_indptr); hipFree(bm->d_doc_terms); hipFree(bm->d_doc_tf);
        hipFree(bm->d_doc_len); hipFree(bm->d_idf);
    }
    hipFree(bm->d_scores);
    hipFree(bm->d_dirty);
    if (bm->stream) hipStreamDestroy(bm->stream);
    delete bm;
    return RR_OK;
}

extern "C" int rr_bm25_get_scores(rr_bm25* bm, const int32_t* h_term_ids, int32_t n_terms_in_query,
                                  double* h_out_scores) {
    RR_REQUIRE(bm && h_out_scores && (n_terms_in_query == 0 || h_term_ids),
               "rr_bm25_get_scores: NULL argument");
    RR_REQUIRE(n_terms_in_query >= 0 && n_terms_in_query <= 4096,
               "rr_bm25_get_scores: %d query tokens out of [0,4096]", n_terms_in_query);
    std::lock_guard<std::mutex> lk(bm->mu);
    RR_HIP_TRY(hipSetDevice(bm->device));
    const unsigned grid = (unsigned)((bm->n_docs + RR_SLICE - 1) / RR_SLICE);
    if (!bm->d_scores) {
        // all zeros once; from then on a call writes only the slices it (or its predecessor) put something in
        RR_HIP_TRY(hipMalloc((void**)&bm->d_scores, sizeof(double) * (size_t)bm->n_docs));
        RR_HIP_TRY(hipMalloc((void**)&bm->d_dirty, grid));
        RR_HIP_TRY(hipMemsetAsync(bm->d_scores, 0, sizeof(double) * (size_t)bm->n_docs, bm->stream));
        RR_HIP_TRY(hipMemsetAsync(bm->d_dirty, 0, grid, bm->stream));
    }
    int32_t* d_terms = nullptr;
    RR_HIP_TRY(hipMalloc((void**)&d_terms, sizeof(int32_t) * (size_t)(n_terms_in_query + 1)));
    if (n_terms_in_query)
        hipMemcpyAsync(d_terms, h_term_ids, sizeof(int32_t) * n_terms_in_query, hipMemcpyHostToDevice,
                       bm->stream);
    hipLaunchKernelGGL(rr_bm25_slices, dim3(grid), dim3(256), 0, bm->stream, rr_view(bm), d_terms,
                       n_terms_in_query, bm->d_scores, bm->d_dirty);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(h_out_scores, bm->d_scores, sizeof(double) * (size_t)bm->n_docs,
                           hipMemcpyDeviceToHost, bm->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(bm->stream);
    hipFree(d_terms);
    if (e != hipSuccess) {
        rr_set_error("rr_bm25_get_scores: %s", hipGetErrorString(e));
        return RR_E_HIP;
    }
    return RR_OK;
}

extern "C" int rr_bm25_scores_at_dev(rr_bm25* bm, const int32_t* d_term_ids,
                                     const int32_t* d_term_off, int32_t n_queries,
                                     const int64_t* d_rows, int32_t pool, int32_t mode,
                                     float* d_out, void* stream) {
    RR_REQUIRE(bm && d_term_ids && d_term_off && d_rows && d_out, "rr_bm25_scores_at_dev: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH && pool >= 1 && pool <= 8 * RR_MAX_POOL,
               "rr_bm25_scores_at_dev: n_queries %d / pool %d out of range", n_queries, pool);
    RR_REQUIRE(mode == 0 || mode == 1, "rr_bm25_scores_at_dev: mode must be 0 (forward) or 1 (postings)");
    RR_HIP_TRY(hipSetDevice(bm->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the device's default stream
    dim3 grid((unsigned)((pool + RR_AT_CANDS - 1) / RR_AT_CANDS), (unsigned)n_queries);
    if (mode == 0)
        hipLaunchKernelGGL((rr_bm25_at<0>), grid, dim3(256), 0, st, rr_view(bm), d_term_ids, d_term_off,
                           d_rows, pool, d_out);
    else
        hipLaunchKernelGGL((rr_bm25_at<1>), grid, dim3(256), 0, st, rr_view(bm), d_term_ids, d_term_off,
                           d_rows, pool, d_out);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

extern "C" int rr_bm25_scores_at(rr_bm25* bm, const int32_t* h_term_ids, const int32_t* h_term_off,
                                 int32_t n_queries, const int64_t* h_rows, int32_t pool,
                                 int32_t mode, float* h_out) {
    RR_REQUIRE(bm && h_term_off && h_rows && h_out, "rr_bm25_scores_at: NULL argument");
    RR_REQUIRE(n_queries >= 1 && n_queries <= RR_MAX_BATCH && pool >= 1 && pool <= 8 * RR_MAX_POOL,
               "rr_bm25_scores_at: n_queries %d / pool %d out of range", n_queries, pool);
    const int total_terms = h_term_off[n_queries];
    RR_REQUIRE(total_terms >= 0 && (total_terms == 0 || h_term_ids), "rr_bm25_scores_at: bad term arrays");
    for (int q = 0; q < n_queries; ++q)
        RR_REQUIRE(h_term_off[q + 1] - h_term_off[q] >= 0, "rr_bm25_scores_at: term offsets must not decrease (query %d)", q);
    std::lock_guard<std::mutex> lk(bm->mu);
    RR_HIP_TRY(hipSetDevice(bm->device));
    int32_t *d_ids = nullptr, *d_off = nullptr;
    int64_t* d_rows = nullptr;
    float* d_out = nullptr;
    const size_t n_out = (size_t)n_queries * pool;
    hipError_t e = hipMalloc((void**)&d_ids, sizeof(int32_t) * (size_t)(total_terms + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&d_off, sizeof(int32_t) * (size_t)(n_queries + 1));
    if (e == hipSuccess) e = hipMalloc((void**)&d_rows, sizeof(int64_t) * n_out);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, sizeof(float) * n_out);
    if (e == hipSuccess && total_terms)
        e = hipMemcpyAsync(d_ids, h_term_ids, sizeof(int32_t) * total_terms, hipMemcpyHostToDevice, bm->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_off, h_term_off, sizeof(int32_t) * (n_queries + 1), hipMemcpyHostToDevice, bm->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_rows, h_rows, sizeof(int64_t) * n_out, hipMemcpyHostToDevice, bm->stream);
    int rc = RR_OK;
    if (e == hipSuccess)
        rc = rr_bm25_scores_at_dev(bm, d_ids, d_off, n_queries, d_rows, pool, mode, d_out, bm->stream);
    if (e == hipSuccess && rc == RR_OK)
        e = hipMemcpyAsync(h_out, d_out, sizeof(float) * n_out, hipMemcpyDeviceToHost, bm->stream);
    if (e == hipSuccess && rc == RR_OK) e = hipStreamSynchronize(bm->stream);
    hipFree(d_ids); hipFree(d_off); hipFree(d_rows); hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) {
        rr_set_error("rr_bm25_scores_at: %s", hipGetErrorString(e));
        return RR_E_HIP;
    }
    return RR_OK;
}
