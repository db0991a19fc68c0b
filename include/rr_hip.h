/*
 * rr_hip.h -- C ABI of librr_hip.so, the MI355X (gfx950) hybrid-retrieval hot path.
 *
 * The reference (Ntropy86/review-recommender) is pure Python and has no FFI of
 * its own; its boundary for this path is a set of Python callables (SURVEY.md
 * section 8b).  Each entry point below names the reference callable it stands
 * in for.  A Python front-end binds these with ctypes (INTEGRATION.md shows
 * the stub); the product wrapper is review-recommender_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, a negative RR_E_* code otherwise;
 *     rr_last_error() returns a thread-local message for the last failure;
 *   - the caller owns every host buffer (C-contiguous, sizes as documented);
 *     the library owns device memory behind opaque handles;
 *   - "h_" parameters are host pointers, "d_" parameters are device pointers
 *     on the handle's device (e.g. a torch tensor's data_ptr(), used by the
 *     multi-GPU exchange so RCCL can move the buffers without a host hop);
 *   - rows are int64 indices into the index's row space; a shard created with
 *     a row offset reports GLOBAL rows (local row + offset);
 *   - handles may be used from several host threads; calls on one handle are
 *     serialised by an internal mutex (Streamlit shares cached handles across
 *     session threads: app/app_product_search.py:53,71,119).
 *   - there is no CPU fallback anywhere behind this ABI.
 */
#ifndef RR_HIP_H
#define RR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_OK            0
#define RR_E_INVALID    -1   /* bad argument (shape, NULL, range) */
#define RR_E_HIP        -2   /* a HIP runtime call failed */
#define RR_E_NOMEM      -3
#define RR_E_STATE      -4   /* handle not ready for this call (e.g. meta not set) */

#define RR_DTYPE_F32     0
#define RR_DTYPE_BF16    1   /* storage only (dim 384): queries, products and accumulation stay fp32 */

#define RR_MAX_POOL   2048   /* upper bound for pool / k */
#define RR_MAX_BATCH  1024
#define RR_MAX_QTERMS   64   /* query tokens the BM25 candidate kernel stages per pass (longer queries take several passes) */

typedef struct rr_index rr_index;   /* dense matrix + per-row metadata of one shard */
typedef struct rr_bm25  rr_bm25;    /* BM25 postings (CSR by term) + forward CSR by doc */

const char* rr_last_error(void);
int rr_version(void);
int rr_device_count(int* out);

/* ------------------------------------------------------------------ index */

/* Device-resident copy of product_emb.npy (nlp/11_build_product_embeddings.py:82-85):
 * n_rows x dim, row-major.  h_matrix may be NULL (fill later with
 * rr_index_upload_rows / rr_index_adopt_device).  row_offset is added to every
 * row this shard reports.  dim is padded internally to a multiple of 64. */
int rr_index_create(const void* h_matrix, int64_t n_rows, int32_t dim, int32_t dtype,
                    int32_t device, int64_t row_offset, rr_index** out);
int rr_index_upload_rows(rr_index* ix, int64_t first_row, int64_t n_rows, const void* h_rows);
/* fp32 host rows into an index of either dtype: rows are optionally l2-normalised in fp32
 * (normalize_eps > 0: utils.py:40-44) and then stored; a bf16 index rounds them once, to nearest
 * even, AFTER the normalisation ("round V (RNE) to bf16 once"). */
int rr_index_upload_rows_f32(rr_index* ix, int64_t first_row, int64_t n_rows, const float* h_rows,
                             float normalize_eps);
/* Use a caller-owned device matrix (n_rows x dim_padded(), already padded) without copying.  The index keeps facts
 * derived from the matrix content (row-norm bounds of the filter scan, the bf16 filter plane): writes made through this
 * ABI invalidate them; after writing to an ADOPTED matrix yourself (e.g. an in-place torch update) call
 * rr_index_matrix_changed, or batched searches filter on the old rows. */
int rr_index_adopt_device(rr_index* ix, const void* d_matrix);
int rr_index_matrix_changed(rr_index* ix);
int rr_index_dim_padded(const rr_index* ix, int32_t* out);
/* l2_normalize (utils.py:40-44) of every row, in place on the device. */
int rr_index_l2_normalize(rr_index* ix, float eps);
/* n_reviews / avg_stars of product_emb_meta.parquet (nlp/11...:86-90), row-aligned,
 * plus log1p(n_reviews) as numpy computes it (kept so the volume prior and trust
 * reproduce the reference bit for bit).  avg_stars may hold NaN. */
int rr_index_set_meta(rr_index* ix, const double* h_n_reviews, const double* h_avg_stars,
                      const double* h_log1p_n);
int rr_index_destroy(rr_index* ix);

/* ------------------------------------------------------------ K1 dense top-k */

/* cosine_similarity_search (utils.py:111-124; _cosine_pool app/app_product_search.py:192-195;
 * cosine_search app/test.py:125-132), batched: for each of n_queries query vectors
 * (n_queries x dim fp32, host) the top `pool` rows by dot product, ordered
 * (score desc, row asc); pool is clamped to n_rows like utils.py:116-117 and the
 * clamped value is written to *pool_out.  out_rows / out_scores: n_queries x pool. */
int rr_dense_topk(rr_index* ix, const float* h_queries, int32_t n_queries, int32_t pool,
                  int64_t* h_out_rows, float* h_out_scores, int32_t* pool_out);
/* Same with device-resident queries and outputs, asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the device's default stream, which is what
 * torch.cuda.current_stream().cuda_stream is unless the caller switched streams). */
int rr_dense_topk_dev(rr_index* ix, const float* d_queries, int32_t n_queries, int32_t pool,
                      int64_t* d_out_rows, float* d_out_scores, void* stream);
/* Timing of the last dense scan kernel on this handle (HIP events, ms). */
int rr_index_last_scan_ms(rr_index* ix, float* out_ms);
/* Every scan launch is bracketed by a HIP event pair on its stream (ring of 512).
 * Drains the pairs recorded since the last call: total ms and launch count. */
int rr_index_scan_stats(rr_index* ix, double* out_total_ms, int64_t* out_launches);
/* Which scan kernel the last scan launch on this handle ran: out8[0] = 1 rr_scan_f32, 2 rr_scan_bf16,
 * 3 rr_scan_mfma_x3, 4 rr_scan_x3w, 5 the filter scans (rr_scan_flt over fp32 rows; over a bf16 stream rr_scan_flt16
 * and, for two query sets in one launch, rr_scan_fltq), 6 rr_scan_mfma_f32, 7 rr_scan_mfma_bf16; [1] = its template
 * variant (query tiles / query slots; filter scans: 8 = rr_scan_flt16 serving two query sets, 9 = rr_scan_fltq);
 * [2] = queries in that launch; [3] = bf16 MFMA terms per dimension
 * (0: not a bf16 matrix-core kernel); [4] = bytes per matrix element the scan streamed (2 = bf16 rows or the bf16
 * filter plane, 4 = fp32 rows).  bench.py names the roofline kernel and its algorithmic bytes from this. */
int rr_index_last_scan_info(rr_index* ix, int32_t* out8);
/* Path taken by the last top-pool selection for its first query: out16[0] = 2: M-tile maxima + rescoring
 * (the batched paths), 1: stored scores through the LDS-resident 3-level selection, 0: generic radix
 * fallback; [1..3] = groups opened, (M-)tiles opened, candidate rows; [4..] = shader-clock cycles of the
 * selection's phases (-1: unused).  Diagnostic: lets tests assert which path ran. */
int rr_index_select_trace(rr_index* ix, int32_t* out16);
/* Diagnostic switch of the batched (5..64 query) scan.  RR_SCAN_MODE_DEFAULT: the scan keeps only
 * M-tile maxima and the candidate M-tiles are rescored (select trace out16[0] == 2), with the
 * stored-score pass as per-query fallback.  RR_SCAN_MODE_STORED: always the single pass that
 * stores every score.  Both return identical rows and scores; tests compare them. */
/* The bf16 filter plane of an fp32 index (default on): the batched filter scan (5..128 queries per launch) streams a
 * once-rounded bf16 copy of the matrix (n_rows x 384 x 2 bytes of extra HBM, built lazily at the first batched search
 * and dropped by any write to the matrix) instead of the fp32 rows -- half the bytes per launch for the same approximate
 * scores and the same error bound -- and candidates are rescored on the fp32 rows exactly as before: identical answers.
 * enable = 0 frees the plane and scans the fp32 rows.  Environment RR_NO_SHADOW=1 does the same process-wide. */
int rr_index_set_shadow(rr_index* ix, int32_t enable);
#define RR_SCAN_MODE_DEFAULT 0
#define RR_SCAN_MODE_STORED 1
int rr_index_set_scan_mode(rr_index* ix, int32_t mode);

/* ------------------------------------------------------------ K2 BM25 */

/* BM25Okapi(corpus) of rank_bm25 as used at app/app_product_search.py:142 and
 * app/test.py:156, over integer term ids.  Inputs (host):
 *   post_indptr[n_terms+1], post_docs[nnz] (ascending per term), post_tf[nnz]
 *   doc_indptr[n_docs+1],  doc_terms[nnz] (ascending per doc),  doc_tf[nnz]
 *   doc_len[n_docs] (token count incl. duplicates), idf[n_terms] (float64, with
 *   the epsilon floor already applied), avgdl, k1, b.
 * doc ids are LOCAL rows of the shard; idf / avgdl are corpus-wide. */
int rr_bm25_create(int32_t device, int64_t n_docs, int64_t n_terms, int64_t nnz,
                   const int64_t* post_indptr, const int32_t* post_docs, const int32_t* post_tf,
                   const int64_t* doc_indptr, const int32_t* doc_terms, const int32_t* doc_tf,
                   const int32_t* doc_len, const double* idf, double avgdl, double k1, double b,
                   int64_t row_offset, rr_bm25** out);
/* Same, with every array already resident on `device` (caller-owned, not copied, must
 * outlive the handle): used when the corpus is built on the GPU. */
int rr_bm25_create_dev(int32_t device, int64_t n_docs, int64_t n_terms, int64_t nnz,
                       const int64_t* d_post_indptr, const int32_t* d_post_docs, const int32_t* d_post_tf,
                       const int64_t* d_doc_indptr, const int32_t* d_doc_terms, const int32_t* d_doc_tf,
                       const int32_t* d_doc_len, const double* d_idf, double avgdl, double k1, double b,
                       int64_t row_offset, rr_bm25** out);
int rr_bm25_destroy(rr_bm25* bm);
/* BM25Okapi.get_scores(tokens) (app/app_product_search.py:206): float64[n_docs],
 * tokens applied in order, duplicates counted again, ids < 0 = unknown token. */
int rr_bm25_get_scores(rr_bm25* bm, const int32_t* h_term_ids, int32_t n_terms_in_query,
                       double* h_out_scores);
/* _bm25_for_candidates (app/app_product_search.py:201-208) / bm25_scores (app/test.py:168-173)
 * without the N-long intermediate: BM25 of each query at its candidate rows only.
 *   h_term_ids[term_off[q] .. term_off[q+1]) are query q's token ids;
 *   rows: n_queries x pool GLOBAL rows (rows outside this shard score 0);
 *   out : n_queries x pool float32 (float64 accumulate, cast like np.float32).
 * mode 0 = forward CSR (doc -> terms), mode 1 = postings lists (binary search). */
int rr_bm25_scores_at(rr_bm25* bm, const int32_t* h_term_ids, const int32_t* h_term_off,
                      int32_t n_queries, const int64_t* h_rows, int32_t pool, int32_t mode,
                      float* h_out);
int rr_bm25_scores_at_dev(rr_bm25* bm, const int32_t* d_term_ids, const int32_t* d_term_off,
                          int32_t n_queries, const int64_t* d_rows, int32_t pool, int32_t mode,
                          float* d_out, void* stream);

/* ------------------------------------------------------------ K3 fuse + top-k */

typedef struct rr_fuse_params {
    double w_dense, w_bm25, w_rerank, w_prior, w_best;  /* run_search weights */
    double prior_C;          /* Bayesian prior strength (prior_C) */
    int32_t min_reviews;     /* trust ramp (min_reviews) */
    int32_t trust_sat;       /* 80 in the app (app/app_product_search.py:303) */
    int32_t apply_trust;     /* 1 = app flavour, 0 = CLI flavour (app/test.py:308) */
    int32_t rerank_active;   /* 1 when rerank_k > 0: _rerank column is float32 */
    int32_t rerank_k;        /* first rerank_k pool rows carry reranker scores */
    int32_t k;               /* rows to return */
    int32_t n_candidates;    /* candidates supplied per query (>= pool when merging shards) */
    int32_t pool;            /* pool size to cut the candidates to before fusing */
    int32_t cand_per_rank;   /* 0: candidate arrays are [query][n_candidates]; else the arrays are
                                gathered shard payloads [rank][query][cand_per_rank] */
    int32_t bm25_f64;        /* 1: the _bm25 column is the CLI's float64 zeros (`cand["_bm25"] = 0.0`, app/test.py:252:
                                no BM25 artefact): the blend is float64 from its second term on; d_bm25 must be NULL */
    int64_t cand_rank_stride_bytes; /* distance between two ranks' payload blocks */
} rr_fuse_params;

/* Everything run_search does after the candidate pool exists
 * (app/app_product_search.py:256-312; CLI app/test.py:250-309): cut the
 * n_candidates supplied per query to the best `pool` by (dense desc, row asc)
 * [the shard merge], min-max of dense / bm25 / prior / rerank / best, Bayesian
 * prior with the pool mean, volume prior, trust, weighted blend, gate, stable
 * sort by final desc, first k.
 * Candidate inputs, n_candidates per query (device pointers): rows, dense, bm25 raw
 * (NULL = zeros), n_reviews, avg_stars, log1p_n (all three NULL = gathered from `ix`
 * by row; required when merging shards).  Pool inputs, `pool` per query, aligned
 * with the pool order this call produces (so a caller that needs them first calls
 * with k = pool and reads out_rows): rerank raw, best raw (NULL = zeros), gate
 * (NULL = ones).
 * Outputs (device): out_rows[nq x pool] global rows in pool order;
 * out_cols[nq x 8 x pool] = _dense,_bm25,_prior,_rerank,_best,_gate,_trust,_final (float64);
 * out_order[nq x k] = pool positions of the top-k, best first. */
int rr_fuse_topk_dev(rr_index* ix, const rr_fuse_params* p, int32_t n_queries,
                     const int64_t* d_rows, const float* d_dense, const float* d_bm25,
                     const double* d_n_reviews, const double* d_avg_stars, const double* d_log1p_n,
                     const float* d_rerank, const float* d_best, const float* d_gate,
                     int64_t* d_out_rows, double* d_out_cols, int32_t* d_out_order, void* stream);
/* Host-buffer form of the same call. */
int rr_fuse_topk(rr_index* ix, const rr_fuse_params* p, int32_t n_queries,
                 const int64_t* h_rows, const float* h_dense, const float* h_bm25,
                 const double* h_n_reviews, const double* h_avg_stars, const double* h_log1p_n,
                 const float* h_rerank, const float* h_best, const float* h_gate,
                 int64_t* h_out_rows, double* h_out_cols, int32_t* h_out_order);
/* Gather per-row metadata for candidate rows (payload of the shard exchange). */
int rr_index_gather_meta_dev(rr_index* ix, const int64_t* d_rows, int64_t n,
                             double* d_n_reviews, double* d_avg_stars, double* d_log1p_n,
                             void* stream);

/* Kernel-driven copies: ONE launch moves up to RR_COPY_MAX_SEGS pitched segments (rows x row_bytes, each side with its
 * own pitch) on `stream`.  Either side of a segment may be PINNED HOST memory that is mapped into the device's address
 * space (hipHostMalloc, torch's pin_memory(): the host pointer is valid on the device): a batch's inputs (token ids)
 * and its answer (rows / order / final scores: what the reference returns to its caller, app/app_product_search.py:312-317)
 * then cross PCIe as the loads / stores of one kernel instead of as one copy command each.  In the same way
 * rr_dense_topk_dev accepts mapped pinned host memory as `d_queries` (its first kernel reads the queries once).
 * Not a product path of its own: the data movement in front of K1 / behind K3. */
#define RR_COPY_MAX_SEGS 4
typedef struct rr_copy_seg {
    void* dst;
    const void* src;
    int64_t row_bytes, rows, dst_pitch, src_pitch;
} rr_copy_seg;
int rr_copy_segments_dev(const rr_copy_seg* segs, int32_t n_segs, int32_t device, void* stream);

/* ------------------------------------------------------------ best review per candidate */

typedef struct rr_reviews rr_reviews;
/* Review embeddings of reviews_with_embeddings.parquet (n_reviews x dim fp32, l2-normalised on the
 * device when normalize_eps > 0 like _best_snippets does, app/app_product_search.py:347-348), grouped
 * by product: reviews of product row p are h_ids[h_indptr[p] .. h_indptr[p+1]), ascending (file order). */
int rr_reviews_create(const float* h_emb, int64_t n_reviews, int32_t dim, int64_t n_products,
                      const int64_t* h_indptr, const int32_t* h_ids, int32_t device,
                      float normalize_eps, rr_reviews** out);
int rr_reviews_destroy(rr_reviews* rv);
/* _best_snippets (app/app_product_search.py:320-370) for the candidates of each query: the review of
 * product rows[q][c] with the largest dot product with query q (first maximum in file order); reviews
 * with id > max_review_id are ignored (the max_rows cut of :342-345).  Outputs n_queries x pool:
 * best score (0 when the product has no review) and best review id (-1 likewise). */
int rr_reviews_best_dev(rr_reviews* rv, const float* d_queries, int32_t n_queries,
                        const int64_t* d_rows, int32_t pool, int64_t row_offset, int32_t max_review_id,
                        float* d_best_score, int32_t* d_best_id, void* stream);

/* The same for a whole batch with the reference's `iloc[:max_rows]` cut (app/app_product_search.py:342-345,
 * app/test.py:200-203) evaluated per query ON THE DEVICE: among the reviews of query q's candidates, in file
 * (= review id) order, only the first max_rows are scored; max_rows <= 0 scores none.  No host round trip. */
int rr_reviews_best_cut_dev(rr_reviews* rv, const float* d_queries, int32_t n_queries,
                            const int64_t* d_rows, int32_t pool, int64_t row_offset, int64_t max_rows,
                            float* d_best_score, int32_t* d_best_id, void* stream);

/* ------------------------------------------------------------ K5 cross-encoder / query encoder forward */

/* BERT-family encoder on the matrix cores (csrc/rr_ce.hip).  Stands in for
 *   CrossEncoder.predict(pairs, batch_size=64, show_progress_bar=False)   app/app_product_search.py:277-278, app/test.py:223-225
 *     (sentence-transformers wraps AutoModelForSequenceClassification; for cross-encoder/ms-marco-MiniLM-L-6-v2 that is
 *      BertForSequenceClassification: 6 layers, hidden 384, 12 heads, FFN 1536, 512 positions, 1 label)
 *   SentenceTransformer.encode([query], normalize_embeddings=True)        app/app_product_search.py:250-251, app/test.py:232
 *     (BAAI/bge-small-en-v1.5: BertModel of the same block shape, 12 layers, CLS pooling; the l2 normalisation is the caller's)
 * The kernels are built for hidden 384 / 12 heads x 32 / FFN 1536; layer count, vocabulary, positions (<= 512) are free.
 * Tokenisation (WordPiece) is host work above this ABI (review-recommender_amd/wordpiece.py). */
typedef struct rr_ce rr_ce;
typedef struct rr_ce_config {
    int32_t hidden, n_layers, n_heads, ffn;       /* 384, L, 12, 1536 */
    int32_t vocab, max_pos, type_vocab;           /* embedding table sizes */
    int32_t n_labels;                             /* classifier outputs (1 for the reranker); 0 = no pooler / classifier */
    float ln_eps;                                 /* 1e-12 for BERT */
    int32_t precision;                            /* RR_CE_PRECISION_BF16 (fast path) | RR_CE_PRECISION_F32 (the reference's arithmetic) */
} rr_ce_config;
/* RR_CE_PRECISION_BF16: Linear weights rounded once to bf16, bf16 MFMA operands, fp32 accumulation / residual / LayerNorm /
 * softmax / GELU: logits within 2.5e-2 of the fp32 reference on O(1) weights.  RR_CE_PRECISION_F32: what the reference runs
 * (fp32 torch, app/app_product_search.py:250-251, 277-278): fp32 weights and activations; every product on the fp16 matrix
 * cores with both operands carried as two fp16 numbers (hi + lo / 2048, three products per fp32 product: as exact as an
 * fp32 multiply-add chain, csrc/rr_ce_h2.hip), an fp32-grade GELU, base-2 softmax: logits and embeddings within 1e-5 of
 * `transformers` (tests/test_gpu_k5.py), ~2.7x the time of the bf16 mode.  Its range: rr_ce_range_status below. */
#define RR_CE_PRECISION_BF16 0
#define RR_CE_PRECISION_F32 1
/* Weights: fp32 host arrays in the layout of a Hugging Face BERT state dict (Linear weights are [out][in]), in this order:
 *   0 word_embeddings [vocab][H]   1 position_embeddings [max_pos][H]   2 token_type_embeddings [type_vocab][H]
 *   3 embeddings.LayerNorm.weight  4 embeddings.LayerNorm.bias
 *   per layer l, 16 tensors from 5 + 16 l: query.weight, query.bias, key.weight, key.bias, value.weight, value.bias,
 *     attention.output.dense.weight, .bias, attention.output.LayerNorm.weight, .bias,
 *     intermediate.dense.weight [FFN][H], .bias, output.dense.weight [H][FFN], .bias, output.LayerNorm.weight, .bias
 *   when n_labels > 0, four more: pooler.dense.weight [H][H], pooler.dense.bias, classifier.weight [n_labels][H], classifier.bias
 * Linear weights are rounded once to bf16 (nearest even) on the device (RR_CE_PRECISION_F32 also keeps them as given, and
 * as fp16 pairs);
 * everything else stays fp32. */
int rr_ce_create(int32_t device, const rr_ce_config* cfg, const float* const* h_tensors, int32_t n_tensors, rr_ce** out);
int rr_ce_destroy(rr_ce* ce);
#define RR_CE_OUT_LOGITS 0   /* d_out [n_seqs][n_labels]: classifier(tanh(pooler([CLS]))), raw logits (no activation) */
#define RR_CE_OUT_CLS    1   /* d_out [n_seqs][hidden]: last_hidden_state[:, 0] (CLS pooling of the query encoder) */
#define RR_CE_OUT_HIDDEN 2   /* d_out [n_tokens][hidden]: last_hidden_state of every token (diagnostic / parity tests) */
/* Forward over PACKED sequences (no padding is computed): sequence s owns tokens [cu_seqlens[s], cu_seqlens[s+1]);
 * token / type / position ids are per token (position = index inside its sequence); max_len = longest sequence.
 * Attention is full inside a sequence (what an all-ones attention mask over the unpadded tokens gives).
 * Asynchronous on `stream`; all pointers are device pointers. */
int rr_ce_forward_dev(rr_ce* ce, const int32_t* d_token_ids, const int32_t* d_type_ids, const int32_t* d_pos_ids,
                      const int32_t* d_cu_seqlens, int32_t n_seqs, int64_t n_tokens, int32_t max_len, int32_t mode,
                      float* d_out, void* stream);
/* HIP-event time of the last forward pass on this handle (ms). */
int rr_ce_last_forward_ms(rr_ce* ce, float* out_ms);
/* RR_CE_PRECISION_F32 multiplies on the fp16 matrix cores: every fp32 operand as hi + lo / 2048 in two fp16 numbers, three
 * products per fp32 product (csrc/rr_ce_h2.hip; as exact as an fp32 multiply-add chain).  fp16 ends at 65504: a forward pass
 * that meets a larger activation raises a flag on the device, writes NaN logits / CLS rows for the whole call (never a wrong
 * finite number; RR_CE_OUT_HIDDEN rows are left as computed: ask here) and reports it here -- *out_of_range = 1 -- once
 * the pass has finished (this call waits for it).
 * rr_ce_set_wide_range(ce, 1) switches the handle to three bf16 terms per operand and six products (any fp32 range,
 * ~1.6 x the time): run the pass again after it.  cross_encoder.py does both by itself on the host path. */
int rr_ce_range_status(rr_ce* ce, int32_t* out_of_range);
int rr_ce_set_wide_range(rr_ce* ce, int32_t on);

/* Two-phase K1 for ROW SHARDS (sharded.py: one process per GPU, this shard's rows in `ix`).  A shard's
 * own top-`top_k` threshold sits far below the corpus-wide one (rank 150 of 1.25M rows ~ rank 1 200 of 10M), so a shard
 * that selects on its own rescoring ~8x the candidates the merged answer needs.  Instead:
 *   1. rr_dense_scan_dev: the scan of rr_dense_topk_dev without its selection; d_bound[q] (device, n_queries floats) =
 *      a lower bound of the score of this shard's `kth`-best row (kth = ceil(top_k / shards)), or -inf;
 *   2. the caller takes the MINIMUM of d_bound over the shards (one all-reduce of n_queries floats): the union of the
 *      shards' kth best rows holds >= top_k rows, so that minimum is a lower bound of the corpus-wide top_k-th best score;
 *   3. rr_dense_select_dev: the selection, opening nothing that cannot reach that floor.  Its lists still hold top_k
 *      rows per query -- every row of the corpus-wide top-k that lives in this shard, filled up with other exactly
 *      scored rows of the shard (not necessarily its next best): only the MERGED top-k of all shards is the answer.
 * `*applied` (host) = 0 when the call cannot be split (fewer than 5 or more than 256 queries, 129..192, a small or
 * non-finite matrix, a pinned scan mode): nothing was launched, d_bound is untouched, use rr_dense_topk_dev.  The
 * queries, n_queries and top_k of phase 2 must be those of phase 1, with no other search on `ix` in between.
 * (Replaces nothing in the reference: utils.py:111-124 is single-process.) */
int rr_dense_scan_dev(rr_index* ix, const float* d_queries, int32_t n_queries, int32_t top_k, int32_t kth,
                      float* d_bound, int32_t* applied, void* stream);
int rr_dense_select_dev(rr_index* ix, const float* d_queries, int32_t n_queries, int32_t top_k,
                        const float* d_floor, int64_t* d_out_rows, float* d_out_scores, void* stream);

/* PIPELINED K1 ("keep K1 -> K2-gather -> allgather -> K3 on-stream ... pipeline batches"): the same two
 * phases with an explicit SCAN SLOT (0, 1 or 2).  A slot holds what one batch's scan writes and its selection reads (staged
 * queries, their bf16 planes and error bounds, tile / group maxima, the parked phase-1 state); the index has three, so batch
 * i + 1 can be scanned into one slot on one stream while batch i's selection (then K2, K3 and the answer's copy) reads the
 * other on ANOTHER stream.  The library orders scan and selection of a slot with events (a scan waits for the slot's
 * previous selection, a selection for its scan and for the previous selection's use of the shared rescoring scratch);
 * calls on one handle stay serialised on the host.  kth = 0 with d_bound = NULL: no bound is computed (single GPU);
 * d_floor may be NULL (the shard selects on its own threshold).  rr_dense_scan_dev / rr_dense_select_dev above are
 * slot 0; every other search entry point uses slot 0 and voids a scan parked there.
 * The scan kernel keeps one 512-register wave on every SIMD of a CU it runs on, so a second stream only makes progress on
 * CUs the scan does not hold: give the scans a stream masked to part of the device (rr_stream_create_cu_range), tell the
 * index how many CUs that is (rr_index_set_scan_cus: every scan's resident grid is sized for it), and run the selections
 * on a stream masked to the rest.  (Replaces nothing in the reference: utils.py:111-124 is one blocking call.) */
int rr_dense_scan_slot_dev(rr_index* ix, int32_t slot, const float* d_queries, int32_t n_queries, int32_t top_k, int32_t kth,
                           float* d_bound, int32_t* applied, void* stream);
int rr_dense_select_slot_dev(rr_index* ix, int32_t slot, int32_t n_queries, int32_t top_k, const float* d_floor,
                             int64_t* d_out_rows, float* d_out_scores, void* stream);
/* rr_dense_topk_dev in a given scan slot (rr_dense_topk_dev itself = slot 0): the whole K1 of a batch the pipeline cannot
 * split (`applied` = 0), without disturbing the scans other batches have parked in the other slots. */
int rr_dense_topk_slot_dev(rr_index* ix, int32_t slot, const float* d_queries, int32_t n_queries, int32_t top_k,
                           int64_t* d_out_rows, float* d_out_scores, void* stream);
/* The selection in its three steps, each possibly on a stream of its own (`parts` = any combination, launched in this order):
 *   RR_SELECT_LIST     per query the 8-row M-tiles whose scan bound reaches the threshold (one workgroup per query: latency-
 *                      bound, a few CUs do);
 *   RR_SELECT_RESCORE  the exact per-row scores of the listed M-tiles: a gather of ~3 MB of rows per query (HBM-bound: wants
 *                      the bandwidth of the many CUs -- e.g. the scans' stream, behind the next batch's scan);
 *   RR_SELECT_ORDER    the top_k of the rescored rows into d_out_rows / d_out_scores + the exact fallbacks of flagged queries.
 * Pass the same d_floor (or NULL) to every part of a batch (LIST reads it; the other two only act on whether there is one);
 * the outputs may be NULL unless ORDER is among the parts.  rr_dense_select_slot_dev = all three. */
#define RR_SELECT_LIST    1
#define RR_SELECT_RESCORE 2
#define RR_SELECT_ORDER   4
int rr_dense_select_part_dev(rr_index* ix, int32_t slot, int32_t parts, int32_t n_queries, int32_t top_k, const float* d_floor,
                             int64_t* d_out_rows, float* d_out_scores, void* stream);
/* A hipStream_t (as void*) whose kernels run only on CUs [first_cu, first_cu + n_cus) of `device`, counted in the driver's
 * CU-mask order (bit i sits on XCD i % 8: a contiguous range takes the same share of every XCD).  n_cus should be a multiple
 * of 32 (the dispatcher deals workgroups to the 32 shader engines in turn).  It is an ordinary BLOCKING stream: commands on
 * the NULL stream synchronise with it -- keep other work off the NULL stream while it is in use.  Destroy with
 * rr_stream_destroy before the process ends. */
int rr_stream_create_cu_range(int32_t device, int32_t first_cu, int32_t n_cus, void** out_stream);
int rr_stream_destroy(void* stream);
/* How many CUs the stream of this index's scans may use (0 or the device's CU count = all). */
int rr_index_set_scan_cus(rr_index* ix, int32_t n_cus);

/* Stream helpers for callers that chain *_dev calls. */
int rr_index_stream(rr_index* ix, void** out_stream);
int rr_index_synchronize(rr_index* ix);

#ifdef __cplusplus
}
#endif
#endif /* RR_HIP_H */
