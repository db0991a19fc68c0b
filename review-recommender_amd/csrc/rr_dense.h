// rr_dense.h -- pieces of K1 shared by the fp32 and bf16 scan files.
#pragma once
#include "rr_common.h"

#define RR_SCAN_THREADS 256
#define RR_MAX_SCAN_WAVES 8192   // 2048 workgroups x 4 waves: upper bound of any resident grid
#define RR_MFMA_MAXQ 64          // queries per exact matrix-core scan launch
#define RR_FLT_MAXQ 128          // queries per filter-scan launch (rr_dense_flt.hip)
#define RR_SEL_MAXQ 256          // queries one selection / rescoring launch may cover: two filter-scan launches (sets)
#define RR_FLT_SAMP_CAP 8192      // sampled tiles of the store prefilter (rr_flt_sample)
#define RR_FLT_NO_BOUND 1000     // rr_dense_chunk_flt: no finite row-norm bound, use the exact scans
#define RR_FLT_SMALL 1001        // rr_dense_chunk_flt: too few tiles for the filter, use the VALU scans

// Geometry of one scan launch, shared by the scan and the selection.
struct rr_scan_geom {
    int64_t n_rows, n_tiles;
    int64_t tiles_per_wave;   // C: wave w owns tiles [w*C, min((w+1)*C, n_tiles)) = one "group"
    int32_t n_waves;          // waves of the launch; every one owns at least one tile
    int32_t qs;               // 0: scores are [query][n_pad], tile maxima [query][n_tiles] (rr_scan_f32)
                              // Q: scores are [row / 16][Q][16], tile maxima [tile][Q], group maxima [wave][Q]
                              //    (rr_scan_mfma_f32 with Q = 16 * NQT query slots: every store of a wave is
                              //    one contiguous block of whole 128-B lines)
    int64_t n_pad;            // 64 * n_tiles
    int32_t mm_pairs;         // M-tile maxima of the two-pass path: 0 = [tile][Q][4] (rr_scan_mfma_x3),
                              // 1 = [32-row tile][Q][2] (rr_scan_x3w: whole lines per store),
                              // 3 = [32-row tile][Q] 4 bytes (rr_scan_flt): bf16 tile maximum rounded up + four 4-bit gaps of
                              //     its 8-ROW M-tiles; the listed ids and the rescoring scratch are then per 8 rows
    int32_t gpw;              // groups per wave (0 or 1: the wave's whole run is one group); rr_scan_flt cuts a run into
    int64_t tiles_per_group;  //   gpw sub-runs of tiles_per_group tiles, group g = wave * gpw + k: fewer tiles to open per group
};


// Splits the tiles into equal contiguous runs, one per wave of (at most) `resident_blocks` x 4 waves.
rr_scan_geom rr_make_geom(const rr_index* ix, int resident_blocks);
// Which kernel a scan launch ran (rr_index_last_scan_info): 1 rr_scan_f32, 2 rr_scan_bf16, 3 rr_scan_mfma_x3,
// 4 rr_scan_x3w, 5 rr_scan_flt, 6 rr_scan_mfma_f32, 7 rr_scan_mfma_bf16; terms = bf16 MFMA terms per dimension.
static inline void rr_scan_note(rr_index* ix, int kernel, int variant, int nq, int terms, int elem_bytes = 0) {
    ix->last_scan[0] = kernel; ix->last_scan[1] = variant; ix->last_scan[2] = nq; ix->last_scan[3] = terms;
    ix->last_scan[4] = elem_bytes ? elem_bytes : (ix->dtype == RR_DTYPE_BF16 ? 2 : 4);
}
// any write to the matrix: the cached row-norm bound and the bf16 filter plane no longer describe it
static inline void rr_matrix_written(rr_index* ix) {
    ix->norm_bound = -1.f;
    ix->shadow_valid = false;
}
// scan slots (rr_common.h: rr_scan_slot): which of the two sets of per-batch scan state the index fields describe
int rr_slot_activate(rr_index* ix, int slot);
void rr_slot_park(rr_index* ix);
// HIP-event pair around a scan launch (rr_index_scan_stats).
int rr_scan_events_begin(rr_index* ix, hipStream_t st);
void rr_scan_events_end(rr_index* ix, int slot, hipStream_t st);
// Exact top-pool of `nq` queries from the three score levels a scan left in the index scratch.
// `only_if` (device, one flag per query, may be null): queries whose flag is 0 are skipped.
int rr_dense_listed_fallback(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows, float* d_scores,
                             int32_t* flags, hipStream_t st);
void rr_launch_select(rr_index* ix, const rr_scan_geom& G, int nq, int pool, int64_t* d_rows,
                      float* d_scores, hipStream_t st, const int32_t* only_if = nullptr, int slices = 1,
                      int64_t sims_slice = 0, int64_t gmax_slice = 0, int64_t smax_slice = 0);
// Two-pass selection of the split-operand scan (see rr_select_mtiles in rr_dense.hip).
#define RR_X3_MCAP 16384         // M-tiles (8 or 16 rows) one query may ask to have rescored
struct rr_x3_scratch {           // (every array RR_FLT_MAXQ queries long)
    // (every array RR_SEL_MAXQ queries long: a selection launch may follow two scan launches)
    uint32_t* mtiles;            // [q][RR_X3_MCAP] M-tile ids to rescore
    int32_t* count;              // [q] how many
    uint32_t* tau;               // [q] key threshold of the query's rows
    int32_t* fb;                 // [q] 1 = the query needs the stored-score fallback
    float* eps;                  // [q] filter scan: error bound of the query's approximate scores
    float* sc;                   // [q][RR_X3_MCAP][16] rescored rows
};
rr_x3_scratch rr_x3_scratch_of(const rr_index* ix);
size_t rr_x3_scratch_bytes();
// `eps` (device, per query, may be null): the scan's scores are approximations within eps of the scores
// the rescoring will produce; M-tiles are then opened down to tau - 2 eps and rows kept down to tau - eps.
// `nq_b` > 0: the launch covers TWO scan launches (sets) of the same geometry: queries [0, nq) come from set 0, [nq,
// nq + nq_b) from set 1, whose tile / group maxima sit `set_stride_words` 4-byte words behind set 0's and whose eps /
// sigma entries start at index RR_FLT_MAXQ.
// `floor` (device, per query of the launch, may be null): row shards -- a lower bound of the corpus-wide pool-th best score.
void rr_launch_select_mtiles(rr_index* ix, const rr_scan_geom& G, int nq, int pool, hipStream_t st,
                             const float* eps = nullptr, const float* sigma = nullptr, int nq_b = 0,
                             int64_t mmax_set_stride = 0, int64_t smax_set_stride = 0, const float* floor = nullptr);
void rr_launch_select_rescored(rr_index* ix, const rr_scan_geom& G, int nq, int pool, int64_t* d_rows,
                               float* d_scores, hipStream_t st, bool floor_mode = false);
void rr_launch_group_kth(rr_index* ix, const rr_scan_geom& G, int nq_a, int nq_b, int kth, const float* eps, float* d_bound,
                         int64_t smax_set_stride, hipStream_t st);
// Waves a kernel can keep resident on the device (occupancy x CUs x waves per workgroup).
int rr_resident_waves(const void* kernel, int threads, int device);
// bf16-storage scans (rr_dense_bf16.hip): up to 8 (VALU) or 9..64 (matrix cores) queries.
int rr_dense_chunk_bf16(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                        float* d_scores, hipStream_t st);
int rr_dense_chunk_mfma_bf16(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                             float* d_scores, hipStream_t st);
// l2_normalize of rows [first, first + n) of an fp32 index, in place
int rr_l2norm_rows_f32(rr_index* ix, int64_t first_row, int64_t n, float eps, hipStream_t st);
// split-bf16 matrix-core scan (rr_dense_x3.hip): 5..64 queries, either storage dtype
int rr_dense_chunk_x3(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                      float* d_scores, hipStream_t st);
// bf16 filter scan + exact per-row-chain rescoring, 5..128 queries (rr_dense_flt.hip); RR_FLT_NO_BOUND when
// the matrix has no finite row-norm bound
// `phase`: 0 = scan + selection; 1 = scan only, bound[q] (rr_group_kth with `kth`) written, state kept in the index;
// 2 = selection of the scan phase 1 left behind, with `floor` (row shards)
// `parts` (phase 2 only): which of the selection's three steps to launch -- 1 the M-tile lists, 2 the rescoring, 4 the
// ordering + the flagged queries' fallbacks (the parked scan stays valid until the last one has been launched)
int rr_dense_chunk_flt(rr_index* ix, const float* d_q, int nq, int pool, int64_t* d_rows,
                       float* d_scores, hipStream_t st, int phase = 0, int kth = 0, float* d_bound = nullptr,
                       const float* d_floor = nullptr, int parts = 7);
// any other search on the index makes a parked phase-1 scan void
void rr_flt_drop_pending(rr_index* ix);
// fp32 rows (device, n x dim) -> the index's bf16 matrix rows [first, first + n), optional l2 normalise
int rr_store_rows_bf16(rr_index* ix, int64_t first_row, int64_t n, float* d_rows_f32, float eps, hipStream_t st);

// rr_api.hip: the address a kernel may use for p (device memory, or pinned host memory through its mapping)
int rr_device_visible(const void* p, const void** out, const char* what);

// rr_dense_flt.hip: rr_pad_queries and rr_flt_prep_queries in ONE launch (the front of a batched filter call): the padded
// fp32 queries into ix->d_q plus their bf16 planes and error bounds.  Returns RR_FLT_NO_BOUND (nothing launched) when the
// matrix has no finite row-norm bound: the caller pads the plain way.
int rr_flt_pad_prep(rr_index* ix, const float* d_queries, int nq, int slots, hipStream_t st);
