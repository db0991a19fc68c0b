// rr_api.hip -- handles, errors and data movement behind the C ABI (include/rr_hip.h).
#include <stdarg.h>

#include "rr_common.h"
#include "rr_dense.h"

static thread_local char g_err[512] = "";

void rr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rr_last_error(void) { return g_err; }
extern "C" int rr_version(void) { return 100; }

extern "C" int rr_device_count(int* out) {
    RR_REQUIRE(out, "rr_device_count: NULL out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out = 0;
        rr_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return RR_E_HIP;
    }
    *out = n;
    return RR_OK;
}

static size_t rr_elem_size(int dtype) { return dtype == RR_DTYPE_BF16 ? 2 : 4; }

// ------------------------------------------------------------------ kernel-driven copies (include/rr_hip.h)
// One launch moves up to RR_COPY_MAX_SEGS pitched segments; either side of a segment may be pinned host memory mapped
// into the device's address space (hipHostMalloc / torch pin_memory: the host pointer is the device pointer).  A
// batch's inputs and answers are a few hundred KB: as copy commands each of them costs 10 - 20 us of fixed latency on
// the stream; as loads / stores of one kernel they cost their PCIe time (~15 us per 0.7 MB).
struct rr_copy_args {
    rr_copy_seg seg[RR_COPY_MAX_SEGS];
};
__global__ __launch_bounds__(256) void rr_copy_segments(rr_copy_args A) {
    const rr_copy_seg sg = A.seg[blockIdx.y];
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
    char* dst = static_cast<char*>(sg.dst);
    const char* src = static_cast<const char*>(sg.src);
    const uint64_t bits = (uint64_t)(uintptr_t)dst | (uint64_t)(uintptr_t)src | (uint64_t)sg.row_bytes | (uint64_t)sg.dst_pitch |
                          (uint64_t)sg.src_pitch;
    if ((bits & 15u) == 0) {
        const int64_t per_row = sg.row_bytes / 16, n = per_row * sg.rows;
        for (int64_t i = tid; i < n; i += nthreads) {
            const int64_t r = i / per_row, c = i % per_row;
            *reinterpret_cast<f32x4*>(dst + r * sg.dst_pitch + 16 * c) = *reinterpret_cast<const f32x4*>(src + r * sg.src_pitch + 16 * c);
        }
    } else if ((bits & 3u) == 0) {
        const int64_t per_row = sg.row_bytes / 4, n = per_row * sg.rows;
        for (int64_t i = tid; i < n; i += nthreads) {
            const int64_t r = i / per_row, c = i % per_row;
            *reinterpret_cast<uint32_t*>(dst + r * sg.dst_pitch + 4 * c) = *reinterpret_cast<const uint32_t*>(src + r * sg.src_pitch + 4 * c);
        }
    } else {
        const int64_t n = sg.row_bytes * sg.rows;
        for (int64_t i = tid; i < n; i += nthreads) {
            const int64_t r = i / sg.row_bytes, c = i % sg.row_bytes;
            dst[r * sg.dst_pitch + c] = src[r * sg.src_pitch + c];
        }
    }
}

// The address a kernel may use for `p`: device memory as it is; pinned host memory through its mapping (on this stack the
// same value, but the runtime is the one to say so); anything else (pageable host memory) is refused.
int rr_device_visible(const void* p, const void** out, const char* what) {
    hipPointerAttribute_t at;
    const hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rr_set_error("%s: %p is neither device memory nor pinned (mapped) host memory", what, p);
        return RR_E_INVALID;
    }
    if (at.type == hipMemoryTypeHost) {
        RR_REQUIRE(at.devicePointer != nullptr, "%s: pinned host memory %p is not mapped into the device", what, p);
        *out = static_cast<const char*>(at.devicePointer);
        return RR_OK;
    }
    RR_REQUIRE(at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeUnified,
               "%s: %p is pageable host memory (pin it, or copy it to the device first)", what, p);
    *out = p;
    return RR_OK;
}

extern "C" int rr_index_matrix_changed(rr_index* ix) {
    RR_REQUIRE(ix != nullptr, "rr_index_matrix_changed: NULL index");
    std::lock_guard<std::mutex> lk(ix->mu);
    rr_matrix_written(ix);      // drops the cached row-norm bounds and the bf16 filter plane; both are rebuilt lazily
    return RR_OK;
}

extern "C" int rr_copy_segments_dev(const rr_copy_seg* segs, int32_t n_segs, int32_t device, void* stream) {
    RR_REQUIRE(segs && n_segs >= 1 && n_segs <= RR_COPY_MAX_SEGS, "rr_copy_segments_dev: 1 .. %d segments", RR_COPY_MAX_SEGS);
    rr_copy_args A;
    int64_t most = 0;
    for (int i = 0; i < n_segs; ++i) {
        const rr_copy_seg& g = segs[i];
        RR_REQUIRE(g.dst && g.src && g.rows >= 0 && g.row_bytes >= 0 && g.dst_pitch >= g.row_bytes && g.src_pitch >= g.row_bytes,
                   "rr_copy_segments_dev: segment %d: NULL pointer, negative size or a pitch below row_bytes", i);
        A.seg[i] = g;
        const void* vis = nullptr;
        int rc = rr_device_visible(g.dst, &vis, "rr_copy_segments_dev (dst)");
        if (rc) return rc;
        A.seg[i].dst = const_cast<void*>(vis);
        rc = rr_device_visible(g.src, &vis, "rr_copy_segments_dev (src)");
        if (rc) return rc;
        A.seg[i].src = vis;
        const int64_t bytes = g.rows * g.row_bytes;
        most = bytes > most ? bytes : most;
    }
    if (most == 0) return RR_OK;
    RR_HIP_TRY(hipSetDevice(device));
    int64_t blocks = (most / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 512 ? 512 : blocks;
    hipLaunchKernelGGL(rr_copy_segments, dim3((unsigned)blocks, (unsigned)n_segs), dim3(256), 0, (hipStream_t)stream, A);
    RR_HIP_TRY(hipGetLastError());
    return RR_OK;
}

extern "C" int rr_index_create(const void* h_matrix, int64_t n_rows, int32_t dim, int32_t dtype,
                               int32_t device, int64_t row_offset, rr_index** out) {
    RR_REQUIRE(out, "rr_index_create: NULL out");
    *out = nullptr;
    RR_REQUIRE(n_rows >= 1 && n_rows < (1ll << 31), "rr_index_create: n_rows %lld out of [1, 2^31)",
               (long long)n_rows);
    RR_REQUIRE(dim >= 1 && dim <= 8192, "rr_index_create: dim %d out of [1, 8192]", dim);
    RR_REQUIRE(dtype == RR_DTYPE_F32 || dtype == RR_DTYPE_BF16, "rr_index_create: unknown dtype %d", dtype);
    RR_REQUIRE(dtype == RR_DTYPE_F32 || dim == 384, "rr_index_create: bf16 storage is built for dim 384 only");
    RR_REQUIRE(row_offset >= 0 && row_offset + n_rows < (1ll << 32),
               "rr_index_create: global rows must stay below 2^32");
    int ndev = 0;
    RR_HIP_TRY(hipGetDeviceCount(&ndev));
    RR_REQUIRE(device >= 0 && device < ndev, "rr_index_create: device %d not in [0,%d)", device, ndev);
    RR_HIP_TRY(hipSetDevice(device));

    rr_index* ix = new rr_index();
    ix->device = device;
    ix->n_rows = n_rows;
    ix->dim = dim;
    ix->dim_pad = (int32_t)rr_round_up(dim, 64);
    ix->dtype = dtype;
    ix->row_offset = row_offset;
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ix->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ix->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->ev_done, hipEventDisableTiming);
    for (int i = 0; i < RR_SCAN_SLOTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ix->slot_ev[i], hipEventDisableTiming);
    if (e == hipSuccess && (hipDeviceGetAttribute(&ix->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ix->n_cus < 1))
        ix->n_cus = 256;
    for (int i = 0; i < rr_index::kRing && e == hipSuccess; ++i) {
        e = hipEventCreate(&ix->ring0[i]);
        if (e == hipSuccess) e = hipEventCreate(&ix->ring1[i]);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_sel_trace, sizeof(int32_t) * 16 * RR_SEL_MAXQ);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_flag_list, sizeof(int32_t) * 16);
    if (e == hipSuccess) e = hipMalloc(&ix->d_qplanes, (size_t)2 * 3 * 64 * 384 * 2)   /* two sets of query planes (paired filter-scan launches) */;
    if (e == hipSuccess) e = hipMalloc(&ix->d_x3, rr_x3_scratch_bytes());
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_eps, sizeof(float) * RR_SEL_MAXQ);
    if (e == hipSuccess) e = hipMalloc(&ix->d_q, sizeof(float) * (size_t)RR_MAX_BATCH * ix->dim_pad);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_rows_out, sizeof(int64_t) * (size_t)RR_MAX_BATCH * RR_MAX_POOL);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_scores_out, sizeof(float) * (size_t)RR_MAX_BATCH * RR_MAX_POOL);
    if (e != hipSuccess) {
        rr_set_error("rr_index_create: %s", hipGetErrorString(e));
        rr_index_destroy(ix);
        return RR_E_HIP;
    }
    if (h_matrix) {
        int rc = rr_index_upload_rows(ix, 0, n_rows, h_matrix);
        if (rc) { rr_index_destroy(ix); return rc; }
    }
    *out = ix;
    return RR_OK;
}

static int rr_alloc_matrix(rr_index* ix) {
    if (ix->d_matrix) return RR_OK;
    const size_t bytes = (size_t)ix->n_rows * ix->dim_pad * rr_elem_size(ix->dtype);
    hipError_t e = hipMalloc(&ix->d_matrix, bytes);
    if (e != hipSuccess) {
        rr_set_error("rr_index: hipMalloc of %zu matrix bytes failed: %s", bytes, hipGetErrorString(e));
        return RR_E_NOMEM;
    }
    ix->owns_matrix = true;
    if (ix->dim_pad != ix->dim) RR_HIP_TRY(hipMemsetAsync(ix->d_matrix, 0, bytes, ix->stream));
    return RR_OK;
}

extern "C" int rr_index_upload_rows(rr_index* ix, int64_t first_row, int64_t n_rows, const void* h_rows) {
    RR_REQUIRE(ix && h_rows, "rr_index_upload_rows: NULL argument");
    RR_REQUIRE(first_row >= 0 && n_rows >= 0 && first_row + n_rows <= ix->n_rows,
               "rr_index_upload_rows: rows [%lld,%lld) outside [0,%lld)", (long long)first_row,
               (long long)(first_row + n_rows), (long long)ix->n_rows);
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    RR_REQUIRE(ix->owns_matrix || !ix->d_matrix, "rr_index_upload_rows: matrix is caller-owned");
    int rc = rr_alloc_matrix(ix);
    if (rc) return rc;
    const size_t es = rr_elem_size(ix->dtype);
    rr_matrix_written(ix);
    char* dst = (char*)ix->d_matrix + (size_t)first_row * ix->dim_pad * es;
    if (n_rows)
        RR_HIP_TRY(hipMemcpy2DAsync(dst, es * ix->dim_pad, h_rows, es * ix->dim, es * ix->dim, (size_t)n_rows,
                                    hipMemcpyHostToDevice, ix->stream));
    RR_HIP_TRY(hipStreamSynchronize(ix->stream));
    return RR_OK;
}

extern "C" int rr_index_upload_rows_f32(rr_index* ix, int64_t first_row, int64_t n_rows, const float* h_rows,
                                        float normalize_eps) {
    RR_REQUIRE(ix && h_rows, "rr_index_upload_rows_f32: NULL argument");
    RR_REQUIRE(first_row >= 0 && n_rows >= 0 && first_row + n_rows <= ix->n_rows,
               "rr_index_upload_rows_f32: rows [%lld,%lld) outside [0,%lld)", (long long)first_row,
               (long long)(first_row + n_rows), (long long)ix->n_rows);
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    RR_REQUIRE(ix->owns_matrix || !ix->d_matrix, "rr_index_upload_rows_f32: matrix is caller-owned");
    int rc = rr_alloc_matrix(ix);
    if (rc || n_rows == 0) return rc;
    rr_matrix_written(ix);   // any write to the matrix (either storage dtype) invalidates the cached bound and filter plane
    if (ix->dtype == RR_DTYPE_F32) {
        char* dst = (char*)ix->d_matrix + (size_t)first_row * ix->dim_pad * 4;
        RR_HIP_TRY(hipMemcpy2DAsync(dst, 4 * (size_t)ix->dim_pad, h_rows, 4 * (size_t)ix->dim, 4 * (size_t)ix->dim,
                                    (size_t)n_rows, hipMemcpyHostToDevice, ix->stream));
        if (normalize_eps > 0.f) rc = rr_l2norm_rows_f32(ix, first_row, n_rows, normalize_eps, ix->stream);
    } else {
        float* tmp = nullptr;
        RR_HIP_TRY(hipMalloc((void**)&tmp, sizeof(float) * (size_t)n_rows * ix->dim));
        hipError_t e = hipMemcpyAsync(tmp, h_rows, sizeof(float) * (size_t)n_rows * ix->dim, hipMemcpyHostToDevice,
                                      ix->stream);
        if (e == hipSuccess) rc = rr_store_rows_bf16(ix, first_row, n_rows, tmp, normalize_eps, ix->stream);
        hipStreamSynchronize(ix->stream);
        hipFree(tmp);
        if (e != hipSuccess) { rr_set_error("rr_index_upload_rows_f32: %s", hipGetErrorString(e)); return RR_E_HIP; }
    }
    RR_HIP_TRY(hipStreamSynchronize(ix->stream));
    return rc;
}

extern "C" int rr_index_adopt_device(rr_index* ix, const void* d_matrix) {
    RR_REQUIRE(ix && d_matrix, "rr_index_adopt_device: NULL argument");
    RR_REQUIRE(((uintptr_t)d_matrix & 15) == 0, "rr_index_adopt_device: matrix must be 16-byte aligned");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    if (ix->d_matrix && ix->owns_matrix) hipFree(ix->d_matrix);
    ix->d_matrix = const_cast<void*>(d_matrix);
    rr_matrix_written(ix);
    ix->owns_matrix = false;
    return RR_OK;
}

extern "C" int rr_index_dim_padded(const rr_index* ix, int32_t* out) {
    RR_REQUIRE(ix && out, "rr_index_dim_padded: NULL argument");
    *out = ix->dim_pad;
    return RR_OK;
}

extern "C" int rr_index_set_meta(rr_index* ix, const double* h_n_reviews, const double* h_avg_stars,
                                 const double* h_log1p_n) {
    RR_REQUIRE(ix && h_n_reviews && h_avg_stars && h_log1p_n, "rr_index_set_meta: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    RR_HIP_TRY(hipSetDevice(ix->device));
    const size_t bytes = sizeof(double) * (size_t)ix->n_rows;
    if (!ix->d_n_reviews) RR_HIP_TRY(hipMalloc((void**)&ix->d_n_reviews, bytes));
    if (!ix->d_avg_stars) RR_HIP_TRY(hipMalloc((void**)&ix->d_avg_stars, bytes));
    if (!ix->d_log1p_n) RR_HIP_TRY(hipMalloc((void**)&ix->d_log1p_n, bytes));
    RR_HIP_TRY(hipMemcpy(ix->d_n_reviews, h_n_reviews, bytes, hipMemcpyHostToDevice));
    RR_HIP_TRY(hipMemcpy(ix->d_avg_stars, h_avg_stars, bytes, hipMemcpyHostToDevice));
    RR_HIP_TRY(hipMemcpy(ix->d_log1p_n, h_log1p_n, bytes, hipMemcpyHostToDevice));
    ix->has_meta = true;
    return RR_OK;
}

extern "C" int rr_index_destroy(rr_index* ix) {
    if (!ix) return RR_OK;
    hipSetDevice(ix->device);
    if (ix->stream) hipStreamSynchronize(ix->stream);
    if (ix->d_matrix && ix->owns_matrix) hipFree(ix->d_matrix);
    hipFree(ix->d_n_reviews); hipFree(ix->d_avg_stars); hipFree(ix->d_log1p_n);
    rr_slot_park(ix);          // both slots' buffers now sit in ix->parked[]
    for (rr_scan_slot& s : ix->parked) {
        hipFree(s.d_q); hipFree(s.d_qplanes); hipFree(s.d_eps); hipFree(s.d_gmax); hipFree(s.d_smax);
        hipFree(s.d_flt_samp); hipFree(s.d_flt_sigma); hipFree(s.d_flt_prog); free(s.flt_pending);
    }
    hipFree(ix->d_sims); hipFree(ix->d_sel_trace); hipFree(ix->d_flag_list); hipFree(ix->d_x3);
    hipFree(ix->d_rows_out); hipFree(ix->d_scores_out); hipFree(ix->d_shadow);
    for (int i = 0; i < RR_SCAN_SLOTS; ++i) if (ix->slot_ev[i]) hipEventDestroy(ix->slot_ev[i]);
    if (ix->ev0) hipEventDestroy(ix->ev0);
    if (ix->ev1) hipEventDestroy(ix->ev1);
    if (ix->ev_done) hipEventDestroy(ix->ev_done);
    for (int i = 0; i < rr_index::kRing; ++i) {
        if (ix->ring0[i]) hipEventDestroy(ix->ring0[i]);
        if (ix->ring1[i]) hipEventDestroy(ix->ring1[i]);
    }
    if (ix->stream) hipStreamDestroy(ix->stream);
    delete ix;
    return RR_OK;
}

// ---------------------------------------------------------------- CU-masked streams (pipelined K1)
// A stream whose kernels may only run on CUs [first_cu, first_cu + n_cus) of the device, in the driver's mask order (bit i =
// XCD i % 8, so a contiguous range takes the same share of every XCD and of its L2).  What lets batch i's selection /
// K2 / K3 run on a few CUs of their own while batch i + 1's scan -- one 512-register wave per SIMD: nothing else fits on a
// CU it occupies -- holds the rest.
extern "C" int rr_stream_create_cu_range(int32_t device, int32_t first_cu, int32_t n_cus, void** out_stream) {
    RR_REQUIRE(out_stream, "rr_stream_create_cu_range: NULL out");
    *out_stream = nullptr;
    int ndev = 0, cus = 0;
    RR_HIP_TRY(hipGetDeviceCount(&ndev));
    RR_REQUIRE(device >= 0 && device < ndev, "rr_stream_create_cu_range: device %d not in [0,%d)", device, ndev);
    RR_HIP_TRY(hipSetDevice(device));
    RR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    RR_REQUIRE(first_cu >= 0 && n_cus >= 1 && first_cu + n_cus <= cus, "rr_stream_create_cu_range: CUs [%d, %d) outside [0, %d)",
               first_cu, first_cu + n_cus, cus);
    uint32_t mask[32] = {0};
    RR_REQUIRE(cus <= 32 * 32, "rr_stream_create_cu_range: %d CUs", cus);
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c / 32] |= 1u << (c % 32);
    hipStream_t st = nullptr;
    RR_HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask));
    *out_stream = (void*)st;
    return RR_OK;
}

extern "C" int rr_stream_destroy(void* stream) {
    if (!stream) return RR_OK;
    RR_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    RR_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return RR_OK;
}

extern "C" int rr_index_set_scan_cus(rr_index* ix, int32_t n_cus) {
    RR_REQUIRE(ix, "rr_index_set_scan_cus: NULL handle");
    RR_REQUIRE(n_cus >= 0 && n_cus <= ix->n_cus, "rr_index_set_scan_cus: %d outside [0, %d]", n_cus, ix->n_cus);
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->scan_cus = n_cus == ix->n_cus ? 0 : n_cus;
    return RR_OK;
}

extern "C" int rr_index_stream(rr_index* ix, void** out_stream) {
    RR_REQUIRE(ix && out_stream, "rr_index_stream: NULL argument");
    *out_stream = (void*)ix->stream;
    return RR_OK;
}

extern "C" int rr_index_synchronize(rr_index* ix) {
    RR_REQUIRE(ix, "rr_index_synchronize: NULL handle");
    RR_HIP_TRY(hipSetDevice(ix->device));
    RR_HIP_TRY(hipStreamSynchronize(ix->stream));
    return RR_OK;
}
