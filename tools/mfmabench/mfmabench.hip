// mfmabench.hip -- dev microbenchmark: MFMA pipe utilisation of dependent accumulator chains, as the
// batched scan issues them: NACC accumulators per wave used round-robin, 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/mfmabench/mfmabench.hip -o tools/mfmabench/mfmabench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool WIDE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)(threadIdx.x & 7), b[i] = (__bf16)1.0f;
    f32x16 w[NACC];
    f32x4 n[NACC];
    for (int t = 0; t < NACC; ++t) {
        for (int e = 0; e < 16; ++e) w[t][e] = 0.f;
        n[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if (WIDE) w[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, w[t], 0, 0, 0);
                else n[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, n[t], 0, 0, 0);
            }
    }
    const long long c1 = clock64();
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) s += WIDE ? w[t][3] : n[t].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

// the scan's operand pattern: per K-step three A planes x three B planes per accumulator in the
// six-term order, two accumulators alternating, every operand in its own registers
__global__ __launch_bounds__(256) void k_terms(float* out, int iters, long long* cyc, const bf16x8* src) {
    bf16x8 a[3], q[2][3];
    for (int i = 0; i < 3; ++i) a[i] = src[threadIdx.x + 256 * i], q[0][i] = src[threadIdx.x + 256 * (3 + i)], q[1][i] = src[threadIdx.x + 256 * (6 + i)];
    f32x16 w[2];
    for (int t = 0; t < 2; ++t) for (int e = 0; e < 16; ++e) w[t][e] = 0.f;
    const long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 am = (i == 0) ? a[2] : (i == 2 || i == 3) ? a[1] : a[0];
                    const bf16x8 bm = (i == 1) ? q[t][2] : (i == 2 || i == 4) ? q[t][1] : q[t][0];
                    w[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, w[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    const long long c1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = w[0][3] + w[1][5];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

template <int NACC, bool WIDE>
static void run(float* out, long long* cyc, int waves_per_simd) {
    const int iters = 2000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, WIDE>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma = (double)iters * 8 * NACC;            // per wave
    const double busy = n_mfma * waves_per_simd * (WIDE ? 32 : 16);   // pipe cycles per SIMD
    printf("%s acc=%d waves/SIMD=%d: %.3f ms, %lld wave cycles (%.1f cyc/MFMA/wave), clock %.2f GHz, pipe busy %.0f %%\n",
           WIDE ? "32x32x16" : "16x16x32", NACC, waves_per_simd, best, c, c / n_mfma, c / (best * 1e6), 100.0 * busy / c);
    fflush(stdout);
}

int main() {
    float* out; long long* cyc;
    if (hipMalloc(&out, 4 << 20) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    {
        bf16x8* src; (void)hipMalloc(&src, 256 * 9 * 16); (void)hipMemset(src, 0, 256 * 9 * 16);
        const int iters = 500;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_terms, dim3(256), dim3(256), 0, 0, out, iters, cyc, src);
        (void)hipDeviceSynchronize();
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("scan operand pattern (3 A x 3 B planes, 2 accumulators): %.1f cyc/MFMA\n", c / (iters * 48.0));
    }
    for (int w = 1; w <= 1; ++w) {
        run<1, true>(out, cyc, w); run<2, true>(out, cyc, w); run<4, true>(out, cyc, w);
        run<1, false>(out, cyc, w); run<2, false>(out, cyc, w); run<4, false>(out, cyc, w);
    }
    return 0;
}
