// How one wave's vector instructions overlap its own MFMAs (one wave per SIMD, gfx950): cycles per MFMA slot for
// [1 MFMA + NV vector FMAs] repeated, MFMAs as ONE dependent accumulator chain or over FOUR accumulators in turn.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int NACC, int SHAPE, int VKIND>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f - i * 0.01f); }
    f32x16 acc32[4];
    f32x4 acc16[4];
    for (int j = 0; j < 4; ++j) {
        for (int e = 0; e < 16; ++e) acc32[j][e] = 0.f;
        for (int e = 0; e < 4; ++e) acc16[j][e] = 0.f;
    }
    f32x2 x[12];
    for (int j = 0; j < 12; ++j) x[j] = (f32x2){threadIdx.x * 0.5f + j, 1.f + j};
    const f32x2 ka = {0.999f, 1.001f}, kb = {0.001f, -0.001f};
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (SHAPE == 32) acc32[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[s % NACC], 0, 0, 0);
            else acc16[s % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[s % NACC], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (VKIND == 0) x[j][0] = __builtin_fmaf(x[j][0], ka[0], kb[0]);                 // v_fma_f32, independent chains
                else if (VKIND == 1) x[j] = __builtin_elementwise_fma(x[j], ka, kb);             // v_pk_fma_f32, independent chains
                else x[0][0] = __builtin_fmaf(x[0][0], ka[0], kb[0]);                            // v_fma_f32, ONE dependent chain
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = clock64();
    float r = 0.f;
    for (int j = 0; j < 4; ++j) { for (int e = 0; e < 16; ++e) r += acc32[j][e]; for (int e = 0; e < 4; ++e) r += acc16[j][e]; }
    for (int j = 0; j < 12; ++j) r += x[j][0] + x[j][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NACC, int SHAPE, int VKIND>
static void run(float* out, unsigned long long* cyc, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<NV, NACC, SHAPE, VKIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((probe<NV, NACC, SHAPE, VKIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("  NV=%2d: %6.1f", NV, (double)h / (iters * 8.0));
}

template <int NACC, int SHAPE, int VKIND>
static void row(float* out, unsigned long long* cyc, int blocks, const char* what) {
    printf("%-62s", what);
    run<0, NACC, SHAPE, VKIND>(out, cyc, blocks);
    run<2, NACC, SHAPE, VKIND>(out, cyc, blocks);
    run<4, NACC, SHAPE, VKIND>(out, cyc, blocks);
    run<6, NACC, SHAPE, VKIND>(out, cyc, blocks);
    run<8, NACC, SHAPE, VKIND>(out, cyc, blocks);
    run<10, NACC, SHAPE, VKIND>(out, cyc, blocks);
    run<12, NACC, SHAPE, VKIND>(out, cyc, blocks);
    printf("\n");
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int blocks : {1, 256}) {
        printf("---- %d workgroup(s) of 4 waves; shader cycles per [1 MFMA + NV vector instructions]\n", blocks);
        row<1, 32, 0>(out, cyc, blocks, "32x32x16, one accumulator chain, v_fma_f32 independent");
        row<4, 32, 0>(out, cyc, blocks, "32x32x16, four accumulators,     v_fma_f32 independent");
        row<1, 32, 1>(out, cyc, blocks, "32x32x16, one accumulator chain, v_pk_fma_f32 independent");
        row<4, 32, 1>(out, cyc, blocks, "32x32x16, four accumulators,     v_pk_fma_f32 independent");
        row<4, 32, 2>(out, cyc, blocks, "32x32x16, four accumulators,     v_fma_f32 one dependent chain");
        row<1, 16, 0>(out, cyc, blocks, "16x16x32, one accumulator chain, v_fma_f32 independent");
        row<4, 16, 0>(out, cyc, blocks, "16x16x32, four accumulators,     v_fma_f32 independent");
        row<4, 16, 1>(out, cyc, blocks, "16x16x32, four accumulators,     v_pk_fma_f32 independent");
    }
    return 0;
}
