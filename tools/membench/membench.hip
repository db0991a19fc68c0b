// membench.hip -- dev microbenchmark: what read bandwidth do the scan kernels' ACCESS SHAPES reach
// on their own (no MFMA, no LDS traffic), at the scan kernels' occupancy?
//   hipcc -O3 --offload-arch=gfx950 tools/membench/membench.hip -o tools/membench/membench
//   tools/membench/membench [rows]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: four rows x 256 contiguous bytes per instruction (rr_scan_f32), 12 loads in flight
// MODE 1: MFMA fragment shape: lane -> row (lane & 15), 16-B piece (lane >> 4): 16 rows x 64 B
// MODE 2: four adjacent lanes = 64 contiguous bytes of one row: lane -> row (lane >> 2), piece (lane & 3)
// MODE 3: tile-major: every instruction reads 1 KiB contiguous
template <int MODE, int RING>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ mat, long n_mtiles, long tiles_per_wave, float* out) {
    extern __shared__ float pad[];          // occupancy control only
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long m0 = wave * tiles_per_wave, m1 = m0 + tiles_per_wave < n_mtiles ? m0 + tiles_per_wave : n_mtiles;
    if (m0 >= n_mtiles) return;
    auto ptr = [&](long mt) -> const f32x4* {
        mt = mt < m1 ? mt : m1 - 1;
        const f32x4* base = mat + mt * (16 * 96);                 // 16 rows x 96 float4 = 24 KB per M-tile
        if (MODE == 0) return base + (lane >> 4) * 96 + (lane & 15);      // + 4 rows*96 per step handled below
        if (MODE == 1) return base + (lane & 15) * 96 + (lane >> 4);
        if (MODE == 2) return base + (lane >> 2) * 96 + (lane & 3);
        return base + lane;
    };
    auto off = [&](int j) -> long {                                 // float4 offset of ring unit j
        if (MODE == 0) return (long)(j / 6) * 4 * 96 + 16 * (j % 6);      // 4 row-quads x 6 column blocks
        if (MODE == 3) return 64 * j;
        return 4 * j;                                               // next 64-B group of the same rows
    };
    f32x4 a[RING];
    float acc = 0.f;
    const f32x4* p = ptr(m0);
#pragma unroll
    for (int j = 0; j < RING; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[j]) : "v"(p + off(j)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (long mt = m0; mt < m1; ++mt) {
        const f32x4* pn = ptr(mt + 1);
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a[j]) : "n"(RING - 1) : "memory");
            acc += a[j].x + a[j].y + a[j].z + a[j].w;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[j]) : "v"(pn + off(j)), "v"(acc) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) out[wave] = acc;
}

template <int MODE>
static void run(const char* name, const f32x4* d, long n_mtiles, float* out, int blocks_per_cu) {
    const int lds = blocks_per_cu >= 4 ? 32 * 1024 : blocks_per_cu == 3 ? 48 * 1024 : blocks_per_cu == 2 ? 72 * 1024 : 150 * 1024;
    hipFuncSetAttribute((const void*)k<MODE, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long waves = 256L * blocks_per_cu * 4;
    const long tpw = (n_mtiles + waves - 1) / waves;
    const long used = (n_mtiles + tpw - 1) / tpw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, 24>), dim3((used + 3) / 4), dim3(256), lds, 0, d, n_mtiles, tpw, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%-44s %d waves/SIMD: %.3f ms  %.0f GB/s\n", name, blocks_per_cu, best, n_mtiles * 24576.0 / best / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 10000000;
    const long n_mtiles = rows / 16;
    f32x4* d; float* out;
    hipMalloc(&d, n_mtiles * 24576L); hipMalloc(&out, 1 << 20);
    hipMemset(d, 0, n_mtiles * 24576L);
    for (int occ = 4; occ >= 1; --occ) {
        if (occ == 3) continue;
        run<0>("4 rows x 256 B per instruction (VALU scan)", d, n_mtiles, out, occ);
        run<1>("16 rows x 64 B, MFMA lane order", d, n_mtiles, out, occ);
        run<2>("16 rows x 64 B, adjacent lanes contiguous", d, n_mtiles, out, occ);
        run<3>("1 KiB contiguous (tile-major layout)", d, n_mtiles, out, occ);
    }
    return 0;
}
