// How do the vector unit and the matrix core treat SUBNORMAL fp16 values (|x| < 2^-14)?  csrc/rr_ce_h2.hip splits an fp32
// operand into hi = fp16(x), lo = fp16((x - hi) * 2048); this probe multiplies 16 x 32 by 32 x 16 matrices whose entries are
// drawn around the fp16 subnormal range, as three MFMA products on the split operands, with and without the split's guard
// (hi = 0 below 2^-14), and prints the worst error against the float64 product.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/f16_flush_probe.hip -o /tmp/f16_flush_probe && /tmp/f16_flush_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool GUARD>
__device__ void split(float x, _Float16& hi, _Float16& lo) {
    hi = (GUARD && __builtin_fabsf(x) < 6.103515625e-05f) ? (_Float16)0.f : (_Float16)x;
    lo = (_Float16)((x - (float)hi) * 2048.f);
}
// A [16][32], B^T [16][32] row-major fp32; D [16][16]; MFMA 16x16x32: lane (r = l & 15, kq = l >> 4) holds k = 8 kq .. 8 kq + 7
template <bool GUARD>
__global__ void probe(const float* A, const float* Bt, float* D) {
    const int l = threadIdx.x, r = l & 15, kq = l >> 4;
    h8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
        _Float16 h, o;
        split<GUARD>(A[r * 32 + 8 * kq + e], h, o); ah[e] = h; al[e] = o;
        split<GUARD>(Bt[r * 32 + 8 * kq + e], h, o); bh[e] = h; bl[e] = o;
    }
    f4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, a2, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, a2, 0, 0, 0);
    for (int e = 0; e < 4; ++e) D[(4 * kq + e) * 16 + r] = a1[e] + a2[e] * (1.f / 2048.f);      // D[row 4 kq + e][col r]
}
int main() {
    float A[512], B[512], D[2][256], *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof(A)); (void)hipMalloc(&dB, sizeof(B)); (void)hipMalloc(&dD, sizeof(D[0]));
    srand(7);
    for (int scale = 0; scale < 3; ++scale) {
        // scale 0: A ~ 1, B ~ 1 (nothing subnormal); 1: A log-uniform in [1e-7, 1e-3] (its hi often subnormal), B ~ 1; 2: both small
        for (int i = 0; i < 512; ++i) {
            const double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX, s1 = rand() & 1 ? 1 : -1, s2 = rand() & 2 ? 1 : -1;
            A[i] = (float)(s1 * (scale == 0 ? 0.5 + u : std::pow(10.0, -7 + 4 * u)));
            B[i] = (float)(s2 * (scale == 2 ? std::pow(10.0, -7 + 4 * v) : 0.5 + v));
        }
        (void)hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe<false>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(D[0], dD, sizeof(D[0]), hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(probe<true>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(D[1], dD, sizeof(D[1]), hipMemcpyDeviceToHost);
        double worst[2] = {0, 0}, mag = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0, ab = 0;
                for (int k = 0; k < 32; ++k) { ref += (double)A[i * 32 + k] * B[j * 32 + k]; ab += std::fabs((double)A[i * 32 + k] * B[j * 32 + k]); }
                mag = std::fmax(mag, ab);
                for (int g = 0; g < 2; ++g) worst[g] = std::fmax(worst[g], std::fabs(D[g][i * 16 + j] - ref) / ab);
            }
        printf("case %d: max |error| / sum |a b|: without the guard %.3e, with it %.3e   (sum |a b| up to %.3e)\n", scale, worst[0], worst[1], mag);
    }
    return 0;
}
