// permlane.hip -- dev check of the gfx950 lane-swap instructions the 32x32x16 scan relies on.
//   hipcc -O3 --offload-arch=gfx950 tools/permlane/permlane.hip -o tools/permlane/permlane
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
    const unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    p[threadIdx.x] = r[0]; p[64 + threadIdx.x] = r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    p[128 + threadIdx.x] = r2[0]; p[192 + threadIdx.x] = r2[1];
}
int main() {
    unsigned* d; unsigned h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[4] = {"permlane16_swap r[0]", "permlane16_swap r[1]", "permlane32_swap r[0]", "permlane32_swap r[1]"};
    for (int a = 0; a < 4; ++a) {
        printf("%s:", names[a]);
        for (int i = 0; i < 64; i += 16) printf("  lanes %2d.. = %3u..%3u", i, h[a * 64 + i], h[a * 64 + i + 15]);
        printf("\n");
    }
    return 0;
}
