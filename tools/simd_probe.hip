// simd_probe.hip -- which SIMD of its CU does wave w of a workgroup land on?  (development probe: the lane-per-item GGS kernel sorts its
// work items by length so that long and short waves share a SIMD; that needs the wave -> SIMD rule.)  HW_REG_HW_ID bits [5:4] = SIMD_ID.
//   hipcc --offload-arch=gfx950 -O3 tools/simd_probe.hip -o tools/simd_probe && tools/simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned *out, int lds_dummy) {
    extern __shared__ float sm[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
    if (lds_dummy < 0) sm[threadIdx.x] = 0;
}
int main() {
    unsigned *d, h[64 * 16];
    hipMalloc(&d, sizeof(h));
    for (int threads : {384, 512, 768}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0xff, sizeof(h));
            hipLaunchKernelGGL(probe, dim3(64), dim3(threads), 150 * 1024, 0, d, 0);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            int hist[16][4] = {};
            for (int b = 0; b < 64; ++b)
                for (int w = 0; w < threads / 64; ++w) hist[w][(h[b * 16 + w] >> 4) & 3]++;
            printf("%d threads (64 workgroups, one per CU by LDS): wave -> {SIMD0, SIMD1, SIMD2, SIMD3} counts\n", threads);
            for (int w = 0; w < threads / 64; ++w) printf("  wave %2d: %2d %2d %2d %2d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        }
    }
    return 0;
}
