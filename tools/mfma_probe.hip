// mfma_probe -- issue rate of the exact-fp32 matrix instructions on gfx950, one wave per SIMD (256 threads per CU):
// cycles per instruction with 1, 2 and 4 independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int BIG>
__global__ void probe(float *out, int iters, unsigned long long *cyc) {
    f32x16 a16[4];
    f32x4 a4[4];
    for (int j = 0; j < 4; ++j) {
        for (int i = 0; i < 16; ++i) a16[j][i] = 0.0f;
        for (int i = 0; i < 4; ++i) a4[j][i] = 0.0f;
    }
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (BIG) a16[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a16[j], 0, 0, 0);
                else a4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[j], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int j = 0; j < 4; ++j) {
        for (int i = 0; i < 16; ++i) s += a16[j][i];
        for (int i = 0; i < 4; ++i) s += a4[j][i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int BIG>
static void run(const char *name, float *d_out, unsigned long long *d_cyc, double ghz) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<NACC, BIG>), dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, BIG>), dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;
    const double flop = (BIG ? 4096.0 : 2048.0) * n * 4 * 256;   // per instruction x instructions x waves per CU x CUs
    printf("%-34s %d accumulators: %6.1f counter ticks / instruction, %7.1f ns / instruction, %6.1f TFLOP/s chip\n", name, NACC,
           (double)cyc / n, ms * 1e6 / n, flop / (ms * 1e-3) / 1e12);
}

int main() {
    float *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_out, 256 * 256 * 4);
    hipMalloc(&d_cyc, 8);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%d CUs, %.2f GHz; one wave per SIMD\n", p.multiProcessorCount, ghz);
    run<1, 1>("v_mfma_f32_32x32x2_f32", d_out, d_cyc, ghz);
    run<2, 1>("v_mfma_f32_32x32x2_f32", d_out, d_cyc, ghz);
    run<4, 1>("v_mfma_f32_32x32x2_f32", d_out, d_cyc, ghz);
    run<1, 0>("v_mfma_f32_16x16x4_f32", d_out, d_cyc, ghz);
    run<2, 0>("v_mfma_f32_16x16x4_f32", d_out, d_cyc, ghz);
    run<4, 0>("v_mfma_f32_16x16x4_f32", d_out, d_cyc, ghz);
    return 0;
}
