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

// round 6: the fp16 matrix instruction of the fp16-plane GEMMs, v_mfma_f32_32x32x16_f16, one wave per SIMD, NACC independent accumulators issued round-robin:
// how many independent chains does a wave need to keep the pipe at its 32-cycle issue rate?
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void probe_h(float *out, int iters, unsigned long long *cyc) {
    f32x16 a16[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) a16[j][i] = 0.0f;
    f16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 1e-3f); y[e] = (_Float16)(1.0f + threadIdx.x * 1e-4f); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) a16[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a16[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += a16[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
static void run_h(float *d_out, unsigned long long *d_cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe_h<NACC>), dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_h<NACC>), dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;
    printf("v_mfma_f32_32x32x16_f16            %d accumulators: %6.1f counter ticks / instruction, %7.1f ns / instruction, %6.1f TFLOP/s chip\n", NACC, (double)cyc / n, ms * 1e6 / n,
           32768.0 * n * 4 * 256 / (ms * 1e-3) / 1e12);
}

// round 6: what the strip GEMM's K loop does around an MFMA -- four v_perm_b32 build the next A fragment, then the MFMA reads it.  MODE 0: every fragment goes to the SAME
// four registers (what the compiler allocates: a write-after-read on the operand of the MFMA just issued); MODE 1: two register sets alternate; MODE 2: no un-zips at all
template <int MODE>
__global__ void probe_perm(float *out, int iters, unsigned long long *cyc, const unsigned *src) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    u4 raw[6];
    for (int j = 0; j < 6; ++j) raw[j] = *(const u4 *)(src + (threadIdx.x * 6 + j) * 4);
    u4 wv = *(const u4 *)(src + threadIdx.x * 4);
    u4 fa, fb;
    fa = raw[0]; fb = raw[1];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const u4 p = raw[(u * 2) % 6], q = raw[(u * 2 + 1) % 6];
            u4 &f = (MODE == 1 && (u & 1)) ? fb : fa;
            if (MODE != 2) {
                asm volatile("v_perm_b32 %0, %4, %5, %8\n\tv_perm_b32 %1, %6, %7, %8\n\tv_perm_b32 %2, %4, %6, %8\n\tv_perm_b32 %3, %5, %7, %8\n\ts_nop 1"
                             : "=&v"(f.x), "=&v"(f.y), "=&v"(f.z), "=&v"(f.w) : "v"(p.x), "v"(p.y), "v"(q.x), "v"(q.y), "s"(0x05040100u));
            }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[u % 3]) : "v"(f), "v"(wv));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(fa.x ^ fb.y);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
static void run_perm(const char *name, float *d_out, unsigned long long *d_cyc, const unsigned *src) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe_perm<MODE>), dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc, src);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe_perm<MODE>), dim3(256), dim3(256), 0, 0, d_out, iters, d_cyc, src);
    hipDeviceSynchronize();
    unsigned long long cyc;
    hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("v_mfma_f32_32x32x16_f16 after 4 v_perm_b32, %-44s %6.1f counter ticks / MFMA\n", name, (double)cyc / ((double)iters * 12));
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
    run_h<1>(d_out, d_cyc);
    run_h<2>(d_out, d_cyc);
    run_h<3>(d_out, d_cyc);
    run_h<4>(d_out, d_cyc);
    run_h<6>(d_out, d_cyc);
    unsigned *src;
    hipMalloc(&src, 256 * 6 * 16 + 64);
    hipMemset(src, 0x3c, 256 * 6 * 16 + 64);
    run_perm<2>("(no un-zips)", d_out, d_cyc, src);
    run_perm<0>("into the SAME registers every time:", d_out, d_cyc, src);
    run_perm<1>("into two alternating register sets:", d_out, d_cyc, src);
    return 0;
}
