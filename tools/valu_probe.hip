// valu_probe.hip -- issue rate of plain vs packed fp32 VALU on gfx950 (development probe, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe && tools/valu_probe
// Each wave runs N_OPS dependent-free instructions in 16 independent chains; reports cycles per wave-instruction per SIMD
// at 1 and 2 waves per SIMD (s_memtime around the loop, one workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 256
template <int MODE>
__global__ void probe(float *out, long long *cyc, float a, float b) {
    float x[16];
    v2f y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = a + i + threadIdx.x; y[i] = (v2f){a + i, b + threadIdx.x}; }
    const v2f aa = {a, b}, bb = {b, a};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    const long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);                       // v_fma_f32
            if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], aa, bb);          // v_pk_fma_f32
            if (MODE == 2) x[i] = x[i] > a ? b : x[i] + 1.0f;                       // v_cmp + v_cndmask + v_add
            if (MODE == 3) x[i] = __builtin_amdgcn_rcpf(x[i]);                      // v_rcp_f32
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long m1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = m1 - m0; }
}
int main() {
    float *out; long long *cyc, h[2];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "cmp+cndmask+add (3 instr)", "v_rcp_f32"};
    for (int mode = 0; mode < 4; ++mode)
        for (int threads : {256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            const double n = 16.0 * REP * (mode == 2 ? 3 : 1);
            printf("%-28s %d waves/SIMD: %.2f shader cycles per wave-instruction (per wave), %.2f per SIMD issue slot; memtime ticks %lld\n",
                   names[mode], threads / 256, h[0] / n, h[0] / n / (threads / 256), h[1]);
        }
    return 0;
}
