// valu_probe3.hip -- dependent-issue latency of packed fp32 on one gfx950 SIMD: C independent chains of v_pk_fma_f32 (each instruction of a
// chain depends on the previous one of that chain), C = 1, 2, 4, 8, 16; W = 1..3 waves per SIMD.  ticks (s_memtime) per instruction per
// wave and per SIMD.  Development probe.   hipcc --offload-arch=gfx950 -O3 tools/valu_probe3.hip -o tools/valu_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
#define F(d) "v_pk_fma_f32 v[" #d "], v[" #d "], v[0:1], v[2:3]\n"
#define C1 F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33) F(32:33)
#define C2 F(32:33) F(34:35) F(32:33) F(34:35) F(32:33) F(34:35) F(32:33) F(34:35) F(32:33) F(34:35) F(32:33) F(34:35) F(32:33) F(34:35) F(32:33) F(34:35)
#define C4 F(32:33) F(34:35) F(36:37) F(38:39) F(32:33) F(34:35) F(36:37) F(38:39) F(32:33) F(34:35) F(36:37) F(38:39) F(32:33) F(34:35) F(36:37) F(38:39)
#define C8 F(32:33) F(34:35) F(36:37) F(38:39) F(40:41) F(42:43) F(44:45) F(46:47) F(32:33) F(34:35) F(36:37) F(38:39) F(40:41) F(42:43) F(44:45) F(46:47)
#define C16 F(32:33) F(34:35) F(36:37) F(38:39) F(40:41) F(42:43) F(44:45) F(46:47) F(48:49) F(50:51) F(52:53) F(54:55) F(56:57) F(58:59) F(60:61) F(62:63)
// a v_mul_f32 (VOP2) dependent chain and a mixed chain pk_fma -> v_rcp -> pk_mul for comparison
#define M(d) "v_mul_f32 v" #d ", v" #d ", v0\n"
#define CM1 M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32) M(32)
#define CLOB "v0", "v1", "v2", "v3", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
    "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
template <int MODE>
__global__ void probe(float *out, long long *cyc, int rep, float a) {
    extern __shared__ float pad[];
    asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 0.5\n" ::"v"(a) : "v0", "v1", "v2", "v3");
    __syncthreads();
    const long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        if (MODE == 0) asm volatile(C1 ::: CLOB);
        if (MODE == 1) asm volatile(C2 ::: CLOB);
        if (MODE == 2) asm volatile(C4 ::: CLOB);
        if (MODE == 3) asm volatile(C8 ::: CLOB);
        if (MODE == 4) asm volatile(C16 ::: CLOB);
        if (MODE == 5) asm volatile(CM1 ::: CLOB);
    }
    const long long m1 = __builtin_amdgcn_s_memtime();
    float s;
    asm volatile("v_add_f32 %0, v32, v63" : "=v"(s)::);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + pad[threadIdx.x & 7];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = m1 - m0;
}
typedef void (*kern_t)(float *, long long *, int, float);
int main() {
    float *out; long long *cyc, h;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 16);
    const char *names[] = {"pk_fma, 1 chain (fully dependent)", "pk_fma, 2 chains", "pk_fma, 4 chains", "pk_fma, 8 chains", "pk_fma, 16 chains", "v_mul_f32 VOP2, 1 chain"};
    kern_t ks[] = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>};
    const size_t lds = 96 * 1024;
    for (int m = 0; m < 6; ++m) (void)hipFuncSetAttribute((const void *)ks[m], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("%-36s %6s %16s %16s\n", "stream", "W/SIMD", "ticks/instr/wave", "ticks/instr/SIMD");
    for (int m = 0; m < 6; ++m)
        for (int w = 1; w <= 3; ++w) {
            const int rep = 4000;
            for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(ks[m], dim3(256), dim3(256 * w), lds, 0, out, cyc, rep, 1.0001f); (void)hipDeviceSynchronize(); }
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-36s %6d %16.2f %16.2f\n", names[m], w, (double)h / (rep * 16.0), (double)h / (rep * 16.0) / w);
        }
    return 0;
}
