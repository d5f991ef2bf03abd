// valu_probe2.hip -- what the fp32 vector pipe of one gfx950 SIMD really issues per unit of WALL time (development probe, not part
// of the library).  Round 2's probe timed with s_memtime only and let hipcc SLP-pack its "plain" loop, so its figures could not be
// turned into a ceiling.  Here every stream is spelled in inline asm (32 independent destinations per block, no dependencies inside
// a block), one workgroup per CU (LDS-padded), W = 1..4 waves per SIMD, timed with hipEvents over a launch of ~1 ms; s_memtime is
// reported beside it, which calibrates the tick.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe2.hip -o tools/valu_probe2 && tools/valu_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define R8(X, a) X(a##0) X(a##1) X(a##2) X(a##3) X(a##4) X(a##5) X(a##6) X(a##7)
// 16 64-bit register pairs v[32:63] as independent destinations, operands v[0:7]
#define BLK_PKFMA                                                                                                             \
    "v_pk_fma_f32 v[32:33], v[32:33], v[0:1], v[2:3]\n v_pk_fma_f32 v[34:35], v[34:35], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[36:37], v[36:37], v[0:1], v[2:3]\n v_pk_fma_f32 v[38:39], v[38:39], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[40:41], v[40:41], v[0:1], v[2:3]\n v_pk_fma_f32 v[42:43], v[42:43], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[44:45], v[44:45], v[0:1], v[2:3]\n v_pk_fma_f32 v[46:47], v[46:47], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[48:49], v[48:49], v[0:1], v[2:3]\n v_pk_fma_f32 v[50:51], v[50:51], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[52:53], v[52:53], v[0:1], v[2:3]\n v_pk_fma_f32 v[54:55], v[54:55], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[56:57], v[56:57], v[0:1], v[2:3]\n v_pk_fma_f32 v[58:59], v[58:59], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[60:61], v[60:61], v[0:1], v[2:3]\n v_pk_fma_f32 v[62:63], v[62:63], v[0:1], v[2:3]\n"
#define BLK_PKMUL                                                                                                             \
    "v_pk_mul_f32 v[32:33], v[32:33], v[0:1]\n v_pk_mul_f32 v[34:35], v[34:35], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[36:37], v[36:37], v[0:1]\n v_pk_mul_f32 v[38:39], v[38:39], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[40:41], v[40:41], v[0:1]\n v_pk_mul_f32 v[42:43], v[42:43], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[44:45], v[44:45], v[0:1]\n v_pk_mul_f32 v[46:47], v[46:47], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[48:49], v[48:49], v[0:1]\n v_pk_mul_f32 v[50:51], v[50:51], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[52:53], v[52:53], v[0:1]\n v_pk_mul_f32 v[54:55], v[54:55], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[56:57], v[56:57], v[0:1]\n v_pk_mul_f32 v[58:59], v[58:59], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[60:61], v[60:61], v[0:1]\n v_pk_mul_f32 v[62:63], v[62:63], v[0:1]\n"
#define I1(op, d) op " v" #d ", v" #d ", v0, v2\n"
#define BLK16_3(op)                                                                                                           \
    I1(op, 32) I1(op, 33) I1(op, 34) I1(op, 35) I1(op, 36) I1(op, 37) I1(op, 38) I1(op, 39) I1(op, 40) I1(op, 41) I1(op, 42)  \
        I1(op, 43) I1(op, 44) I1(op, 45) I1(op, 46) I1(op, 47)
#define I2(op, d) op " v" #d ", v" #d ", v0\n"
#define BLK16_2(op)                                                                                                           \
    I2(op, 32) I2(op, 33) I2(op, 34) I2(op, 35) I2(op, 36) I2(op, 37) I2(op, 38) I2(op, 39) I2(op, 40) I2(op, 41) I2(op, 42)  \
        I2(op, 43) I2(op, 44) I2(op, 45) I2(op, 46) I2(op, 47)
#define I1U(op, d) op " v" #d ", v" #d "\n"
#define BLK16_1(op)                                                                                                           \
    I1U(op, 32) I1U(op, 33) I1U(op, 34) I1U(op, 35) I1U(op, 36) I1U(op, 37) I1U(op, 38) I1U(op, 39) I1U(op, 40) I1U(op, 41)  \
        I1U(op, 42) I1U(op, 43) I1U(op, 44) I1U(op, 45) I1U(op, 46) I1U(op, 47)
#define IDPP(d) "v_add_f32_dpp v" #d ", v" #d ", v" #d " row_shl:4 row_mask:0xf bank_mask:0x5\n"
#define BLK16_DPP                                                                                                             \
    IDPP(32) IDPP(33) IDPP(34) IDPP(35) IDPP(36) IDPP(37) IDPP(38) IDPP(39) IDPP(40) IDPP(41) IDPP(42) IDPP(43) IDPP(44)       \
        IDPP(45) IDPP(46) IDPP(47)
#define ICND(d) "v_cndmask_b32 v" #d ", v" #d ", v0, vcc\n"
#define BLK16_CND                                                                                                             \
    ICND(32) ICND(33) ICND(34) ICND(35) ICND(36) ICND(37) ICND(38) ICND(39) ICND(40) ICND(41) ICND(42) ICND(43) ICND(44)       \
        ICND(45) ICND(46) ICND(47)
// the instruction mix of ONE packed two-match Sampson step as sampson_step2 (csrc/pd_ggs.hip) compiles: 28 v_pk_fma, 11 v_pk_mul,
// 4 v_pk_add, 2 v_rcp, 2 v_cmp, 2 v_cndmask, 1 v_min3 = 50 VALU, dependencies as independent as the real step's are after scheduling
#define BLK_STEPMIX                                                                                                           \
    BLK_PKFMA                                                                                                                 \
    "v_pk_fma_f32 v[32:33], v[32:33], v[0:1], v[2:3]\n v_pk_fma_f32 v[34:35], v[34:35], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[36:37], v[36:37], v[0:1], v[2:3]\n v_pk_fma_f32 v[38:39], v[38:39], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[40:41], v[40:41], v[0:1], v[2:3]\n v_pk_fma_f32 v[42:43], v[42:43], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[44:45], v[44:45], v[0:1], v[2:3]\n v_pk_fma_f32 v[46:47], v[46:47], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[48:49], v[48:49], v[0:1], v[2:3]\n v_pk_fma_f32 v[50:51], v[50:51], v[0:1], v[2:3]\n"                     \
    "v_pk_fma_f32 v[52:53], v[52:53], v[0:1], v[2:3]\n v_pk_fma_f32 v[54:55], v[54:55], v[0:1], v[2:3]\n"                     \
    "v_pk_mul_f32 v[56:57], v[56:57], v[0:1]\n v_pk_mul_f32 v[58:59], v[58:59], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[60:61], v[60:61], v[0:1]\n v_pk_mul_f32 v[62:63], v[62:63], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[32:33], v[32:33], v[0:1]\n v_pk_mul_f32 v[34:35], v[34:35], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[36:37], v[36:37], v[0:1]\n v_pk_mul_f32 v[38:39], v[38:39], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[40:41], v[40:41], v[0:1]\n v_pk_mul_f32 v[42:43], v[42:43], v[0:1]\n"                                     \
    "v_pk_mul_f32 v[44:45], v[44:45], v[0:1]\n"                                                                               \
    "v_pk_add_f32 v[46:47], v[46:47], v[0:1]\n v_pk_add_f32 v[48:49], v[48:49], v[0:1]\n"                                     \
    "v_pk_add_f32 v[50:51], v[50:51], v[0:1]\n v_pk_add_f32 v[52:53], v[52:53], v[0:1]\n"                                     \
    "v_rcp_f32 v54, v54\n v_rcp_f32 v55, v55\n"                                                                               \
    "v_cmp_lt_f32 vcc, v56, v0\n v_cndmask_b32 v57, v57, v0, vcc\n"                                                           \
    "v_cmp_lt_f32 vcc, v58, v0\n v_cndmask_b32 v59, v59, v0, vcc\n"                                                           \
    "v_min3_f32 v60, v60, |v61|, |v62|\n"

#define CLOBBER                                                                                                               \
    "v0", "v1", "v2", "v3", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", \
        "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc"

template <int MODE>
__global__ void probe(float *out, long long *cyc, int rep, float a) {
    extern __shared__ float pad[];
    asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, %0\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 0.5\n" ::"v"(a) : "v0", "v1", "v2", "v3");
    __syncthreads();
    const long long m0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        if (MODE == 0) asm volatile(BLK16_3("v_fma_f32") BLK16_3("v_fma_f32")::: CLOBBER);
        if (MODE == 1) asm volatile(BLK_PKFMA BLK_PKFMA ::: CLOBBER);
        if (MODE == 2) asm volatile(BLK_PKMUL BLK_PKMUL ::: CLOBBER);
        if (MODE == 3) asm volatile(BLK16_2("v_mul_f32") BLK16_2("v_mul_f32")::: CLOBBER);
        if (MODE == 4) asm volatile(BLK16_2("v_add_f32") BLK16_2("v_add_f32")::: CLOBBER);
        if (MODE == 5) asm volatile(BLK16_DPP BLK16_DPP ::: CLOBBER);
        if (MODE == 6) asm volatile(BLK16_CND BLK16_CND ::: CLOBBER);
        if (MODE == 7) asm volatile(BLK16_1("v_rcp_f32") BLK16_1("v_rcp_f32")::: CLOBBER);
        if (MODE == 8) asm volatile(BLK16_1("v_mov_b32") BLK16_1("v_mov_b32")::: CLOBBER);
        if (MODE == 9) asm volatile(BLK_STEPMIX ::: CLOBBER);
    }
    const long long m1 = __builtin_amdgcn_s_memtime();
    float s;
    asm volatile("v_add_f32 %0, v32, v63" : "=v"(s)::);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + pad[threadIdx.x & 7];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = m1 - m0;
}

typedef void (*kern_t)(float *, long long *, int, float);
int main() {
    float *out;
    long long *cyc, h;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 16);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_mul_f32 (VOP2)", "v_add_f32 (VOP2)", "v_add_f32_dpp row_shl bank-masked",
                           "v_cndmask_b32", "v_rcp_f32", "v_mov_b32", "sampson_step2 mix (50 VALU)"};
    const int per_block[] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 50};
    kern_t ks[] = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 96 * 1024;   // one workgroup per CU
    for (int m = 0; m < 10; ++m) hipFuncSetAttribute((const void *)ks[m], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("%-36s %5s %12s %12s %12s %10s\n", "stream", "W/SIMD", "ns/instr/SIMD", "tick/instr", "ticks/us", "chip Ginstr/s");
    for (int m = 0; m < 10; ++m)
        for (int w = 1; w <= 4; ++w) {
            const int rep = 20000;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(ks[m], dim3(256), dim3(256 * w), lds, 0, out, cyc, rep, 1.0001f);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double n = (double)rep * per_block[m] * w;     // wave-instructions per SIMD
            printf("%-36s %5d %12.3f %12.3f %12.1f %10.1f\n", names[m], w, ms * 1e6 / n, (double)h / n, (double)h / (ms * 1e3), n * 1024 / (ms * 1e6));
        }
    return 0;
}
