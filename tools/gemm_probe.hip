// gemm_probe.hip -- the large-batch denoiser GEMMs alone (development probe, not part of the library): exact-fp32 MFMA tiles of
// csrc/pd_gemm_stream.h at the bench's 5 120 token rows (and 15 360 = three contexts' rows in one launch), every tile shape, timed
// with hipEvents; TFLOP/s against the 157.3 TF exact-fp32 MFMA peak.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Iinclude tools/gemm_probe.hip -o tools/gemm_probe && tools/gemm_probe
#include "../posediffusion_amd/csrc/pd_gemm_stream.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
void pd_set_error(const char *, ...) {}

template <int EPI, int WM, int WN, bool ALN, int BARE = 0>
static float run(const float *A, const float *W, const float *bias, float *C, int M, int Nout, int K, int reps, hipStream_t s) {
    static float2 *stats = nullptr;
    if (!stats) (void)hipMalloc(&stats, 15360 * sizeof(float2));
    PdStreamArgs g{A, W, bias, C, M, Nout, K, K, K, stats};
    if (ALN) hipLaunchKernelGGL(pd_ln_stats_kernel<512>, dim3((M + 3) / 4), dim3(256), 0, s, A, stats, M, 1e-5f);
    const size_t lds = (size_t)2 * (64 * WM + 64 * WN) * PD_STREAM_LR * sizeof(float);
    auto kern = pd_gemm_stream_kernel<EPI, WM, WN, ALN, BARE>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = ((M + 64 * WM - 1) / (64 * WM)) * (Nout / (64 * WN));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int EPI, bool ALN, int WM = 1, int WN = 1>
static float run_dma(const float *A, const float *W, const float *bias, float *C, int M, int Nout, int K, int reps, hipStream_t s) {
    static float2 *stats = nullptr;
    if (!stats) (void)hipMalloc(&stats, 15360 * sizeof(float2));
    PdStreamArgs g{A, W, bias, C, M, Nout, K, K, K, stats};
    if (ALN) hipLaunchKernelGGL(pd_ln_stats_kernel<512>, dim3((M + 3) / 4), dim3(256), 0, s, A, stats, M, 1e-5f);
    const size_t lds = (size_t)2 * (64 * WM + 64 * WN) * 32 * sizeof(float);
    auto kern = pd_gemm_dma_kernel<EPI, ALN, WM, WN>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = ((M + 64 * WM - 1) / (64 * WM)) * (Nout / (64 * WN));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
// C of the LDS-DMA kernel against C of the register-staged kernel on the same inputs: must be bitwise equal (EPI 0 / 1: C is not an input)
template <int EPI, bool ALN, int WM = 1, int WN = 1>
static long long diff_dma(const float *A, const float *W, const float *bias, float *C, float *C2, int M, int Nout, int K, hipStream_t s) {
    run<EPI, 1, 1, ALN>(A, W, bias, C, M, Nout, K, 1, s);
    run_dma<EPI, ALN, WM, WN>(A, W, bias, C2, M, Nout, K, 1, s);
    (void)hipStreamSynchronize(s);
    std::vector<float> a((size_t)M * Nout), b((size_t)M * Nout);
    (void)hipMemcpy(a.data(), C, a.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), C2, b.size() * 4, hipMemcpyDeviceToHost);
    long long bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
    return bad;
}

int main(int argc, char **argv) {
    const int Mmax = 15360, Kmax = 1024, Nmax = 1536;
    float *A, *W, *bias, *C;
    (void)hipMalloc(&A, (size_t)Mmax * Kmax * 4);
    (void)hipMalloc(&W, (size_t)Nmax * Kmax * 4);
    (void)hipMalloc(&bias, Nmax * 4);
    (void)hipMalloc(&C, (size_t)Mmax * Nmax * 4);
    std::vector<float> h((size_t)Mmax * Kmax);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h.data(), (size_t)Nmax * Kmax * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, h.data(), Nmax * 4, hipMemcpyHostToDevice);
    (void)hipMemset(C, 0, (size_t)Mmax * Nmax * 4);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    struct Shape { const char *name; int N, K; };
    const Shape shapes[] = {{"QKV  512->1536", 1536, 512}, {"FF1  512->1024", 1024, 512}, {"out  512-> 512", 512, 512}, {"FF2 1024-> 512", 512, 1024}};
    printf("%-16s %6s %10s %10s %10s %10s   (TFLOP/s, exact-fp32 MFMA; peak 157.3)\n", "GEMM", "rows", "64x64", "128x64", "64x128", "128x128");
    for (int M : {5120, 15360})
        for (const Shape &sh : shapes) {
            const double gf = 2.0 * M * sh.N * sh.K * 1e-9;
            float t[4];
            if (sh.K == 512 && sh.N > 512) {      // LayerNorm in the staging, as the engine runs them
                t[0] = run<0, 1, 1, true>(A, W, bias, C, M, sh.N, sh.K, 20, s);
                t[1] = run<0, 2, 1, true>(A, W, bias, C, M, sh.N, sh.K, 20, s);
                t[2] = run<0, 1, 2, true>(A, W, bias, C, M, sh.N, sh.K, 20, s);
                t[3] = run<0, 2, 2, true>(A, W, bias, C, M, sh.N, sh.K, 20, s);
            } else {
                t[0] = run<2, 1, 1, false>(A, W, bias, C, M, sh.N, sh.K, 20, s);
                t[1] = run<2, 2, 1, false>(A, W, bias, C, M, sh.N, sh.K, 20, s);
                t[2] = run<2, 1, 2, false>(A, W, bias, C, M, sh.N, sh.K, 20, s);
                t[3] = run<2, 2, 2, false>(A, W, bias, C, M, sh.N, sh.K, 20, s);
            }
            printf("%-16s %6d", sh.name, M);
            for (int i = 0; i < 4; ++i) t[i] > 0 ? printf(" %5.1f(%3.0fus)", gf / t[i], t[i] * 1e3) : printf(" %10s", "-");
            printf("\n");
            // the same GEMM without the LayerNorm staging, for the cost of the pre-pass
            if (sh.K == 512 && sh.N > 512) {
                const float a = run<0, 1, 1, false>(A, W, bias, C, M, sh.N, sh.K, 20, s), b = run<0, 2, 1, false>(A, W, bias, C, M, sh.N, sh.K, 20, s),
                            c = run<0, 1, 2, false>(A, W, bias, C, M, sh.N, sh.K, 20, s), d = run<0, 2, 2, false>(A, W, bias, C, M, sh.N, sh.K, 20, s);
                printf("%-16s %6d %5.1f(%3.0fus) %5.1f(%3.0fus) %5.1f(%3.0fus) %5.1f(%3.0fus)\n", "  (no LayerNorm)", M, gf / a, a * 1e3, gf / b, b * 1e3, gf / c, c * 1e3,
                       gf / d, d * 1e3);
            }
        }
    {
        float *C2;
        (void)hipMalloc(&C2, (size_t)Mmax * Nmax * 4);
        printf("LDS-DMA staging (pd_gemm_dma_kernel) vs register staging, 64x64 tiles; elements that differ: QKV+LN %lld, FF1+LN %lld, plain 512->512 %lld, ragged 5000 rows %lld\n",
               diff_dma<0, true>(A, W, bias, C, C2, 5120, 1536, 512, s), diff_dma<1, true>(A, W, bias, C, C2, 5120, 1024, 512, s),
               diff_dma<0, false>(A, W, bias, C, C2, 5120, 512, 1024, s), diff_dma<0, false>(A, W, bias, C, C2, 5000, 512, 512, s));
        printf("LDS-DMA tiles 128x64 / 64x128 / 128x128 vs 64x64, elements that differ: %lld %lld %lld (QKV+LN), %lld (ragged 5000 rows, 128x128)\n",
               diff_dma<0, true, 2, 1>(A, W, bias, C, C2, 5120, 1536, 512, s), diff_dma<0, true, 1, 2>(A, W, bias, C, C2, 5120, 1536, 512, s),
               diff_dma<0, true, 2, 2>(A, W, bias, C, C2, 5120, 1536, 512, s), diff_dma<0, false, 2, 2>(A, W, bias, C, C2, 5000, 512, 512, s));
        for (int M : {5120, 15360}) {
            const double q = 2.0 * M * 1536 * 512 * 1e-9, f1 = 2.0 * M * 1024 * 512 * 1e-9, f2 = 2.0 * M * 512 * 1024 * 1e-9;
            printf("  %5d rows, LDS-DMA TFLOP/s 64x64 / 128x64 / 64x128 / 128x128:  QKV+LN %5.1f %5.1f %5.1f %5.1f   FF1+LN %5.1f %5.1f %5.1f %5.1f   FF2 %5.1f %5.1f %5.1f %5.1f\n", M,
                   q / run_dma<0, true>(A, W, bias, C, M, 1536, 512, 20, s), q / run_dma<0, true, 2, 1>(A, W, bias, C, M, 1536, 512, 20, s),
                   q / run_dma<0, true, 1, 2>(A, W, bias, C, M, 1536, 512, 20, s), q / run_dma<0, true, 2, 2>(A, W, bias, C, M, 1536, 512, 20, s),
                   f1 / run_dma<1, true>(A, W, bias, C, M, 1024, 512, 20, s), f1 / run_dma<1, true, 2, 1>(A, W, bias, C, M, 1024, 512, 20, s),
                   f1 / run_dma<1, true, 1, 2>(A, W, bias, C, M, 1024, 512, 20, s), f1 / run_dma<1, true, 2, 2>(A, W, bias, C, M, 1024, 512, 20, s),
                   f2 / run_dma<2, false>(A, W, bias, C, M, 512, 1024, 20, s), f2 / run_dma<2, false, 2, 1>(A, W, bias, C, M, 512, 1024, 20, s),
                   f2 / run_dma<2, false, 1, 2>(A, W, bias, C, M, 512, 1024, 20, s), f2 / run_dma<2, false, 2, 2>(A, W, bias, C, M, 512, 1024, 20, s));
        }
        for (int M : {5120, 15360}) {
            const double q = 2.0 * M * 1536 * 512 * 1e-9, f1 = 2.0 * M * 1024 * 512 * 1e-9, o = 2.0 * M * 512 * 512 * 1e-9, f2 = 2.0 * M * 512 * 1024 * 1e-9;
            printf("  %5d rows, TFLOP/s register -> LDS-DMA:  QKV+LN %5.1f -> %5.1f   FF1+LN %5.1f -> %5.1f   out %5.1f -> %5.1f   FF2 %5.1f -> %5.1f\n", M,
                   q / run<0, 1, 1, true>(A, W, bias, C, M, 1536, 512, 20, s), q / run_dma<0, true>(A, W, bias, C, M, 1536, 512, 20, s),
                   f1 / run<1, 1, 1, true>(A, W, bias, C, M, 1024, 512, 20, s), f1 / run_dma<1, true>(A, W, bias, C, M, 1024, 512, 20, s),
                   o / run<2, 1, 1, false>(A, W, bias, C, M, 512, 512, 20, s), o / run_dma<2, false>(A, W, bias, C, M, 512, 512, 20, s),
                   f2 / run<2, 1, 1, false>(A, W, bias, C, M, 512, 1024, 20, s), f2 / run_dma<2, false>(A, W, bias, C, M, 512, 1024, 20, s));
        }
    }
    // where the matrix pipe's idle time comes from: the K loop stripped step by step (QKV shape, no LayerNorm; results meaningless)
    printf("bisect (QKV 512->1536, TFLOP/s): tile    full   no-global/no-LDS-store   +no-barrier   +no-fragment-reads\n");
    for (int M : {5120, 15360}) {
        const double gf = 2.0 * M * 1536 * 512 * 1e-9;
        const float a0 = run<0, 1, 1, false, 0>(A, W, bias, C, M, 1536, 512, 20, s), a1 = run<0, 1, 1, false, 1>(A, W, bias, C, M, 1536, 512, 20, s),
                    a2 = run<0, 1, 1, false, 2>(A, W, bias, C, M, 1536, 512, 20, s), a3 = run<0, 1, 1, false, 3>(A, W, bias, C, M, 1536, 512, 20, s);
        const float a4 = run<0, 1, 1, false, 4>(A, W, bias, C, M, 1536, 512, 20, s), a5 = run<0, 1, 1, false, 5>(A, W, bias, C, M, 1536, 512, 20, s);
        printf("  %5d rows  64x64   %6.1f %6.1f %6.1f %6.1f   global loads only %6.1f   LDS stores only %6.1f\n", M, gf / a0, gf / a1, gf / a2, gf / a3, gf / a4, gf / a5);
        const float b0 = run<0, 2, 2, false, 0>(A, W, bias, C, M, 1536, 512, 20, s), b1 = run<0, 2, 2, false, 1>(A, W, bias, C, M, 1536, 512, 20, s),
                    b2 = run<0, 2, 2, false, 2>(A, W, bias, C, M, 1536, 512, 20, s), b3 = run<0, 2, 2, false, 3>(A, W, bias, C, M, 1536, 512, 20, s);
        printf("  %5d rows  128x128 %6.1f %6.1f %6.1f %6.1f\n", M, gf / b0, gf / b1, gf / b2, gf / b3);
    }
    return 0;
}
