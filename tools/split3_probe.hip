// split3_probe.hip -- the six-product bf16 GEMM (tools/pd_gemm_split3.h) alone (development probe, not part of the library): error against
// an fp64 CPU product on sampled rows next to the exact-fp32 MFMA kernel's, and time per tile shape at the bench's row counts.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Iinclude tools/split3_probe.hip -o tools/split3_probe && tools/split3_probe
#include "pd_gemm_split3.h"
#include "../posediffusion_amd/csrc/pd_gemm_split.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
void pd_set_error(const char *, ...) {}

static float2 *g_stats;
static float time_it(void (*launch)(hipStream_t), int reps, hipStream_t s) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch(s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) launch(s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
static const float *gA, *gW, *gBias;
static const uint4 *gW3;
static float *gC;
static int gM, gN, gK;
template <int EPI, bool ALN, int WM, int WN>
static void l_split3(hipStream_t s) {
    auto kern = pd_gemm_split3_kernel<EPI, ALN, WM, WN>;
    (void)kern;
    pd_gemm_split3<EPI, ALN, WM, WN>(gA, gK, gW3, gK, gBias, gC, gM, gN, s, g_stats);
}
static const unsigned *gW2;
template <int EPI, int WM, int WN>
static void l_split2(hipStream_t s) { pd_gemm_split<EPI, WM, WN>((const unsigned *)gA, gK, gW2, gK, gBias, gC, gM, gN, s); }
template <int RT, int BARE>
static void l_strip_bare(hipStream_t s) {
    VitSplitArgs g{(const unsigned *)gA, gW2, gBias, gC, gM, gN, gK, gK, 0.5f, 4.0f};
    hipLaunchKernelGGL((pd_gemm_strip_kernel<0, RT, true, 1, BARE>), dim3(((gM + 32 * RT - 1) / (32 * RT)) * (gN / 128)), dim3(256), (size_t)2 * 32 * RT * 32 * sizeof(unsigned), s, g);
}
template <int EPI, int RT, int CT = 1>
static void l_strip(hipStream_t s) { pd_gemm_strip<EPI, RT, true, CT>((const unsigned *)gA, gK, gW2, gK, gBias, gC, gM, gN, s, 0.5f, 4.0f); }
template <int EPI, int WM, int WN>
static void l_split2h(hipStream_t s) { pd_gemm_split<EPI, WM, WN, true>((const unsigned *)gA, gK, gW2, gK, gBias, gC, gM, gN, s, 0.5f, 4.0f); }
// C of the strip kernel against C of vit_gemm_split_kernel on the same operands (arbitrary bit patterns read as fp16 pairs are fine for
// a bitwise comparison only if no NaN arises: the operands are made of small integers instead)
template <int EPI, int RT, int CT = 1>
static long long diff_strip(float *C2, int M, int N, int K, hipStream_t s) {
    gM = M; gN = N; gK = K;
    float *keep = gC;
    (void)hipMemsetAsync(gC, 0, (size_t)M * N * 4, s);
    (void)hipMemsetAsync(C2, 0, (size_t)M * N * 4, s);
    l_split2h<EPI, 1, 2>(s);
    gC = C2;
    l_strip<EPI, RT, CT>(s);
    gC = keep;
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> a((size_t)M * N), b((size_t)M * N);
    (void)hipMemcpy(a.data(), gC, a.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), C2, b.size() * 4, hipMemcpyDeviceToHost);
    long long bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    return bad;
}
template <int WM, int WN, int BARE>
static void l_bare(hipStream_t s) {
    VitSplitArgs g{(const unsigned *)gA, gW2, gBias, gC, gM, gN, gK, gK, 1.0f, 1.0f};
    constexpr int TM = 64 * WM, TN = 64 * WN;
    hipLaunchKernelGGL((vit_gemm_split_kernel<0, WM, WN, true, BARE>), dim3(((gM + TM - 1) / TM) * (gN / TN)), dim3(256), pd_split_lds(WM), s, g);
}
template <int EPI, bool ALN>
static void l_exact(hipStream_t s) { pd_gemm_dma<EPI, ALN>(gA, gK, gW, gK, gBias, gC, gM, gN, s, g_stats); }

int main() {
    const int Mmax = 15360, Kmax = 1024, Nmax = 1536;
    float *A, *W, *bias, *C;
    uint4 *W3;
    (void)hipMalloc(&A, (size_t)Mmax * Kmax * 4);
    (void)hipMalloc(&W, (size_t)Nmax * Kmax * 4);
    (void)hipMalloc(&W3, pd_split3_weight_bytes(Nmax, Kmax));
    (void)hipMalloc(&bias, Nmax * 4);
    (void)hipMalloc(&C, (size_t)Mmax * Nmax * 4);
    (void)hipMalloc(&g_stats, Mmax * sizeof(float2));
    std::vector<float> hA((size_t)Mmax * Kmax), hW((size_t)Nmax * Kmax), hb(Nmax);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    auto gauss = [&]() { return sqrt(-2.0 * log(rnd() + 1e-300)) * cos(6.283185307179586 * rnd()); };
    for (auto &v : hA) v = (float)gauss();
    for (auto &v : hW) v = (float)(0.02 * gauss());
    for (auto &v : hb) v = (float)(0.1 * gauss());
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    gA = A; gW = W; gBias = bias; gC = C; gW3 = W3;
    (void)hipFuncSetAttribute((const void *)pd_gemm_split3_kernel<0, false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pd_split3_lds(2));

    // ---- accuracy: 512 -> 1536 and 1024 -> 512 on 1 000 rows (ragged: not a multiple of 128), against fp64 on the host
    for (int K : {512, 1024}) {
        const int N = K == 512 ? 1536 : 512, M = 1000;
        gM = M; gN = N; gK = K;
        const size_t total = (size_t)(N / 32) * (K / 16) * 64;
        hipLaunchKernelGGL(pd_frag_split3_kernel, dim3(512), dim3(256), 0, s, W, K, total, W3);
        std::vector<double> ref((size_t)M * N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double a = hb[n];
                for (int k = 0; k < K; ++k) a += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
                ref[(size_t)m * N + n] = a;
            }
        double rmax = 0;
        for (double v : ref) rmax = fmax(rmax, fabs(v));
        std::vector<float> out((size_t)M * N);
        auto err = [&](const char *tag) {
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost);
            double emax = 0, e2 = 0;
            for (size_t i = 0; i < out.size(); ++i) {
                const double e = fabs((double)out[i] - ref[i]);
                emax = fmax(emax, e);
                e2 += e * e;
            }
            printf("  %-34s max |err| / max |C| = %.3e   rms err / max |C| = %.3e\n", tag, emax / rmax, sqrt(e2 / out.size()) / rmax);
        };
        printf("K = %d -> %d columns, %d rows, A ~ N(0,1), W ~ N(0, 0.02^2); error against the fp64 product:\n", K, N, M);
        (void)hipMemsetAsync(C, 0, out.size() * 4, s); l_exact<0, false>(s); err("exact fp32 MFMA (pd_gemm_dma)");
        (void)hipMemsetAsync(C, 0, out.size() * 4, s); l_split3<0, false, 2, 2>(s); err("bf16 x 3, six products, 128x128");
        (void)hipMemsetAsync(C, 0, out.size() * 4, s); l_split3<0, false, 1, 1>(s); err("bf16 x 3, six products, 64x64");
        (void)hipMemsetAsync(C, 0, out.size() * 4, s); l_split3<0, false, 2, 1>(s); err("bf16 x 3, six products, 128x64");
        (void)hipMemsetAsync(C, 0, out.size() * 4, s); l_split3<0, false, 1, 2>(s); err("bf16 x 3, six products, 64x128");
    }
    // ---- time: the two-plane, three-product kernel of the fast mode (vit_gemm_split_kernel) for comparison (operands: any bits)
    gW2 = (const unsigned *)W3;
    (void)hipFuncSetAttribute((const void *)vit_gemm_split_kernel<2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pd_split_lds(2));
    (void)hipFuncSetAttribute((const void *)vit_gemm_split_kernel<2, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pd_split_lds(2));
    printf("two planes, three products (vit_gemm_split_kernel, EPI 2), useful TFLOP/s by tile 64x64 / 128x64 / 64x128 / 128x128 (bf16 peak / 3 = 838):\n");
    for (int M : {5120, 15360})
        for (int sh = 0; sh < 4; ++sh) {
            const int Ns[] = {1536, 1024, 512, 512}, Ks[] = {512, 512, 512, 1024};
            gM = M; gN = Ns[sh]; gK = Ks[sh];
            const double gf = 2.0 * M * gN * gK * 1e-9;
            const float a = time_it(l_split2<2, 1, 1>, 20, s), b = time_it(l_split2<2, 2, 1>, 20, s), c = time_it(l_split2<2, 1, 2>, 20, s), d = time_it(l_split2<2, 2, 2>, 20, s);
            printf("  %4d->%4d %6d rows: %5.1f(%3.0fus) %5.1f(%3.0fus) %5.1f(%3.0fus) %5.1f(%3.0fus)\n", gK, gN, M, gf / a, a * 1e3, gf / b, b * 1e3, gf / c, c * 1e3, gf / d, d * 1e3);
        }
    {
        // operands for the bitwise comparison: split words of small multiples of 1/8 (exact in fp16, no overflow)
        std::vector<unsigned> wa((size_t)Mmax * Kmax);
        for (size_t i = 0; i < wa.size(); ++i) {
            const _Float16 h = (_Float16)((float)((int)((i * 2654435761u) >> 20) % 97 - 48) * 0.125f), l = (_Float16)((float)((int)((i * 40503u) >> 7) % 31 - 15) * 0.0009765625f);
            unsigned short hb, lb;
            memcpy(&hb, &h, 2); memcpy(&lb, &l, 2);
            wa[i] = (unsigned)hb | ((unsigned)lb << 16);
        }
        unsigned *Aw;
        float *C2;
        (void)hipMalloc(&Aw, wa.size() * 4);
        (void)hipMalloc(&C2, (size_t)Mmax * Nmax * 4);
        (void)hipMemcpy(Aw, wa.data(), wa.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy((void *)W3, wa.data(), (size_t)Nmax * Kmax * 4, hipMemcpyHostToDevice);     // the same words as weight fragments
        const float *keepA = gA;
        gA = (const float *)Aw;
        printf("strip kernel vs vit_gemm_split_kernel<.., F16>, elements that differ: QKV (EPI 0, RT 4) %lld, ragged 5000 rows (EPI 0, RT 2) %lld, FF1 (EPI 4, RT 4) %lld, "
               "FF2 (EPI 2, RT 2, K 1024) %lld, out (EPI 2, RT 4, 1000 rows) %lld\n",
               diff_strip<0, 4>(C2, 5120, 1536, 512, s), diff_strip<0, 2>(C2, 5000, 512, 512, s), diff_strip<4, 4>(C2, 5120, 1024, 512, s),
               diff_strip<2, 2>(C2, 5120, 512, 1024, s), diff_strip<2, 4>(C2, 1000, 512, 512, s));
        printf("  two column tiles per wave: QKV (EPI 0, RT 2) %lld, FF1 (EPI 4, RT 4) %lld, ragged 5000 rows FF2 (EPI 2, RT 2) %lld\n",
               diff_strip<0, 2, 2>(C2, 5120, 1536, 512, s), diff_strip<4, 4, 2>(C2, 5120, 1024, 512, s), diff_strip<2, 2, 2>(C2, 5000, 512, 1024, s));
        printf("strip kernel, us per launch (64 x 128 | 128 x 128 | 64 x 256 | 128 x 256 tile) against vit_gemm_split_kernel 64 x 128 | 64 x 64:\n");
        for (int M : {5120, 15360}) {
            struct { const char *n; int N, K, epi; } sh[] = {{"QKV", 1536, 512, 0}, {"FF1", 1024, 512, 4}, {"out", 512, 512, 2}, {"FF2", 512, 1024, 2}};
            for (auto &x : sh) {
                gM = M; gN = x.N; gK = x.K;
                float t[6];
                if (x.epi == 0) { t[0] = time_it(l_strip<0, 2>, 20, s); t[1] = time_it(l_strip<0, 4>, 20, s); t[2] = time_it(l_split2h<0, 1, 2>, 20, s); t[3] = time_it(l_split2h<0, 1, 1>, 20, s); t[4] = time_it(l_strip<0, 2, 2>, 20, s); t[5] = time_it(l_strip<0, 4, 2>, 20, s); }
                else if (x.epi == 4) { t[0] = time_it(l_strip<4, 2>, 20, s); t[1] = time_it(l_strip<4, 4>, 20, s); t[2] = time_it(l_split2h<4, 1, 2>, 20, s); t[3] = time_it(l_split2h<4, 1, 1>, 20, s); t[4] = time_it(l_strip<4, 2, 2>, 20, s); t[5] = time_it(l_strip<4, 4, 2>, 20, s); }
                else { t[0] = time_it(l_strip<2, 2>, 20, s); t[1] = time_it(l_strip<2, 4>, 20, s); t[2] = time_it(l_split2h<2, 1, 2>, 20, s); t[3] = time_it(l_split2h<2, 1, 1>, 20, s); t[4] = time_it(l_strip<2, 2, 2>, 20, s); t[5] = time_it(l_strip<2, 4, 2>, 20, s); }
                printf("  %s %5d rows: %5.1f | %5.1f | %5.1f | %5.1f   against %5.1f | %5.1f\n", x.n, M, t[0] * 1e3, t[1] * 1e3, t[4] * 1e3, t[5] * 1e3, t[2] * 1e3, t[3] * 1e3);
            }
        }
        printf("what bounds the strip kernel (EPI 0, 64 x 128 tile), us per launch: full | no W loads | no DMA | neither | neither, no barrier | no MFMA | no fragment reads\n");
        for (int M : {5120, 15360})
            for (int sh = 0; sh < 2; ++sh) {
                const int Ns[] = {1536, 512}, Ks[] = {512, 1024};
                gM = M; gN = Ns[sh]; gK = Ks[sh];
                printf("  %4d->%4d %6d rows: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f\n", gK, gN, M, 1e3 * time_it(l_strip_bare<2, 0>, 20, s), 1e3 * time_it(l_strip_bare<2, 1>, 20, s),
                       1e3 * time_it(l_strip_bare<2, 2>, 20, s), 1e3 * time_it(l_strip_bare<2, 3>, 20, s), 1e3 * time_it(l_strip_bare<2, 4>, 20, s), 1e3 * time_it(l_strip_bare<2, 5>, 20, s),
                       1e3 * time_it(l_strip_bare<2, 6>, 20, s));
            }
        gA = keepA;
        (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    }
    printf("what bounds the two-plane kernel (fp16, EPI 0), us per launch: full | no W loads | no A staging | neither | neither, no barrier | no MFMA\n");
    for (int M : {5120, 15360})
        for (int sh = 0; sh < 2; ++sh) {
            const int Ns[] = {1536, 512}, Ks[] = {512, 1024};
            gM = M; gN = Ns[sh]; gK = Ks[sh];
            printf("  %4d->%4d %6d rows  64x128: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f   128x128: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f\n", gK, gN, M,
                   1e3 * time_it(l_bare<1, 2, 0>, 20, s), 1e3 * time_it(l_bare<1, 2, 1>, 20, s), 1e3 * time_it(l_bare<1, 2, 2>, 20, s), 1e3 * time_it(l_bare<1, 2, 3>, 20, s),
                   1e3 * time_it(l_bare<1, 2, 4>, 20, s), 1e3 * time_it(l_bare<1, 2, 5>, 20, s),
                   1e3 * time_it(l_bare<2, 2, 0>, 20, s), 1e3 * time_it(l_bare<2, 2, 1>, 20, s), 1e3 * time_it(l_bare<2, 2, 2>, 20, s), 1e3 * time_it(l_bare<2, 2, 3>, 20, s),
                   1e3 * time_it(l_bare<2, 2, 4>, 20, s), 1e3 * time_it(l_bare<2, 2, 5>, 20, s));
        }
    // ---- time
    struct Shape { const char *name; int N, K; };
    const Shape shapes[] = {{"QKV  512->1536", 1536, 512}, {"FF1  512->1024", 1024, 512}, {"out  512-> 512", 512, 512}, {"FF2 1024-> 512", 512, 1024}};
    printf("%-16s %6s %12s | bf16 x 3: %10s %10s %10s %10s   (useful TFLOP/s = 2 M N K / time; exact-fp32 MFMA peak 157.3, bf16 peak / 6 = 419)\n", "GEMM", "rows", "exact 64x64",
           "64x64", "128x64", "64x128", "128x128");
    for (int M : {5120, 15360})
        for (const Shape &sh : shapes) {
            gM = M; gN = sh.N; gK = sh.K;
            const size_t total = (size_t)(sh.N / 32) * (sh.K / 16) * 64;
            hipLaunchKernelGGL(pd_frag_split3_kernel, dim3(512), dim3(256), 0, s, W, sh.K, total, W3);
            const bool ln = sh.K == 512 && sh.N > 512;
            if (ln) hipLaunchKernelGGL(pd_ln_stats_kernel<512>, dim3((M + 3) / 4), dim3(256), 0, s, A, g_stats, M, 1e-5f);
            const double gf = 2.0 * M * sh.N * sh.K * 1e-9;
            float t[5];
            if (ln) {
                t[0] = time_it(l_exact<0, true>, 20, s);
                t[1] = time_it(l_split3<0, true, 1, 1>, 20, s); t[2] = time_it(l_split3<0, true, 2, 1>, 20, s);
                t[3] = time_it(l_split3<0, true, 1, 2>, 20, s); t[4] = time_it(l_split3<0, true, 2, 2>, 20, s);
            } else {
                t[0] = time_it(l_exact<2, false>, 20, s);
                t[1] = time_it(l_split3<2, false, 1, 1>, 20, s); t[2] = time_it(l_split3<2, false, 2, 1>, 20, s);
                t[3] = time_it(l_split3<2, false, 1, 2>, 20, s); t[4] = time_it(l_split3<2, false, 2, 2>, 20, s);
            }
            printf("%-16s %6d %5.1f(%3.0fus) |          ", sh.name, M, gf / t[0], t[0] * 1e3);
            for (int i = 1; i < 5; ++i) printf(" %5.1f(%3.0fus)", gf / t[i], t[i] * 1e3);
            printf("%s\n", ln ? "  (+LN)" : "");
        }
    return 0;
}
