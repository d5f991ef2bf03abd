// strip_legs_probe.hip -- where a launch of pd_gemm_strip_kernel<.., F16, K64> goes (round 6; development probe, not part of the library): every workgroup
// stamps {wall start / end (wall_clock64, 100 MHz), shader cycles at entry / first operands landed / K loop done / stores drained, XCC id, HW id}; the host
// prints the launch's timeline (start skew, how many workgroups share a CU) and the legs.  Shapes: the large-batch denoiser's GEMMs at 5 120 rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DPD_STRIP_LEGS -Iinclude tools/strip_legs_probe.hip -o tools/strip_legs_probe && tools/strip_legs_probe
#include "../posediffusion_amd/csrc/pd_gemm_split.h"
#include <algorithm>
#include <map>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
void pd_set_error(const char *, ...) {}

__global__ void fill_words(unsigned *p, size_t n, unsigned seed) {      // split words of small random values (hi = lo = finite fp16 bit patterns)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const float v = ((int)(x & 0xffff) - 32768) * (1.0f / 4096.0f);
        p[i] = pd_split_word_h(v);
    }
}
__global__ void fill_planes(uint4 *p, size_t n, unsigned seed) {       // weight planes: any finite fp16 values
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned h = 0x2c002c00u | (x & 0x03ff03ffu);              // fp16 in [1/16, 1/8)
        p[i] = make_uint4(h, h ^ 0x80000000u, h, h ^ 0x00008000u);
    }
}

template <int EPI, int RT>
static void run(const char *name, int M, int N, int K, unsigned *A, uint4 *W, float *bias, float *C, long long *legs_d, hipStream_t s) {
    constexpr int TM = 32 * RT;
    const int wgs = ((M + TM - 1) / TM) * (N / 128);
    VitSplitArgs g{A, (const unsigned *)W, bias, C, M, N, K, K, 0.5f, 4.0f};
    g.legs = nullptr;
    (void)hipFuncSetAttribute((const void *)pd_gemm_strip_kernel<EPI, RT, true, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pd_gemm_strip_lds<RT, true>());
    auto launch = [&]() { hipLaunchKernelGGL((pd_gemm_strip_kernel<EPI, RT, true, 1, 0, true>), dim3(wgs), dim3(256), (pd_gemm_strip_lds<RT, true>()), s, g); };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    g.legs = legs_d;
    launch();
    launch();
    (void)hipStreamSynchronize(s);
    std::vector<long long> L((size_t)wgs * 8);
    (void)hipMemcpy(L.data(), legs_d, L.size() * 8, hipMemcpyDeviceToHost);
    long long w0 = L[0], w1 = 0;
    for (int b = 0; b < wgs; ++b) { w0 = std::min(w0, L[b * 8]); w1 = std::max(w1, L[b * 8 + 1]); }
    double pro = 0, loop = 0, epi = 0, dur = 0, late = 0, wb = 0, wa = 0, bar = 0;
    std::vector<double> starts, ends;
    for (int b = 0; b < wgs; ++b) {
        const long long *l = &L[b * 8];
        pro += l[3] - l[2]; loop += l[4] - l[3]; epi += l[5] - l[4]; dur += l[5] - l[2];
        starts.push_back((l[0] - w0) * 0.01); ends.push_back((l[1] - w0) * 0.01);
        if ((l[0] - w0) * 0.01 > 2.0) late += 1;
        wb += (double)(l[6] >> 32); wa += (double)(l[6] & 0xffffffffll); bar += (double)l[7];
    }
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
    const double mfma = (double)RT * (K / 16) * 3 * 32;       // MFMA pipe cycles per wave (32 cycles per 32x32x16 instruction)
#ifdef PD_STRIP_STEP_CLOCKS
    {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int b = 0; b < wgs; ++b) { s0 += (double)(L[b * 8] >> 32); s1 += (double)(L[b * 8] & 0xffffffffll); s2 += (double)(L[b * 8 + 1] >> 32); s3 += (double)(L[b * 8 + 1] & 0xffffffffll); }
        const int chunks = K / 64;
        printf("%-28s RT=%d: cycles per 16-k step of %d MFMAs (wave 0, mean over workgroups and chunks; incl. ~ 50 of the two s_memtime): step 0 (with DMA pieces) %5.0f | step 1 (with DMA pieces) %5.0f | step 2 %5.0f | step 3 %5.0f\n",
               name, RT, 3 * RT, s0 / wgs / chunks, s1 / wgs / chunks, s2 / wgs / chunks, s3 / wgs / chunks);
        return;
    }
#endif
    printf("%-28s RT=%d %4d workgroups: %6.2f us per launch (hipEvents, 20 back to back); stamped launch %6.2f us wall; workgroup starts: median %5.2f, p90 %5.2f, max %5.2f us, "
           "%3.0f start > 2 us; ends: p10 %5.2f median %5.2f max %5.2f us\n", name, RT, wgs, ms * 1e3 / 20, (w1 - w0) * 0.01, starts[wgs / 2], starts[wgs * 9 / 10], starts.back(),
           late, ends[wgs / 10], ends[wgs / 2], ends.back());
    printf("    per workgroup (mean shader cycles): first operands landed %6.0f | K loop %6.0f (MFMA pipe alone: %5.0f = %4.1f %%) | epilogue + store drain %6.0f | total %6.0f; "
           "inside the K loop, wave 0: wait for the 2nd block's weights %5.0f | wait for the next rows + 1st block %5.0f | barrier %5.0f\n", pro / wgs, loop / wgs, mfma, 100.0 * mfma / (loop / wgs), epi / wgs,
           dur / wgs, wb / wgs, wa / wgs, bar / wgs);
}

int main() {
    const int M = 5120;
    unsigned *A;
    uint4 *W;
    float *bias, *C;
    long long *legs;
    hipStream_t s;
    (void)hipStreamCreate(&s);
    (void)hipMalloc(&A, (size_t)M * 1024 * 4);
    (void)hipMalloc(&W, (size_t)1024 * 1024 * 4);
    (void)hipMalloc(&bias, 2048 * 4);
    (void)hipMalloc(&C, (size_t)M * 1024 * 4);
    (void)hipMalloc(&legs, (size_t)4096 * 8 * 8);
    hipLaunchKernelGGL(fill_words, dim3(1024), dim3(256), 0, s, A, (size_t)M * 1024, 1u);
    hipLaunchKernelGGL(fill_planes, dim3(1024), dim3(256), 0, s, W, (size_t)1024 * 1024 / 4, 7u);
    (void)hipMemsetAsync(bias, 0, 2048 * 4, s);
    (void)hipMemsetAsync(C, 0, (size_t)M * 1024 * 4, s);
    (void)hipStreamSynchronize(s);
    for (int rep = 0; rep < 2; ++rep) {
        run<2, 2>("out-projection 512 <- 512", M, 512, 512, A, W, bias, C, legs, s);
        run<2, 3>("out-projection 512 <- 512", M, 512, 512, A, W, bias, C, legs, s);
        run<2, 2>("FF2 512 <- 1024", M, 512, 1024, A, W, bias, C, legs, s);
        run<2, 3>("FF2 512 <- 1024", M, 512, 1024, A, W, bias, C, legs, s);
        run<4, 2>("FF1 1024 <- 512 (ReLU)", M, 1024, 512, A, W, bias, C, legs, s);
        run<4, 3>("FF1 1024 <- 512 (ReLU)", M, 1024, 512, A, W, bias, C, legs, s);
    }
    return 0;
}
