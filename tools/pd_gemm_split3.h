// pd_gemm_split3.h (tools/: a measured alternative, not part of libpd_engine.so; used by split3_probe.hip only) -- fp32-grade GEMM on the bf16 matrix pipe: every fp32 operand is carried as THREE bf16 values
// (x = hi + mid + lo exactly to 24 mantissa bits: each of the two differences below is exact in fp32), and
//     x * w ~= hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid)
// -- the six products whose weight is >= 2^-16 of the leading one; the three dropped ones (mid*lo, lo*mid, lo*lo) are <= 2^-24 of it,
// the size of the rounding of an fp32 product.  Six v_mfma_f32_32x32x16_bf16 (each 16 x the rate of v_mfma_f32_32x32x2_f32 per
// product) with fp32 accumulation: 16 / 6 = 2.7 x the exact-fp32 matrix rate at fp32 accuracy (tests/perf/denoiser_precision_study.py,
// mode "split3"; measured against the exact kernels in tests/test_gpu_parity_r3.py).
//
// Activations stay fp32 in memory (the kernels around the GEMMs are the exact mode's): the A rows are split while they are staged
// into LDS (LayerNorm applied first where the GEMM follows one); the weights are split once, at engine creation, into MFMA fragment
// order and go from L2 straight to registers one k-chunk ahead.
#pragma once
#include "../posediffusion_amd/csrc/pd_gemm_stream.h"

#define PD_S3_LR 52      // LDS row stride in 32-bit words: 4 groups of 8 k x (hi | mid | lo) x 4 words, + 4: fragment reads conflict free

struct PdBf16x3 {
    __bf16 h, m, l;
};
__device__ __forceinline__ PdBf16x3 pd_split3(float v) {
    PdBf16x3 r;
    r.h = (__bf16)v;
    const float r1 = v - (float)r.h;       // exact: <= 16 significant bits
    r.m = (__bf16)r1;
    r.l = (__bf16)(r1 - (float)r.m);       // the difference is exact again
    return r;
}
__device__ __forceinline__ unsigned pd_bf16_pair(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
// 8 consecutive k of one row -> the three 16-byte fragment pieces
__device__ __forceinline__ void pd_split3x8(const float (&v)[8], uint4 &h, uint4 &m, uint4 &l) {
    PdBf16x3 s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = pd_split3(v[e]);
    h = make_uint4(pd_bf16_pair(s[0].h, s[1].h), pd_bf16_pair(s[2].h, s[3].h), pd_bf16_pair(s[4].h, s[5].h), pd_bf16_pair(s[6].h, s[7].h));
    m = make_uint4(pd_bf16_pair(s[0].m, s[1].m), pd_bf16_pair(s[2].m, s[3].m), pd_bf16_pair(s[4].m, s[5].m), pd_bf16_pair(s[6].m, s[7].m));
    l = make_uint4(pd_bf16_pair(s[0].l, s[1].l), pd_bf16_pair(s[2].l, s[3].l), pd_bf16_pair(s[4].l, s[5].l), pd_bf16_pair(s[6].l, s[7].l));
}

// W[n][k] (fp32, row-major, LayerNorm scale already folded where one applies) -> [n / 32][k / 16][hi | mid | lo][lane] x 16 B,
// lane = (n % 32) + 32 * ((k / 8) % 2), 8 consecutive k per lane: one wave-wide 16-byte load is 1 KB contiguous
static __global__ void pd_frag_split3_kernel(const float *__restrict__ W, int K, size_t total, uint4 *__restrict__ out) {
    const int KS = K / 16;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t t = idx >> 6;
        const int ks = (int)(t % KS), nt = (int)(t / KS);
        const int n = nt * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = W[(size_t)n * K + k0 + e];
        uint4 h, m, l;
        pd_split3x8(v, h, m, l);
        out[(t * 3 + 0) * 64 + lane] = h;
        out[(t * 3 + 1) * 64 + lane] = m;
        out[(t * 3 + 2) * 64 + lane] = l;
    }
}
static inline size_t pd_split3_weight_bytes(int Nout, int K) { return (size_t)(Nout / 32) * (K / 16) * 3 * 64 * sizeof(uint4); }

struct PdSplit3Args {
    const float *A;          // fp32 [M][lda]
    const uint4 *W;          // pd_frag_split3_kernel's order
    const float *bias;
    float *C;                // fp32 [M][Nout]
    int M, Nout, K, lda;
    const float2 *ln_stats;  // ALN: (mean, 1 / sqrt(var + eps)) per row of A
};

// EPI as pd_gemm_dma_kernel: 0 bias, 1 bias + relu, 2 bias + residual (C += ...), 3 bias + gelu.
// One workgroup of 4 waves per (64 WM) x (64 WN) tile, a (32 WM) x (32 WN) quadrant per wave, k in chunks of 32 (two MFMA steps).
template <int EPI, bool ALN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void pd_gemm_split3_kernel(PdSplit3Args g) {
    constexpr int KC = 32, LR = PD_S3_LR, TM = 64 * WM, TN = 64 * WN, GROUP = 2048 / TM;
    extern __shared__ __attribute__((aligned(16))) unsigned s3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave & 1, wn = wave >> 1;
    const int MT = (g.M + TM - 1) / TM, NT = g.Nout / TN;
    int mtile, ntile;
    {
        const int b = blockIdx.x, full = (MT / GROUP) * GROUP * NT;
        if (b < full) {
            const int grp = b / (NT * GROUP), r = b - grp * (NT * GROUP);
            ntile = r / GROUP;
            mtile = grp * GROUP + r % GROUP;
        } else {
            const int r = b - full, rest = MT % GROUP;
            ntile = r / rest;
            mtile = (MT / GROUP) * GROUP + r % rest;
        }
    }
    const int m0 = mtile * TM, n0 = ntile * TN;
    // staging: 4 threads per row and chunk, 8 k each; passes of 64 rows
    const int sr = tid >> 2, sg = tid & 3;
    const float *ap[WM];
    float ln_mu[WM], ln_rs[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) {
        const int row = min(m0 + sr + 64 * j, g.M - 1);
        ap[j] = g.A + (size_t)row * g.lda + 8 * sg;
        if constexpr (ALN) {
            const float2 st2 = g.ln_stats[row];
            ln_mu[j] = st2.x;
            ln_rs[j] = st2.y;
        }
    }
    const int KS = g.K / 16;
    const uint4 *wq[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) wq[j] = g.W + (size_t)(n0 / 32 + wn * WN + j) * KS * 192 + lane;
    float4 ra[WM][2];
    uint4 cw[WN][2][3], nw[WN][2][3];
    auto load = [&](int nc) {
#pragma unroll
        for (int j = 0; j < WM; ++j) {
            ra[j][0] = *(const float4 *)(ap[j] + nc * KC);
            ra[j][1] = *(const float4 *)(ap[j] + nc * KC + 4);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) nw[j][s][p] = wq[j][(size_t)((nc * 2 + s) * 3 + p) * 64];
    };
    auto store = [&](unsigned *da) {
#pragma unroll
        for (int j = 0; j < WM; ++j) {
            float v[8] = {ra[j][0].x, ra[j][0].y, ra[j][0].z, ra[j][0].w, ra[j][1].x, ra[j][1].y, ra[j][1].z, ra[j][1].w};
            if constexpr (ALN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] - ln_mu[j]) * ln_rs[j];
            }
            uint4 h, m, l;
            pd_split3x8(v, h, m, l);
            unsigned *d = da + (sr + 64 * j) * LR + 12 * sg;
            *(uint4 *)d = h;
            *(uint4 *)(d + 4) = m;
            *(uint4 *)(d + 8) = l;
        }
    };
    auto roll = [&]() {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) cw[j][s][p] = nw[j][s][p];
    };
    load(0);
    store(s3_lds);
    roll();
    __syncthreads();
    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.0f;
    const int nk = g.K / KC;
    const int aoff = (wm * 32 * WM + l31) * LR + 12 * hi;        // lanes 0-31 take group 2 s, lanes 32-63 group 2 s + 1
    for (int kc = 0; kc < nk; ++kc) {
        load(min(kc + 1, nk - 1));                               // the chunk after the last is the last again
        __builtin_amdgcn_sched_barrier(0);
        const unsigned *a = s3_lds + (kc & 1) * TM * LR + aoff;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[WM][3];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[mi][p] = __builtin_bit_cast(bf16x8, *(const uint4 *)(a + mi * 32 * LR + 24 * s + 4 * p));
            // smallest terms first; consecutive MFMAs go to different accumulators
#define PD_S3_MMA(PA, PB)                                                                                                              \
    _Pragma("unroll") for (int mi = 0; mi < WM; ++mi) _Pragma("unroll") for (int ni = 0; ni < WN; ++ni)                                 \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][PA], __builtin_bit_cast(bf16x8, cw[ni][s][PB]), acc[mi][ni], 0, 0, 0);
            PD_S3_MMA(2, 0)
            PD_S3_MMA(0, 2)
            PD_S3_MMA(1, 1)
            PD_S3_MMA(1, 0)
            PD_S3_MMA(0, 1)
            PD_S3_MMA(0, 0)
#undef PD_S3_MMA
        }
        __builtin_amdgcn_sched_barrier(0);
        store(s3_lds + ((kc + 1) & 1) * TM * LR);
        roll();
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int col = n0 + (wn * WN + ni) * 32 + l31, r0 = m0 + (wm * WM + mi) * 32 + 4 * hi;
            const float bias = g.bias[col];
            float res[16];
            if constexpr (EPI == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) res[i] = g.C[(size_t)min(r0 + (i & 3) + 8 * (i >> 2), g.M - 1) * g.Nout + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = r0 + (i & 3) + 8 * (i >> 2);
                float v = acc[mi][ni][i] + bias;
                if constexpr (EPI == 1) v = fmaxf(v, 0.0f);
                if constexpr (EPI == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                if constexpr (EPI == 2) v += res[i];
                if (row < g.M) g.C[(size_t)row * g.Nout + col] = v;
            }
        }
}

static constexpr size_t pd_split3_lds(int WM) { return (size_t)2 * 64 * WM * PD_S3_LR * sizeof(unsigned); }
template <int EPI, bool ALN = false, int WM = 2, int WN = 2>
static inline void pd_gemm_split3(const float *A, int lda, const uint4 *W, int K, const float *bias, float *C, int M, int Nout, hipStream_t s,
                                  const float2 *ln_stats = nullptr) {
    PdSplit3Args g{A, W, bias, C, M, Nout, K, lda, ln_stats};
    constexpr int TM = 64 * WM, TN = 64 * WN;
    hipLaunchKernelGGL((pd_gemm_split3_kernel<EPI, ALN, WM, WN>), dim3(((M + TM - 1) / TM) * (Nout / TN)), dim3(256), pd_split3_lds(WM), s, g);
}
