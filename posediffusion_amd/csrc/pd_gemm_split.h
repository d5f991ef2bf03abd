// pd_gemm_split.h -- split-precision GEMM on the bf16 matrix pipe, shared by the image feature extractor (pd_vit.hip) and the
// denoiser's fast mode (pd_denoiser.hip): every fp32 operand is carried as bf16 hi + bf16 lo (x ~= hi + lo, 16 mantissa bits),
// x * w ~= hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (16 x the rate of the f32 instruction per product, three
// products), fp32 accumulation.  Activations between kernels are one 32-bit word per element {hi | lo << 16}
// (pd_split_word, pd_gemm_stream.h).
//
// F16 (the denoiser's fp16-plane mode): the same kernel with fp16 halves (pd_split_word_h) on v_mfma_f32_32x32x16_f16 -- 11 + 11
// mantissa bits, the dropped lo*lo term is 2^-22 of the product: with fp32 accumulation the result is as close to the fp64 product
// as the exact-fp32 kernel's (tools/split3_probe.hip).  fp16's narrow exponent range is handled by the caller with POWER-OF-TWO
// scales fixed at engine creation from provable bounds on every operand (pd_denoiser_build_split): A arrives as words of
// a * 2^ea, W was split as w * 2^ew, the epilogue multiplies the accumulator by c_scale = 2^-(ea + ew) (exact) and, where it
// writes split words itself (EPI 3 / 4), by out_scale = 2^e of the next GEMM's operand.
#pragma once
#include "pd_gemm_stream.h"

// GELU for the split-precision path: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below the 2^-17 the split operands
// keep), a dozen instructions instead of erff's ~35 -- the epilogue of fc1 is as long as its matrix loop otherwise.
// 0.5 v (1 + erf(v / sqrt 2)) = v (1 - q / 2) for v >= 0 and v q / 2 for v < 0, q = erfc(|v| / sqrt 2) (no cancellation).
__device__ __forceinline__ float vit_gelu_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = 0.5f * p * t * __expf(-x * x);
    return v * (v >= 0.0f ? 1.0f - q : q);
}

// W[n][k] * gamma[k] -> split and packed in MFMA fragment order: [n / 32][k / 16][hi | lo][lane] x 16 B, lane = (n % 32) +
// 32 * ((k / 8) % 2), 8 consecutive k per lane: one wave-wide 16-byte load is 1 KB contiguous
// f16: fp16 halves of w * scale instead of bf16 halves of w
static __global__ void vit_frag_split_kernel(const float *__restrict__ W, const float *__restrict__ gamma, int K, size_t total, uint4 *__restrict__ out,
                                             int f16 = 0, float scale = 1.0f) {
    const int KS = K / 16;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const size_t t = idx >> 6;
        const int ks = (int)(t % KS), nt = (int)(t / KS);
        const int n = nt * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
        unsigned w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = W[(size_t)n * K + k0 + e];
            w[e] = f16 ? pd_split_word_h(v * scale) : pd_split_word(gamma ? v * gamma[k0 + e] : v);
        }
        uint4 hi, lo;
        hi.x = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u); lo.x = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u);
        hi.y = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u); lo.y = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
        hi.z = __builtin_amdgcn_perm(w[5], w[4], 0x05040100u); lo.z = __builtin_amdgcn_perm(w[5], w[4], 0x07060302u);
        hi.w = __builtin_amdgcn_perm(w[7], w[6], 0x05040100u); lo.w = __builtin_amdgcn_perm(w[7], w[6], 0x07060302u);
        out[(t * 2) * 64 + lane] = hi;
        out[(t * 2 + 1) * 64 + lane] = lo;
    }
}

struct VitSplitArgs {
    const unsigned *A, *W;      // A: split words [M][lda]; W: vit_frag_split_kernel's fragment order
    const float *bias;
    void *C;                    // EPI 0 / 2: fp32 [M][Nout]; EPI 3 (gelu) / 4 (relu): split words [M][Nout]
    int M, Nout, K, lda;
    float c_scale, out_scale;   // F16 only (see the header): accumulator scale, scale of split-word outputs
#ifdef PD_STRIP_LEGS
    long long *legs;            // tools/strip_legs_probe.hip only: [workgroup][8] = {wall start, wall end, cycles: entry, operands landed, K loop done, end, XCC id, CU id}
#endif
};
#ifndef PD_STRIP_PIPE
#define PD_STRIP_PIPE 1
#endif
#ifndef PD_STRIP_RES_AHEAD
#define PD_STRIP_RES_AHEAD 1   // the residual tile of an EPI 2 product requested during the last K chunk (needs the wide epilogue and the 64-k form)
#endif
#ifndef PD_STRIP_WIDE_EPI
#define PD_STRIP_WIDE_EPI 1
#endif
#ifdef PD_STRIP_LEGS
#define PD_LEG(i, v) do { if (g.legs && threadIdx.x == 0) g.legs[(size_t)blockIdx.x * 8 + (i)] = (long long)(v); } while (0)
#else
#define PD_LEG(i, v) do { } while (0)
#endif

// A rows stream through LDS (un-zipped into hi / lo fragments on the way, shared by the waves of a row block); the weight
// fragments go straight from global memory / L2 to registers, one chunk ahead (they are already in operand order, and
// keeping them out of LDS halves its traffic -- the LDS array, not the matrix pipe, limited the first version).
// BARE (tools/split3_probe.hip only): 1 no weight-fragment loads after the first chunk, 2 no A loads / LDS stores after it, 3 both,
// 4 all of that and no barrier, 5 no MFMAs (everything else as in production) -- what bounds the kernel
template <int EPI, int WM, int WN, bool F16 = false, int BARE = 0>
__global__ __launch_bounds__(256) void vit_gemm_split_kernel(VitSplitArgs g) {
    constexpr int KC = PD_STREAM_KC, LR = PD_STREAM_LR, TM = 64 * WM, TN = 64 * WN, GROUP = 2048 / TM;
    static_assert(KC == 32 && WM <= 2 && WN <= 2, "staging: 4 groups of 8 per row chunk, passes of 64 rows");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned *As = (unsigned *)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5, wm = wave & 1, wn = wave >> 1;
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
    const int sr = tid >> 2, sg = tid & 3, st = sr * LR + 8 * sg;      // 4 threads per row chunk, one group of 8 each
    const int KS = g.K / 16;
#define VP_EACH(X) X(0) X(1)
#define VP_DECL(j)                                                                                                            \
    const uint4 *ap##j = (const uint4 *)(g.A + (size_t)min(m0 + sr + 64 * (j < WM ? j : 0), g.M - 1) * g.lda) + 2 * sg;        \
    const uint4 *wq##j = (const uint4 *)g.W + (size_t)(n0 / 32 + wn * WN + (j < WN ? j : 0)) * KS * 128 + lane;                 \
    uint4 ra##j##a, ra##j##b, cw##j##0h, cw##j##0l, cw##j##1h, cw##j##1l, nw##j##0h, nw##j##0l, nw##j##1h, nw##j##1l;
#define VP_LOAD(j)                                  \
    if constexpr (j < WM) {                         \
        ra##j##a = ap##j[nx];                       \
        ra##j##b = ap##j[nx + 1];                   \
    }                                               \
    if constexpr (j < WN) {                         \
        nw##j##0h = wq##j[(size_t)(nc * 4 + 0) * 64]; \
        nw##j##0l = wq##j[(size_t)(nc * 4 + 1) * 64]; \
        nw##j##1h = wq##j[(size_t)(nc * 4 + 2) * 64]; \
        nw##j##1l = wq##j[(size_t)(nc * 4 + 3) * 64]; \
    }
#define VP_LOAD_A(j)                                \
    if constexpr (j < WM) {                         \
        ra##j##a = ap##j[nx];                       \
        ra##j##b = ap##j[nx + 1];                   \
    }
#define VP_LOAD_W(j)                                \
    if constexpr (j < WN) {                         \
        nw##j##0h = wq##j[(size_t)(nc * 4 + 0) * 64]; \
        nw##j##0l = wq##j[(size_t)(nc * 4 + 1) * 64]; \
        nw##j##1h = wq##j[(size_t)(nc * 4 + 2) * 64]; \
        nw##j##1l = wq##j[(size_t)(nc * 4 + 3) * 64]; \
    }
#define VP_ROLL(j)             \
    if constexpr (j < WN) {    \
        cw##j##0h = nw##j##0h; \
        cw##j##0l = nw##j##0l; \
        cw##j##1h = nw##j##1h; \
        cw##j##1l = nw##j##1l; \
    }
#define VP_STORE(j)                                                                                       \
    if constexpr (j < WM) {                                                                               \
        uint4 h, l;                                                                                       \
        h.x = __builtin_amdgcn_perm(ra##j##a.y, ra##j##a.x, 0x05040100u);                                 \
        l.x = __builtin_amdgcn_perm(ra##j##a.y, ra##j##a.x, 0x07060302u);                                 \
        h.y = __builtin_amdgcn_perm(ra##j##a.w, ra##j##a.z, 0x05040100u);                                 \
        l.y = __builtin_amdgcn_perm(ra##j##a.w, ra##j##a.z, 0x07060302u);                                 \
        h.z = __builtin_amdgcn_perm(ra##j##b.y, ra##j##b.x, 0x05040100u);                                 \
        l.z = __builtin_amdgcn_perm(ra##j##b.y, ra##j##b.x, 0x07060302u);                                 \
        h.w = __builtin_amdgcn_perm(ra##j##b.w, ra##j##b.z, 0x05040100u);                                 \
        l.w = __builtin_amdgcn_perm(ra##j##b.w, ra##j##b.z, 0x07060302u);                                 \
        *(uint4 *)(da + st + j * 64 * LR) = h;                                                            \
        *(uint4 *)(da + st + j * 64 * LR + 4) = l;                                                        \
    }
// the three products of one (column tile j, k step s) against every row tile
#define VP_MMA(j, s)                                                                                                         \
    if constexpr (j < WN) {                                                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < WM; ++mi) {                                                                  \
            acc[mi][j < WN ? j : 0] = mma(al##s[mi], cw##j##s##h, acc[mi][j < WN ? j : 0]);                                    \
            acc[mi][j < WN ? j : 0] = mma(ah##s[mi], cw##j##s##l, acc[mi][j < WN ? j : 0]);                                    \
            acc[mi][j < WN ? j : 0] = mma(ah##s[mi], cw##j##s##h, acc[mi][j < WN ? j : 0]);                                    \
        }                                                                                                                    \
    }
    auto mma = [](const uint4 &a, const uint4 &b, const f32x16 &c) {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    VP_EACH(VP_DECL)
    {
        const int nx = 0, nc = 0;
        unsigned *da = As;
        VP_EACH(VP_LOAD)
        VP_EACH(VP_STORE)
        VP_EACH(VP_ROLL)
    }
    __syncthreads();
    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.0f;
    const int nk = g.K / KC;
    const int aoff = (wm * 32 * WM + l31) * LR + 8 * hi;
    for (int kc = 0; kc < nk; ++kc) {
        const int nc = min(kc + 1, nk - 1), nx = nc * (KC / 4);       // the chunk after the last is the last again
        if constexpr (BARE == 0 || BARE == 5) { VP_EACH(VP_LOAD) }
        else if constexpr (BARE == 1) { VP_EACH(VP_LOAD_A) }
        else if constexpr (BARE == 2) { VP_EACH(VP_LOAD_W) }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned *a = As + (kc & 1) * TM * LR + aoff;
        uint4 ah0[WM], al0[WM], ah1[WM], al1[WM];     // 16 k per step: lanes 0-31 take group 2 s, lanes 32-63 group 2 s + 1
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
            ah0[mi] = *(const uint4 *)(a + mi * 32 * LR);
            al0[mi] = *(const uint4 *)(a + mi * 32 * LR + 4);
            ah1[mi] = *(const uint4 *)(a + mi * 32 * LR + 16);
            al1[mi] = *(const uint4 *)(a + mi * 32 * LR + 20);
        }
        if constexpr (BARE != 5) {
            VP_MMA(0, 0)
            VP_MMA(1, 0)
            VP_MMA(0, 1)
            VP_MMA(1, 1)
        }
        __builtin_amdgcn_sched_barrier(0);
        unsigned *da = As + ((kc + 1) & 1) * TM * LR;
        if constexpr (BARE == 0 || BARE == 1 || BARE == 5) { VP_EACH(VP_STORE) }
        if constexpr (BARE == 0 || BARE == 2 || BARE == 5) { VP_EACH(VP_ROLL) }
        if constexpr (BARE != 4) __syncthreads();
    }
#undef VP_DECL
#undef VP_LOAD
#undef VP_LOAD_A
#undef VP_LOAD_W
#undef VP_ROLL
#undef VP_STORE
#undef VP_MMA
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int col = n0 + (wn * WN + ni) * 32 + l31, r0 = m0 + (wm * WM + mi) * 32 + 4 * hi;
            const float bias = g.bias[col];
            float res[16];
            if constexpr (EPI == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) res[i] = ((const float *)g.C)[(size_t)min(r0 + (i & 3) + 8 * (i >> 2), g.M - 1) * g.Nout + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = r0 + (i & 3) + 8 * (i >> 2);
                float v = F16 ? fmaf(acc[mi][ni][i], g.c_scale, bias) : acc[mi][ni][i] + bias;
                if constexpr (EPI == 3) v = vit_gelu_fast(v);
                if constexpr (EPI == 4) v = pd_relu(v);
                if constexpr (EPI == 2) v += res[i];
                if (row < g.M) {
                    if constexpr (EPI == 3 || EPI == 4)
                        ((unsigned *)g.C)[(size_t)row * g.Nout + col] = pd_split_word_as<F16 ? 2 : 1>(v, g.out_scale);
                    else
                        ((float *)g.C)[(size_t)row * g.Nout + col] = v;
                }
            }
        }
}

// ---- the same arithmetic, operands delivered differently (round 3) ------------------------------------------------------------------
// vit_gemm_split_kernel is bound by operand delivery, not by the matrix pipe (tools/split3_probe.hip, BARE): both waves of a row block
// fetch the same weight fragments from L2 (104 B/clk/CU asked of a 64 B/clk port at the 64 x 128 tile) and the A rows make a
// global -> VGPR -> v_perm -> ds_write round trip.  Here a workgroup's tile is (32 RT) rows x (128 CT) columns and every wave owns a
// STRIP of 32 CT columns of it over all RT row tiles: no weight fragment is fetched twice (they still go L2 -> registers, one chunk ahead), and
// the A rows -- raw split words -- go L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, hand-issued as in pd_gemm_dma_kernel, 16-byte slots
// XOR-swizzled by (row >> 1) & 7) and are un-zipped into hi / lo fragments when they are read.  The products of an output element are
// accumulated in the same order as in vit_gemm_split_kernel: bitwise the same C.
// BARE (probe only): 1 no weight-fragment loads in the loop, 2 no DMA in the loop, 3 neither, 4 neither and no barrier / wait,
// 5 everything but the MFMAs, 6 everything but the fragment reads + un-zip (the first chunk's are reused)
// K64 (round 4): A chunks of 64 k per barrier (two 32-k blocks of the same LDS layout), the weight fragments of a block requested half a chunk
// ahead into the registers the previous block just freed: half the barriers, every load has >= half a chunk of matrix work to hide behind,
// the same accumulation order (bitwise the same C).  K must be a multiple of 64.
template <int EPI, int RT, bool F16, int CT = 1, int BARE = 0, bool K64 = false>
__global__ __launch_bounds__(256) void pd_gemm_strip_kernel(VitSplitArgs g) {
    constexpr int KC = 32, TM = 32 * RT, TN = 128 * CT, GROUP = RT == 3 ? 24 : (RT == 1 ? 32 : 2048 / TM), CHA = TM * KC;     // words of A per chunk; GROUP row tiles (a multiple of
                                                                                                         //   8: a row block's column tiles share an XCD) per block group
    static_assert(RT >= 1 && RT <= 4 && (CT == 1 || CT == 2), "pieces of 8 rows, RT per wave and chunk");
    extern __shared__ __attribute__((aligned(1024))) unsigned strip_lds[];
    PD_LEG(0, wall_clock64());
    PD_LEG(2, __builtin_amdgcn_s_memtime());
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // staging: pieces of 1 KiB = 8 rows x 32 words; a chunk has 4 RT of them, wave w moves pieces [RT w, RT (w + 1))
    const int prow = lane >> 3, pslot = lane & 7;
    unsigned oa[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        const int r = 8 * (RT * wave + j) + prow;
        oa[j] = (unsigned)(((size_t)min(m0 + r, g.M - 1) * g.lda + 4 * (pslot ^ ((r >> 1) & 7))) * sizeof(unsigned));
    }
    const unsigned lds_a = (unsigned)(size_t)(strip_lds + RT * wave * 256);
    auto stage = [&](int kc, int buf) {
        const float *ab = (const float *)(g.A + kc * KC);
        const unsigned da = __builtin_amdgcn_readfirstlane(lds_a + buf * CHA * 4);
#pragma unroll
        for (int j = 0; j < RT; ++j) pd_dma_piece(ab, oa[j], da + j * 1024);
    };
    const int KS = g.K / 16;
    const uint4 *wq = (const uint4 *)g.W + (size_t)(n0 / 32 + CT * wave) * KS * 128 + lane;      // this wave's first column tile; the next one KS * 128 further
    uint4 cw[CT][2][2], nw[CT][2][2];               // [column tile][k step][hi | lo]
    auto mma = [](const uint4 &a, const uint4 &b, const f32x16 &c) {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    };
    f32x16 acc[RT][CT];
    typedef unsigned res_v4 __attribute__((ext_vector_type(4)));
    res_v4 resid[(EPI == 2 && K64) ? RT : 1][4];      // (PD_STRIP_RES_AHEAD, below)
    if constexpr (K64) {
        static_assert(CT == 1 && BARE == 0, "the 64-k form exists for one column tile per wave");
        typedef unsigned wv4 __attribute__((ext_vector_type(4)));          // a weight fragment as a native vector (an inline-asm register operand)
#pragma unroll
        for (int mi = 0; mi < RT; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][0][i] = 0.0f;
        auto mmaw = [](const uint4 &a, const wv4 &b, const f32x16 &c) {
            if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
        };
        // weights of 32-k block b (b = 2 * chunk + half): 4 fragments {k step 0 | 1} x {hi | lo}, 1 KiB apart -- hand-issued like the DMA, so that
        // every vector-memory operation of the loop is counted by the s_waitcnt below and by nothing else (the compiler's own waits assume
        // it sees every load in flight)
#define PD_STRIP_WLOAD(w0, w1, w2, w3, b)                                                                                            \
    do {                                                                                                                             \
        const uint4 *wp_ = wq + (size_t)(b) * 256;                                                                                   \
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"                           \
                     "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"                    \
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(wp_) : "memory");                                            \
    } while (0)
#define PD_STRIP_WAIT(n, w0, w1, w2, w3) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : : "memory")
#define PD_STRIP_WAITN(n, w0, w1, w2, w3) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "n"(n) : "memory")
        auto stage64 = [&](int c, int buf) {            // both 32-k blocks of chunk c: 2 RT pieces per wave
            const unsigned da = __builtin_amdgcn_readfirstlane(lds_a + buf * 2 * CHA * 4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *ab = (const float *)(g.A + (2 * c + h) * KC);
#pragma unroll
                for (int j = 0; j < RT; ++j) pd_dma_piece(ab, oa[j], da + h * CHA * 4 + j * 1024);
            }
        };
        // one 32-k block: fragment reads + un-zip + 6 RT MFMAs; w0 / w1 = k step 0 {hi | lo}, w2 / w3 = k step 1
        auto block = [&](const unsigned *a, const wv4 &w0, const wv4 &w1, const wv4 &w2, const wv4 &w3) {
            uint4 ah[RT], al[RT];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int mi = 0; mi < RT; ++mi) {
                    const uint4 p = *(const uint4 *)(a + mi * 32 * KC + 4 * ((4 * st + 2 * hi) ^ ((l31 >> 1) & 7)));
                    const uint4 q = *(const uint4 *)(a + mi * 32 * KC + 4 * ((4 * st + 2 * hi + 1) ^ ((l31 >> 1) & 7)));
                    ah[mi] = make_uint4(__builtin_amdgcn_perm(p.y, p.x, 0x05040100u), __builtin_amdgcn_perm(p.w, p.z, 0x05040100u),
                                        __builtin_amdgcn_perm(q.y, q.x, 0x05040100u), __builtin_amdgcn_perm(q.w, q.z, 0x05040100u));
                    al[mi] = make_uint4(__builtin_amdgcn_perm(p.y, p.x, 0x07060302u), __builtin_amdgcn_perm(p.w, p.z, 0x07060302u),
                                        __builtin_amdgcn_perm(q.y, q.x, 0x07060302u), __builtin_amdgcn_perm(q.w, q.z, 0x07060302u));
                }
                const wv4 &wh = st ? w2 : w0, &wl = st ? w3 : w1;
#pragma unroll
                for (int mi = 0; mi < RT; ++mi) acc[mi][0] = mmaw(al[mi], wh, acc[mi][0]);
#pragma unroll
                for (int mi = 0; mi < RT; ++mi) acc[mi][0] = mmaw(ah[mi], wl, acc[mi][0]);
#pragma unroll
                for (int mi = 0; mi < RT; ++mi) acc[mi][0] = mmaw(ah[mi], wh, acc[mi][0]);
            }
        };
#if PD_STRIP_PIPE
        // Round 6: the 16-k steps of the K loop, scheduled by hand.  A wave issues in order and an MFMA holds its issue port for the matrix pipe's 32 cycles unless
        // something else is there to issue: the compiler's schedule -- a step's 24 un-zips and 6 fragment reads in clumps, its 9 MFMAs back to back -- ran
        // 36 MFMAs + 234 other instructions per 64-k chunk in ~ 2 100 cycles = their SUM (tools/strip_legs_probe.hip: waits + barrier are 5 % of the loop, yet the
        // matrix pipe was 41 - 55 % busy; neither later operands, nor hidden LDS latency, nor moved DMA issue changed that).  Here every MFMA is preceded by the
        // few instructions that fit its shadow: the four un-zips of its own A fragment, or (second product) four un-zips + the NEXT step's two fragment reads of
        // that row tile, or (third product) one LDS-DMA piece of the next chunk; sched_barrier(0) pins the order.  Raw fragments double-buffered in registers, read
        // one step ahead and waited for (lgkmcnt(0)) a step later.  The same un-zips and MFMAs per accumulator in the same order as `block`: bitwise the same C.
        wv4 raw[2][RT][2];
        const int swz = (l31 >> 1) & 7;
        auto read_mi = [&](int rb, const unsigned *a, int st, int mi) {
            const unsigned p_ = (unsigned)(size_t)(a + mi * 32 * KC + 4 * ((4 * st + 2 * hi) ^ swz));
            const unsigned q_ = (unsigned)(size_t)(a + mi * 32 * KC + 4 * ((4 * st + 2 * hi + 1) ^ swz));
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(raw[rb][mi][0]), "=&v"(raw[rb][mi][1]) : "v"(p_), "v"(q_) : "memory");
        };
        auto wait_reads = [&](int rb) {
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[rb][mi][0]), "+v"(raw[rb][mi][1]) : : "memory");
        };
        // piece p (0 .. 2 RT - 1) of chunk dma_cn's rows -> buffer dma_nb (the vector-memory issue ORDER of the loop is unchanged -- all 2 RT pieces before the
        // first block's weight loads --, so the hand-counted vmcnt waits stand)
        int dma_cn = 0, dma_nb = 0;
        auto dma_piece_at = [&](int p) {
            const unsigned da = __builtin_amdgcn_readfirstlane(lds_a + dma_nb * 2 * CHA * 4);
            const int h = p / RT, j = p % RT;
            pd_dma_piece((const float *)(g.A + (2 * dma_cn + h) * KC), oa[j], da + h * CHA * 4 + j * 1024);
        };
        auto step = [&](int rb, const wv4 &wh, const wv4 &wl, bool has_next, int nrb, const unsigned *na, int nst, int dma_step) {
            uint4 ah[RT], al[RT];
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) {
                const wv4 p = raw[rb][mi][0], q = raw[rb][mi][1];
#ifdef PD_STRIP_NOPERM      // (probe only: what the loop costs WITHOUT the un-zip -- wrong products, right instruction count otherwise)
                al[mi] = make_uint4(q.x, q.y, q.z, q.w);
#else
                al[mi] = make_uint4(__builtin_amdgcn_perm(p.y, p.x, 0x07060302u), __builtin_amdgcn_perm(p.w, p.z, 0x07060302u),
                                    __builtin_amdgcn_perm(q.y, q.x, 0x07060302u), __builtin_amdgcn_perm(q.w, q.z, 0x07060302u));
#endif
                __builtin_amdgcn_sched_barrier(0);
                acc[mi][0] = mmaw(al[mi], wh, acc[mi][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) {
                const wv4 p = raw[rb][mi][0], q = raw[rb][mi][1];
#ifdef PD_STRIP_NOPERM
                ah[mi] = make_uint4(p.x, p.y, p.z, p.w);
#else
                ah[mi] = make_uint4(__builtin_amdgcn_perm(p.y, p.x, 0x05040100u), __builtin_amdgcn_perm(p.w, p.z, 0x05040100u),
                                    __builtin_amdgcn_perm(q.y, q.x, 0x05040100u), __builtin_amdgcn_perm(q.w, q.z, 0x05040100u));
#endif
                if (has_next) read_mi(nrb, na, nst, mi);
                __builtin_amdgcn_sched_barrier(0);
                acc[mi][0] = mmaw(ah[mi], wl, acc[mi][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) {
                if (dma_step >= 0) dma_piece_at(dma_step * RT + mi);
                __builtin_amdgcn_sched_barrier(0);
                acc[mi][0] = mmaw(ah[mi], wh, acc[mi][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#endif
        wv4 a0, a1, a2, a3, b0, b1, b2, b3;               // weight fragments of the chunk's first / second 32-k block
        const int nk64 = g.K / 64;
        // Round 6: the residual tile (EPI 2: C += ...) requested at the end of the chunk BEFORE the last, in the wide epilogue's own 16-byte pattern, instead of one
        // row tile at a time inside the epilogue (in-place C: the compiler cannot lift a row tile's loads over the previous tile's stores, so every row tile exposed
        // a whole global-load latency: ~ 1.9 k cycles each).  Hand-issued, counted by the loop's waits (`chunk`, below), 4 RT registers of 16 bytes.  The same values
        // into the same adds: bitwise the same C.
        auto resid_load = [&]() {
            const int pr_ = lane >> 3, colb_ = n0 + wave * 32 + (lane & 7) * 4;
#pragma unroll
            for (int mi = 0; mi < RT; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *rp_ = (const float *)g.C + (size_t)min(m0 + mi * 32 + 8 * q + pr_, g.M - 1) * g.Nout + colb_;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(resid[mi][q]) : "v"(rp_) : "memory");
                }
        };
        stage64(0, 0);
        PD_STRIP_WLOAD(a0, a1, a2, a3, 0);
        PD_STRIP_WLOAD(b0, b1, b2, b3, 1);
        PD_STRIP_WAIT(0, a0, a1, a2, a3);
        PD_STRIP_WAIT(0, b0, b1, b2, b3);
        __syncthreads();
        PD_LEG(3, __builtin_amdgcn_s_memtime());
#ifdef PD_STRIP_LEGS
        long long leg_t0 = 0, leg_wb = 0, leg_wa = 0, leg_bar = 0, leg_s0 = 0, leg_s1 = 0, leg_s2 = 0, leg_s3 = 0;
#define PD_LEG_T0() leg_t0 = __builtin_amdgcn_s_memtime()
#define PD_LEG_ACC(v) v += __builtin_amdgcn_s_memtime() - leg_t0
#else
#define PD_LEG_T0() do { } while (0)
#define PD_LEG_ACC(v) do { } while (0)
#endif
        // (round 6, measured and dropped: the A rows TWO chunks ahead through three LDS buffers -- the K loop's cycles did not move, 17.0 k -> 18.0 k per
        //  workgroup of the out-projection: what keeps the matrix pipe at 41 - 55 % is one wave per SIMD issuing its LDS reads, un-zips and waits between its
        //  own MFMAs, not late operands; profiles/round6_strip_legs.txt)
        // One 64-k chunk.  MODE 0: a chunk with a successor -- the successor's rows and weights are requested while this one is multiplied.  MODE 1: the one before the
        // last -- as 0, and (EPI 2) the residual tile is requested behind everything else, so that a whole chunk of matrix work hides it: returns are in order, the
        // end-of-chunk wait lets it fly (+ RESN) and so does the last chunk's first wait; the last chunk's end collects it.  MODE 2: the last chunk requests nothing
        // (round 6; until then it re-staged itself "for nothing": one chunk in nk64 of useless L2 traffic) and leaves the closing barrier to the epilogue.
        constexpr int RESN = (PD_STRIP_RES_AHEAD && EPI == 2) ? 4 * RT : 0;
        auto chunk = [&](int c, auto mode_) {
            constexpr int MODE = decltype(mode_)::value;
            const int cn = c + 1;
            const unsigned *a = strip_lds + (c & 1) * 2 * CHA + l31 * KC;
#if PD_STRIP_PIPE
            dma_cn = cn;                                         // in flight, oldest first: the second block's weights [4, from the previous turn], this DMA [2 RT]
            dma_nb = (c + 1) & 1;                                //   (its pieces are issued inside the first block's MFMA groups, below)
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) read_mi(0, a, 0, mi);                // (the one exposed read per chunk: its rows were published by the barrier just passed)
            wait_reads(0);
            PD_LEG_T0();
            step(0, a0, a1, true, 1, a, 1, MODE == 2 ? -1 : 0);
            PD_LEG_ACC(leg_s0);
            wait_reads(1);
            PD_LEG_T0();
            step(1, a2, a3, true, 0, a + CHA, 0, MODE == 2 ? -1 : 1);
            PD_LEG_ACC(leg_s1);
#else
            if constexpr (MODE != 2) stage64(cn, (c + 1) & 1);   // in flight, oldest first: the second block's weights [4, from the previous turn], this DMA [2 RT]
            block(a, a0, a1, a2, a3);
#endif
            if constexpr (MODE != 2) PD_STRIP_WLOAD(a0, a1, a2, a3, 2 * cn);     //   ... + the next chunk's first block [4]
            // the second block's weights must have landed; the DMA and the loads just issued may still be in flight (in-order returns)
            PD_LEG_T0();
            PD_STRIP_WAITN((MODE == 2 ? RESN : 2 * RT + 4), b0, b1, b2, b3);     // (2 RT DMA pieces + 4 weight loads may stay in flight; last chunk: the residual tile)
            PD_LEG_ACC(leg_wb);
#if PD_STRIP_PIPE
            wait_reads(0);
            PD_LEG_T0();
            step(0, b0, b1, true, 1, a + CHA, 1, -1);
            PD_LEG_ACC(leg_s2);
            wait_reads(1);
            PD_LEG_T0();
            step(1, b2, b3, false, 0, a, 0, -1);
            PD_LEG_ACC(leg_s3);
#else
            block(a + CHA, b0, b1, b2, b3);
#endif
            if constexpr (MODE != 2) {
                PD_STRIP_WLOAD(b0, b1, b2, b3, 2 * cn + 1);      //   ... + the next chunk's second block [4]
                if constexpr (MODE == 1 && RESN > 0) resid_load();
                PD_LEG_T0();
                PD_STRIP_WAITN((MODE == 1 ? 4 + RESN : 4), a0, a1, a2, a3);      // the next chunk's rows and first block have landed; the second block (and the residual tile) may still fly
                PD_LEG_ACC(leg_wa);
                PD_LEG_T0();
                __syncthreads();
                PD_LEG_ACC(leg_bar);
            }
        };
        if constexpr (RESN > 0) {
            if (nk64 < 2) {                                      // (a single chunk: nothing to hide behind)
                resid_load();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        for (int c = 0; c < nk64 - 2; ++c) chunk(c, std::integral_constant<int, 0>());
        if (nk64 >= 2) chunk(nk64 - 2, std::integral_constant<int, 1>());
        chunk(nk64 - 1, std::integral_constant<int, 2>());
        PD_STRIP_WAIT(0, b0, b1, b2, b3);
#if PD_STRIP_RES_AHEAD
        if constexpr (EPI == 2) {
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) asm volatile("" : "+v"(resid[mi][0]), "+v"(resid[mi][1]), "+v"(resid[mi][2]), "+v"(resid[mi][3]));     // (landed: vmcnt(0) above)
        }
#endif
        PD_LEG(4, __builtin_amdgcn_s_memtime());
#ifdef PD_STRIP_LEGS
        PD_LEG(6, (leg_wb << 32) | (leg_wa & 0xffffffffll));
        PD_LEG(7, leg_bar);
#ifdef PD_STRIP_STEP_CLOCKS
        PD_LEG(0, (leg_s0 << 32) | (leg_s1 & 0xffffffffll));       // (overwrites the wall-clock stamps: the step-timing build of the probe does not print the timeline)
        PD_LEG(1, (leg_s2 << 32) | (leg_s3 & 0xffffffffll));
#endif
#endif
#undef PD_STRIP_WLOAD
#undef PD_STRIP_WAIT
#undef PD_STRIP_WAITN
    } else {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) cw[c][st][pl] = wq[(size_t)c * KS * 128 + (size_t)(st * 2 + pl) * 64];
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < RT; ++mi)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][c][i] = 0.0f;
    const int nk = g.K / KC, sw = (l31 >> 1) & 7;
    for (int kc = 0; kc < nk; ++kc) {
        const int nc = min(kc + 1, nk - 1);          // the chunk after the last is the last again
        if constexpr (BARE == 0 || BARE == 2 || BARE >= 5) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) nw[c][st][pl] = wq[(size_t)c * KS * 128 + (size_t)(nc * 4 + st * 2 + pl) * 64];
        }
        if constexpr (BARE == 0 || BARE == 1 || BARE >= 5) stage(nc, (kc + 1) & 1);
        const unsigned *a = strip_lds + (BARE == 6 ? 0 : (kc & 1)) * CHA + l31 * KC;
        uint4 ah[RT], al[RT];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (BARE != 6 || kc == 0)
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) {
                const uint4 p = *(const uint4 *)(a + mi * 32 * KC + 4 * ((4 * st + 2 * hi) ^ sw));
                const uint4 q = *(const uint4 *)(a + mi * 32 * KC + 4 * ((4 * st + 2 * hi + 1) ^ sw));
                ah[mi] = make_uint4(__builtin_amdgcn_perm(p.y, p.x, 0x05040100u), __builtin_amdgcn_perm(p.w, p.z, 0x05040100u),
                                    __builtin_amdgcn_perm(q.y, q.x, 0x05040100u), __builtin_amdgcn_perm(q.w, q.z, 0x05040100u));
                al[mi] = make_uint4(__builtin_amdgcn_perm(p.y, p.x, 0x07060302u), __builtin_amdgcn_perm(p.w, p.z, 0x07060302u),
                                    __builtin_amdgcn_perm(q.y, q.x, 0x07060302u), __builtin_amdgcn_perm(q.w, q.z, 0x07060302u));
            }
            // per output element: lo x hi, hi x lo, hi x hi of this step, in this order (vit_gemm_split_kernel's)
            if constexpr (BARE != 5) {
#pragma unroll
                for (int mi = 0; mi < RT; ++mi)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[mi][c] = mma(al[mi], cw[c][st][0], acc[mi][c]);
#pragma unroll
                for (int mi = 0; mi < RT; ++mi)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[mi][c] = mma(ah[mi], cw[c][st][1], acc[mi][c]);
#pragma unroll
                for (int mi = 0; mi < RT; ++mi)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[mi][c] = mma(ah[mi], cw[c][st][0], acc[mi][c]);
            } else {
#pragma unroll
                for (int mi = 0; mi < RT; ++mi) acc[mi][0][st] += __uint_as_float(al[mi].x ^ ah[mi].w ^ cw[0][st][0].y ^ cw[0][st][1].z);
            }
        }
        if constexpr (BARE == 0 || BARE == 2 || BARE >= 5) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) cw[c][st][pl] = nw[c][st][pl];
        }
        if constexpr (BARE != 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next chunk's rows have landed in the other buffer
            __syncthreads();
        }
    }
    }
#if PD_STRIP_WIDE_EPI
    if constexpr (BARE == 0 && K64) {              // (the 64-k form's staging LDS holds the four waves' patches: 18 KB <= 32 KB at RT = 2)
        // Epilogue with 16-byte accesses (round 6; tools/strip_legs_probe.hip: the 4-byte form -- 16 loads + 16 stores per row tile and wave, each covering two
        // 128-byte row segments -- was 27 - 33 % of a workgroup's time, store-issue bound).  Every wave turns its 32 x 32 accumulator tile through a private 32 x 36
        // float patch of the (now idle) staging LDS: written column-per-lane as the MFMA leaves it, read back as four consecutive columns of one row per lane, so
        // that residual loads and stores are dwordx4 (four per row tile instead of sixteen).  Per element the arithmetic is the old epilogue's: the same bits.
        __syncthreads();                               // every wave is past its last fragment read (the last chunk ends without a barrier): the staging buffers are free
        float *patch = (float *)strip_lds + wave * (32 * 36);
        const int pr = lane >> 3, pc = (lane & 7) * 4;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int colb = n0 + (CT * wave + c) * 32 + pc;
            const float4 bias4 = *(const float4 *)(g.bias + colb);
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) {
#pragma unroll
                for (int i = 0; i < 16; ++i) patch[((i & 3) + 8 * (i >> 2) + 4 * hi) * 36 + l31] = acc[mi][c][i];
                float4 res4[4];
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#if PD_STRIP_RES_AHEAD
                        res4[q] = __builtin_bit_cast(float4, resid[mi][q]);
#else
                        res4[q] = *(const float4 *)((const float *)g.C + (size_t)min(m0 + mi * 32 + 8 * q + pr, g.M - 1) * g.Nout + colb);
#endif
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = m0 + mi * 32 + 8 * q + pr;
                    const float4 a4 = *(const float4 *)(patch + (8 * q + pr) * 36 + pc);
                    float v[4] = {a4.x, a4.y, a4.z, a4.w};
                    const float b4[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
                    const float r4[4] = {EPI == 2 ? res4[q].x : 0.0f, EPI == 2 ? res4[q].y : 0.0f, EPI == 2 ? res4[q].z : 0.0f, EPI == 2 ? res4[q].w : 0.0f};
                    unsigned o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = F16 ? fmaf(v[e], g.c_scale, b4[e]) : v[e] + b4[e];
                        if constexpr (EPI == 3) t = F16 ? 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)) : vit_gelu_fast(t);   // fp16 planes: nn.GELU()'s exact form
                        if constexpr (EPI == 4) t = pd_relu(t);
                        if constexpr (EPI == 2) t += r4[e];
                        if constexpr (EPI == 3 || EPI == 4) o[e] = pd_split_word_as<F16 ? 2 : 1>(t, g.out_scale);
                        else o[e] = __float_as_uint(t);
                    }
                    if (row < g.M) *(uint4 *)((unsigned *)g.C + (size_t)row * g.Nout + colb) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    } else
#endif
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = n0 + (CT * wave + c) * 32 + l31;
        const float bias = g.bias[col];
#pragma unroll
        for (int mi = 0; mi < RT; ++mi) {
            const int r0 = m0 + mi * 32 + 4 * hi;
            float res[16];
            if constexpr (EPI == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) res[i] = ((const float *)g.C)[(size_t)min(r0 + (i & 3) + 8 * (i >> 2), g.M - 1) * g.Nout + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = r0 + (i & 3) + 8 * (i >> 2);
                float v = F16 ? fmaf(acc[mi][c][i], g.c_scale, bias) : acc[mi][c][i] + bias;
                if constexpr (EPI == 3) v = F16 ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)) : vit_gelu_fast(v);   // fp16 planes: nn.GELU()'s exact form
                if constexpr (EPI == 4) v = pd_relu(v);
                if constexpr (EPI == 2) v += res[i];
                if (row < g.M) {
                    if constexpr (EPI == 3 || EPI == 4)
                        ((unsigned *)g.C)[(size_t)row * g.Nout + col] = pd_split_word_as<F16 ? 2 : 1>(v, g.out_scale);
                    else
                        ((float *)g.C)[(size_t)row * g.Nout + col] = v;
                }
            }
        }
    }
#ifdef PD_STRIP_LEGS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left
    PD_LEG(5, __builtin_amdgcn_s_memtime());
#ifndef PD_STRIP_STEP_CLOCKS
    PD_LEG(1, wall_clock64());
#endif
#endif
}
// dynamic LDS of a launch: two 32-k buffers, or (64-k form) two buffers of two 32-k blocks
template <int RT, bool K64>
static constexpr size_t pd_gemm_strip_lds() { return (size_t)(K64 ? 4 : 2) * 32 * RT * 32 * sizeof(unsigned); }
template <int EPI, int RT, bool F16, int CT = 1, bool K64 = false>
static inline void pd_gemm_strip(const unsigned *A, int lda, const unsigned *W, int K, const float *bias, void *C, int M, int Nout, hipStream_t s,
                                 float c_scale = 1.0f, float out_scale = 1.0f) {
    VitSplitArgs g{A, W, bias, C, M, Nout, K, lda, c_scale, out_scale};
    constexpr int TM = 32 * RT;
    hipLaunchKernelGGL((pd_gemm_strip_kernel<EPI, RT, F16, CT, 0, K64>), dim3(((M + TM - 1) / TM) * (Nout / (128 * CT))), dim3(256),
                       (pd_gemm_strip_lds<RT, K64>()), s, g);
}

static constexpr size_t pd_split_lds(int WM) { return (size_t)2 * 64 * WM * PD_STREAM_LR * sizeof(float); }
template <int EPI, int WM, int WN, bool F16 = false>
static inline void pd_gemm_split(const unsigned *A, int lda, const unsigned *W, int K, const float *bias, void *C, int M, int Nout, hipStream_t s,
                                 float c_scale = 1.0f, float out_scale = 1.0f) {
    VitSplitArgs g{A, W, bias, C, M, Nout, K, lda, c_scale, out_scale};
    constexpr int TM = 64 * WM, TN = 64 * WN;
    hipLaunchKernelGGL((vit_gemm_split_kernel<EPI, WM, WN, F16>), dim3(((M + TM - 1) / TM) * (Nout / TN)), dim3(256), pd_split_lds(WM), s, g);
}
