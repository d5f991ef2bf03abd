// pd_gemm_stream.h -- GEMM for many rows (>= PD_STREAM_MIN_ROWS), shared by the image feature extractor (pd_vit.hip) and
// the denoiser at large batches (pd_denoiser.hip): exact fp32 on v_mfma_f32_32x32x2_f32, A and W rows streamed through LDS.
#pragma once
#include "pd_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// split precision (see vit_gemm_split_kernel in pd_vit.hip): fp32 -> {hi bf16 | lo bf16 << 16}, v ~= hi + lo
__device__ __forceinline__ unsigned pd_split_word(float v) {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

// the same with fp16 halves: v ~= hi + lo to 22 mantissa bits while |lo| stays a normal fp16, i.e. |v| >= 2^-3 (below that the
// absolute error levels off at 2^-25) -- the caller scales v by a power of two into [.., 65504) first (pd_gemm_split.h, F16)
__device__ __forceinline__ unsigned pd_split_word_h(float v) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
// SPLIT (template parameter of the kernels that feed the split-precision GEMMs): 0 fp32, 1 bf16 words, 2 fp16 words of v * scale
template <int SPLIT>
__device__ __forceinline__ unsigned pd_split_word_as(float v, float scale) {
    if constexpr (SPLIT == 2) return pd_split_word_h(v * scale);
    return pd_split_word(v);
}

// sum over the 8 consecutive lanes that share an activation row in the staging of pd_gemm_stream_kernel (DPP: xor-1, xor-2
// quad permutes + half-row mirror; every lane of the group ends with the total)
template <int CTRL>
__device__ __forceinline__ float pd_stream_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float pd_stream_sum8(float v) {
    v = pd_stream_dpp_add<0xB1>(v);
    v = pd_stream_dpp_add<0x4E>(v);
    return pd_stream_dpp_add<0x141>(v);
}

// LayerNorm without affine (folded into the next weight): x [M, D] -> xn; one wave per row, D / 64 values per lane
template <int D, int SPLIT>
__global__ __launch_bounds__(256) void pd_ln_rows_kernel(const float *__restrict__ x, float *__restrict__ xn, int M, float eps, float scale) {
    constexpr int PER = D / 64;
    static_assert(D % 64 == 0, "one wave per row");
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float *src = x + (size_t)row * D;
    float v[PER], s = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = src[lane + 64 * i];
        s += v[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * (1.0f / D);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < PER; ++i) q += (v[i] - mean) * (v[i] - mean);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float o = (v[i] - mean) * rstd;
        if constexpr (SPLIT != 0)
            ((unsigned *)xn)[(size_t)row * D + lane + 64 * i] = pd_split_word_as<SPLIT>(o, scale);      // for vit_gemm_split_kernel
        else
            xn[(size_t)row * D + lane + 64 * i] = o;
    }
}
// LayerNorm statistics only: x [M, D] -> stats [M] = (mean, 1 / sqrt(var + eps)), two-pass in registers, one wave per row.  The streamed
// GEMM applies them while it stages its A rows (ALN): no normalised copy of the activations, and -- unlike a pre-pass inside the GEMM,
// which every column tile of a row block repeats (24 x for the 1 536-wide QKV: measured 18 - 20 % of the GEMM at 5 120 rows,
// profiles/round3_gemm_probe.txt) -- the rows are read once.
template <int D>
__global__ __launch_bounds__(256) void pd_ln_stats_kernel(const float *__restrict__ x, float2 *__restrict__ stats, int M, float eps) {
    constexpr int PER = D / 64;
    static_assert(D % 256 == 0, "one wave per row, float4 per lane");
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float4 *src = (const float4 *)(x + (size_t)row * D) + lane;
    float4 v[PER / 4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < PER / 4; ++i) {
        v[i] = src[64 * i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * (1.0f / D);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < PER / 4; ++i)
        q += ((v[i].x - mean) * (v[i].x - mean) + (v[i].y - mean) * (v[i].y - mean)) + ((v[i].z - mean) * (v[i].z - mean) + (v[i].w - mean) * (v[i].w - mean));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    if (lane == 0) stats[row] = make_float2(mean, 1.0f / sqrtf(q * (1.0f / D) + eps));
}
// W[n][k] * gamma[k] -> Wf (row-major copy with the LayerNorm scale folded in)
static __global__ void pd_scale_cols_kernel(const float *__restrict__ W, const float *__restrict__ gamma, int K, size_t total, float *__restrict__ Wf) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
        Wf[idx] = gamma ? W[idx] * gamma[idx % K] : W[idx];
}

//   C[m, n] = epi( sum_k A[m, k] W[n, k] + bias[n] ): a (64 WM) x (64 WN) tile per workgroup, one (32 WM) x (32 WN) quadrant
//   per wave over the whole K; A and W (both row-major, k contiguous) stream through LDS in 32-deep chunks, double buffered
//   with a register stage.  Blocks walk groups of ~2048 rows x all column tiles, so that a group's A rows and the whole W
//   stay in the L2s.  EPI 0: + bias   1: relu(+ bias)   2: + bias + C (residual, in place)   3: gelu(+ bias), exact erf form
//   4 (pd_gemm_dma_kernel only): + bias + R (another [M, Nout] array).
struct PdStreamArgs {
    const float *A, *W, *bias;
    float *C;
    int M, Nout, K, lda, ldw;
    const float2 *ln_stats;   // ALN only: (mean, rstd) of every A row (pd_ln_stats_kernel)
    const float *R;           // EPI 4 only: [M, Nout] added to the result (the hoisted z piece of the denoiser's _first)
};
#define PD_STREAM_KC 32
#define PD_STREAM_LR (PD_STREAM_KC + 4)      // LDS row stride: fragment reads and staging writes both conflict free
// staging registers are named scalars (arrays of float4 held across the K loop end up in scratch)
#define VS_EACH(X) X(0) X(1) X(2) X(3)

// ALN: A' = LayerNorm(A) without affine (gamma / beta folded into W / bias): every thread stages the same PA rows in every chunk, so it
// loads their (mean, rstd) from g.ln_stats once and the staging stores apply them -- no normalised copy of the activations in memory.
// BARE (tools/gemm_probe.hip only; 0 in the library): 1 = no global loads / LDS stores inside the K loop, 2 = also no barrier, 3 = also no
// fragment reads, 4 = global loads kept but never stored, 5 = LDS stores of stale registers -- bisects where the matrix pipe's idle time
// comes from (profiles/round3_gemm_probe.txt: of ~127 TFLOP/s for the bare matrix work at the clock the chip holds under this load, the
// fragment reads cost ~5 %, the LDS stores ~7 %, the global loads ~17 %; a second register stage, wave priorities, staggered starts and an
// XCD-owns-column-tiles block order were each measured and changed nothing).  Results are meaningless for BARE > 0.
template <int EPI, int WM, int WN, bool ALN = false, int BARE = 0>
__global__ __launch_bounds__(256) void pd_gemm_stream_kernel(PdStreamArgs g) {
    constexpr int KC = PD_STREAM_KC, LR = PD_STREAM_LR, TM = 64 * WM, TN = 64 * WN, PA = 2 * WM, PW = 2 * WN, GROUP = 2048 / TM;
    static_assert(KC == 32 && PA <= 4 && PW <= 4, "staging: 8 float4 per row, passes of 32 rows");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds, *Ws = lds + 2 * TM * LR;
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
    const int sr = tid >> 3, sc = tid & 7, st = sr * LR + 4 * sc;
#define VS_DECL(j)                                                                                                        \
    const float4 *ap##j = (const float4 *)(g.A + (size_t)min(m0 + sr + 32 * (j < PA ? j : 0), g.M - 1) * g.lda) + sc;       \
    const float4 *wp##j = (const float4 *)(g.W + (size_t)(n0 + sr + 32 * (j < PW ? j : 0)) * g.ldw) + sc;                   \
    float4 ra##j, rw##j;
#define VS_LOAD(j)                       \
    if constexpr (j < PA) ra##j = ap##j[nx]; \
    if constexpr (j < PW) rw##j = wp##j[nx];
#define VS_STORE(j)                                                                                          \
    if constexpr (j < PA) {                                                                                   \
        float4 t_ = ra##j;                                                                                    \
        if constexpr (ALN) {                                                                                  \
            t_.x = (t_.x - ln_mu[j < PA ? j : 0]) * ln_rs[j < PA ? j : 0];                                    \
            t_.y = (t_.y - ln_mu[j < PA ? j : 0]) * ln_rs[j < PA ? j : 0];                                    \
            t_.z = (t_.z - ln_mu[j < PA ? j : 0]) * ln_rs[j < PA ? j : 0];                                    \
            t_.w = (t_.w - ln_mu[j < PA ? j : 0]) * ln_rs[j < PA ? j : 0];                                    \
        }                                                                                                     \
        *(float4 *)(da + st + j * 32 * LR) = t_;                                                              \
    }                                                                                                         \
    if constexpr (j < PW) *(float4 *)(dw + st + j * 32 * LR) = rw##j;
    VS_EACH(VS_DECL)
    float ln_mu[PA], ln_rs[PA];
    if constexpr (ALN) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const float2 st2 = g.ln_stats[min(m0 + sr + 32 * j, g.M - 1)];
            ln_mu[j] = st2.x;
            ln_rs[j] = st2.y;
        }
    }
    {
        const int nx = 0;
        float *da = As, *dw = Ws;
        VS_EACH(VS_LOAD)
        VS_EACH(VS_STORE)
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
    const int aoff = (wm * 32 * WM + l31) * LR + 4 * hi, boff = (wn * 32 * WN + l31) * LR + 4 * hi;
    // the matrix work of one K chunk out of LDS buffer `buf`
    auto mma_chunk = [&](int buf) {
        const float *a = As + buf * TM * LR + aoff, *b = Ws + buf * TN * LR + boff;
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            float4 af[WM], bf[WN];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) af[mi] = (BARE != 3) ? *(const float4 *)(a + mi * 32 * LR + kk * 8) : make_float4(1.f + kk, 2.f, 3.f, 4.f);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) bf[ni] = (BARE != 3) ? *(const float4 *)(b + ni * 32 * LR + kk * 8) : make_float4(1.f, 2.f + kk, 3.f, 4.f);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
    };
    for (int kc = 0; kc < nk; ++kc) {
        // the chunk after the last is the last again: loads and LDS writes stay unconditional (straight-line loop body)
        const int nx = min(kc + 1, nk - 1) * (KC / 4);
        if constexpr (BARE == 0 || BARE == 4) {      // (4: global loads kept, LDS stores dropped; 5: the reverse)
            VS_EACH(VS_LOAD)
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ahead of the matrix work
        mma_chunk(kc & 1);
        __builtin_amdgcn_sched_barrier(0);
        float *da = As + ((kc + 1) & 1) * TM * LR, *dw = Ws + ((kc + 1) & 1) * TN * LR;
        if constexpr (BARE == 0 || BARE == 5) {
            VS_EACH(VS_STORE)
        }
        if constexpr (BARE == 4) {                   // keep the loads alive without storing them
            asm volatile("" ::"v"(ra0.x), "v"(ra1.x), "v"(rw0.x), "v"(rw1.x));
        }
        if constexpr (BARE < 2 || BARE >= 4) __syncthreads();
    }
#undef VS_DECL
#undef VS_LOAD
#undef VS_STORE
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int col = n0 + (wn * WN + ni) * 32 + l31, r0 = m0 + (wm * WM + mi) * 32 + 4 * hi;
            const float bias = g.bias[col];
            float res[16];
            if constexpr (EPI == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i)      // all residual loads in flight at once (rows past M re-read the last row)
                    res[i] = g.C[(size_t)min(r0 + (i & 3) + 8 * (i >> 2), g.M - 1) * g.Nout + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = r0 + (i & 3) + 8 * (i >> 2);
                float v = acc[mi][ni][i] + bias;
                if constexpr (EPI == 1) v = pd_relu(v);
                if constexpr (EPI == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                if constexpr (EPI == 2) v += res[i];
                if (row < g.M) g.C[(size_t)row * g.Nout + col] = v;
            }
        }
}


// ---- the same GEMM with LDS-DMA staging (64 x 64 tiles) -------------------------------------------------------------------------------
// pd_gemm_stream_kernel stages through registers: global_load -> VGPRs -> ds_write.  Bisecting it (BARE above) shows the matrix pipe idle
// ~17 % of the time because of those global loads and ~7 % because of the LDS stores -- whatever the prefetch depth, the block order or
// the wave priorities.  Here the K chunks go from L2 straight into LDS (global_load_lds_dwordx4: 64 lanes x 16 B land lane-linear at a
// wave-uniform LDS address; no VGPR round trip, no ds_write).  A lane-linear image has no row padding, so the 16-byte slots of a row are
// XOR-swizzled instead -- slot s of row r holds k-chunk s ^ ((r >> 1) & 7); a lane may fetch any 16 bytes it likes -- which keeps the
// fragment reads (32 rows x one k-chunk per half wave) conflict free.  LayerNorm (ALN) moves from the staging stores to the fragment
// reads: (a - mean) * rstd with the statistics of the lane's own row, the same two roundings.  Same MFMA chain as the register-staged
// kernel: bitwise the same C.
#ifndef PD_DMA_WIDE_EPI
#define PD_DMA_WIDE_EPI 1
#endif
// one 1 KiB piece: 64 lanes x 16 B from `base` + the lane's byte offset to the wave-uniform LDS byte address `dst`
__device__ __forceinline__ void pd_dma_piece(const float *base, unsigned off, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o], %[b]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep)
                 : [d] "s"(dst), [b] "s"(base), [o] "v"(off)
                 : "memory");
}
// (64 WM) x (64 WN) tile per workgroup, a (32 WM) x (32 WN) quadrant per wave (WM, WN in {1, 2})
template <int EPI, bool ALN, int WM = 1, int WN = 1>
__global__ __launch_bounds__(256) void pd_gemm_dma_kernel(PdStreamArgs g) {
    constexpr int KC = 32, TM = 64 * WM, TN = 64 * WN, GROUP = 2048 / TM, CHA = TM * KC, CHW = TN * KC;   // floats per operand and chunk
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    float *As = lds, *Ws = lds + 2 * CHA;
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
    // staging: pieces of 1 KiB = 8 rows x 32 floats; a chunk has 8 WM of A and 8 WN of W, wave w moves pieces [2 WM w, 2 WM (w + 1)) of A
    // and [2 WN w, 2 WN (w + 1)) of W.  Hand-issued (a wave-uniform 64-bit base in SGPRs + a 32-bit byte offset per lane, M0 = the LDS
    // byte address): the compiler's own waitcnt logic cannot tell the two LDS buffers apart and would drain the DMA before the first
    // fragment read of the chunk that hides it.
    const int prow = lane >> 3, pslot = lane & 7;
    unsigned oa[2 * WM], ow[2 * WN];
#pragma unroll
    for (int j = 0; j < 2 * WM; ++j) {
        const int r = 8 * (2 * WM * wave + j) + prow;                              // row of the tile this lane fetches
        oa[j] = (unsigned)(((size_t)min(m0 + r, g.M - 1) * g.lda + 4 * (pslot ^ ((r >> 1) & 7))) * sizeof(float));
    }
#pragma unroll
    for (int j = 0; j < 2 * WN; ++j) {
        const int r = 8 * (2 * WN * wave + j) + prow;
        ow[j] = (unsigned)(((size_t)(n0 + r) * g.ldw + 4 * (pslot ^ ((r >> 1) & 7))) * sizeof(float));
    }
    const unsigned lds_a = (unsigned)(size_t)(As + 2 * WM * wave * 256), lds_w = (unsigned)(size_t)(Ws + 2 * WN * wave * 256);   // LDS byte addresses
    auto stage = [&](int kc, int buf) {
        const float *ab = g.A + kc * KC, *wb = g.W + kc * KC;
        const unsigned da = __builtin_amdgcn_readfirstlane(lds_a + buf * CHA * 4), dw = __builtin_amdgcn_readfirstlane(lds_w + buf * CHW * 4);
#pragma unroll
        for (int j = 0; j < 2 * WM; ++j) pd_dma_piece(ab, oa[j], da + j * 1024);
#pragma unroll
        for (int j = 0; j < 2 * WN; ++j) pd_dma_piece(wb, ow[j], dw + j * 1024);
    };
    // fragments: lane (l31, hi) reads k-chunk 2 kk + hi of row (wave's rows) + 32 mi + l31; the swizzle term (row >> 1) & 7 is that of l31
    const int arow = wm * 32 * WM + l31, brow = wn * 32 * WN + l31, sw = (l31 >> 1) & 7;
    float ln_mu[WM], ln_rs[WM];
    if constexpr (ALN) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
            const float2 st2 = g.ln_stats[min(m0 + arow + 32 * mi, g.M - 1)];
            ln_mu[mi] = st2.x;
            ln_rs[mi] = st2.y;
        }
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.0f;
    const int nk = g.K / KC;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) stage(kc + 1, (kc + 1) & 1);   // (round 6: the last chunk requests nothing; it used to re-stage itself to keep the loop body straight-line)
        const float *a = As + (kc & 1) * CHA + arow * KC, *b = Ws + (kc & 1) * CHW + brow * KC;
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            const int so = 4 * ((2 * kk + hi) ^ sw);
            float4 af[WM], bf[WN];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                af[mi] = *(const float4 *)(a + mi * 32 * KC + so);
                if constexpr (ALN) {
                    af[mi].x = (af[mi].x - ln_mu[mi]) * ln_rs[mi];
                    af[mi].y = (af[mi].y - ln_mu[mi]) * ln_rs[mi];
                    af[mi].z = (af[mi].z - ln_mu[mi]) * ln_rs[mi];
                    af[mi].w = (af[mi].w - ln_mu[mi]) * ln_rs[mi];
                }
            }
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) bf[ni] = *(const float4 *)(b + ni * 32 * KC + so);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next chunk has landed in the other buffer
        __syncthreads();
    }
#if PD_DMA_WIDE_EPI
    {
        // Epilogue with 16-byte accesses (round 6, as in pd_gemm_strip_kernel): every wave turns its 32 x 32 accumulator tiles through a private 32 x 36 float patch of
        // the idle staging LDS so that a lane holds four consecutive columns of a row: residual loads and stores are dwordx4.  The same arithmetic per element.
        __syncthreads();                               // (kept: cheap, and the staging buffers are free for certain)
        float *patch = lds + wave * (32 * 36);
        const int pr = lane >> 3, pc = (lane & 7) * 4;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) {
                const int colb = n0 + (wn * WN + ni) * 32 + pc, rb = m0 + (wm * WM + mi) * 32;
                const float4 bias4 = *(const float4 *)(g.bias + colb);
#pragma unroll
                for (int i = 0; i < 16; ++i) patch[((i & 3) + 8 * (i >> 2) + 4 * hi) * 36 + l31] = acc[mi][ni][i];
                float4 res4[4];
                if constexpr (EPI == 2 || EPI == 4) {
                    const float *rsrc = EPI == 2 ? g.C : g.R;
#pragma unroll
                    for (int q = 0; q < 4; ++q) res4[q] = *(const float4 *)(rsrc + (size_t)min(rb + 8 * q + pr, g.M - 1) * g.Nout + colb);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rb + 8 * q + pr;
                    const float4 a4 = *(const float4 *)(patch + (8 * q + pr) * 36 + pc);
                    float v[4] = {a4.x + bias4.x, a4.y + bias4.y, a4.z + bias4.z, a4.w + bias4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (EPI == 1) v[e] = pd_relu(v[e]);
                        if constexpr (EPI == 3) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
                    }
                    if constexpr (EPI == 2 || EPI == 4) {
                        v[0] += res4[q].x; v[1] += res4[q].y; v[2] += res4[q].z; v[3] += res4[q].w;
                    }
                    if (row < g.M) *(float4 *)(g.C + (size_t)row * g.Nout + colb) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        return;
    }
#endif
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int col = n0 + (wn * WN + ni) * 32 + l31, r0 = m0 + (wm * WM + mi) * 32 + 4 * hi;
            const float bias = g.bias[col];
            float res[16];
            if constexpr (EPI == 2 || EPI == 4) {
                const float *rsrc = EPI == 2 ? g.C : g.R;
#pragma unroll
                for (int i = 0; i < 16; ++i) res[i] = rsrc[(size_t)min(r0 + (i & 3) + 8 * (i >> 2), g.M - 1) * g.Nout + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = r0 + (i & 3) + 8 * (i >> 2);
                float v = acc[mi][ni][i] + bias;
                if constexpr (EPI == 1) v = pd_relu(v);
                if constexpr (EPI == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                if constexpr (EPI == 2 || EPI == 4) v += res[i];
                if (row < g.M) g.C[(size_t)row * g.Nout + col] = v;
            }
        }
}

template <int EPI, bool ALN = false, int WM = 1, int WN = 1>
static inline void pd_gemm_dma(const float *A, int lda, const float *W, int K, const float *bias, float *C, int M, int Nout, hipStream_t s,
                               const float2 *ln_stats = nullptr, const float *R = nullptr) {
    PdStreamArgs g{A, W, bias, C, M, Nout, K, lda, K, ln_stats, R};
    hipLaunchKernelGGL((pd_gemm_dma_kernel<EPI, ALN, WM, WN>), dim3(((M + 64 * WM - 1) / (64 * WM)) * (Nout / (64 * WN))), dim3(256),
                       (size_t)2 * (64 * WM + 64 * WN) * 32 * sizeof(float), s, g);
}

#define PD_STREAM_MIN_ROWS 1024
// Tile shapes, every one alone at 5 120 / 15 360 rows (profiles/round3_gemm_probe.txt, TFLOP/s): 64 x 64 is the best or equal for the
// 512- and 1 024-wide outputs (84 - 88 at 5 120 rows, 128-row / 128-column tiles 66 - 75 there: too few workgroups); the 1 536-wide QKV
// runs 88 -> 97 on 128 x 128 tiles (480 workgroups of four 64 x 64 quadrants: half the LDS traffic per FLOP); at 15 360 rows the large
// tiles lead everywhere by 2 - 10 %.  The image feature extractor measured 64 x 64 best in round 1 (profiles/round1_j_vit_notes.md).
template <int EPI, bool ALN = false, int WM = 1, int WN = 1>
static inline void pd_gemm_stream(const float *A, int lda, const float *W, int K, const float *bias, float *C, int M, int Nout, hipStream_t s,
                                  const float2 *ln_stats = nullptr) {
    PdStreamArgs g{A, W, bias, C, M, Nout, K, lda, K, ln_stats, nullptr};
    const size_t lds = (size_t)2 * (64 * WM + 64 * WN) * PD_STREAM_LR * sizeof(float);
    hipLaunchKernelGGL((pd_gemm_stream_kernel<EPI, WM, WN, ALN>), dim3(((M + 64 * WM - 1) / (64 * WM)) * (Nout / (64 * WN))), dim3(256), lds, s, g);
}
