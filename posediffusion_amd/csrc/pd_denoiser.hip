// pd_denoiser.hip -- the transformer Denoiser + DDPM posterior update as hand-written gfx950 kernels.
//
// Replaces (paths relative to /root/reference/pose_diffusion/):
//   models/denoiser.py:53-76        Denoiser.forward (embed -> _first -> 8 encoder layers -> _last)
//   models/denoiser.py:79-98        nn.TransformerEncoderLayer, pre-norm, ReLU, eps 1e-5, eval mode
//   util/embedding.py:13-50         TimeStepEmbedding (hoisted into a [T,128] table) + PoseEmbedding
//   models/gaussian_diffuser.py:190-209, :231-246, :280   x0 / posterior mean / sample update
//
// Numerics: everything is fp32.  GEMMs run on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, 157 TF peak) so the engine matches the
// reference's fp32 path to rounding-order differences only (SURVEY.md headline fact 5).
//
// GEMM structure (M = B*N tokens is tiny: 20..1280; weights are [out,in] row-major = "B^T"):
//   * one workgroup (4 waves) per 32x32 output tile; the four waves split K (each SIMD's matrix
//     pipe works on a quarter of K) and their accumulators are summed through LDS in fixed order;
//   * the 32 activation rows are staged once in LDS in full 128-B lines (row stride K+4 floats:
//     ds_read_b128 fragment reads are conflict-free) with LayerNorm / the harmonic+time+z
//     embedding fused into the staging pass, so no normalised activations ever touch HBM;
//   * weights are re-packed at engine creation into MFMA-fragment order
//     Wp[n_tile][k_chunk][lane][4] so every wave-level load is one fully coalesced 1 KiB line
//     streamed straight to VGPRs (each weight byte is read by exactly one wave per M-tile);
//   * bias / ReLU / residual are fused into the epilogue.
#include "pd_denoiser_dev.h"
#include "pd_gemm_stream.h"
#include "pd_gemm_split.h"
#include "pd_qkv_attn.h"
#ifndef PD_STRIP_RT3
#define PD_STRIP_RT3 1         // round 6: 96-row tiles for the 512-wide strip GEMMs when that takes them from more tiles than CUs to at most one per CU
#endif
#ifndef PD_STRIP_RT1
#define PD_STRIP_RT1 0         // (probe: 32-row tiles instead -- 640 half tiles, more workgroups per CU, twice the weight bytes per MFMA)
#endif
#ifndef PD_STRIP_RT3_FF1
#define PD_STRIP_RT3_FF1 0     // (the 1 024-wide FF1 has 640 tiles at 64 rows, 432 at 96: three tiles of 1 or two of 1.5 on the busiest CU -- the same)
#endif
#ifndef PD_STRIP_K64
#define PD_STRIP_K64 true      // the strip GEMMs of the fp16-plane mode: A chunks of 64 k per barrier (pd_gemm_split.h)
#endif

#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>

// --------------------------------------------------------------------------------------------
// weight repack: W[Nout][K] row-major -> MFMA-fragment order (zero padded), optionally with a
// LayerNorm gamma folded in as a column scale (W' = W diag(gamma): LN(x) W^T = xhat (W diag(gamma))^T + W beta)
//   NT = 32 (v_mfma_f32_32x32x2_f32):  Wp[nt][kc][lane][4] = W[nt*32 + (l & 31)][kc*8  + 4*(l >> 5) + e]
//   NT = 16 (v_mfma_f32_16x16x4_f32):  Wp[nt][kc][lane][4] = W[nt*16 + (l & 15)][kc*16 + 4*(l >> 4) + e]
// --------------------------------------------------------------------------------------------
__global__ void pd_repack_kernel(const float *__restrict__ W, int Nout, int K, int KC, float *__restrict__ Wp, size_t total,
                                 int first_perm, int nt_width, const float *__restrict__ colscale) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int l = (idx >> 2) & 63;
        const size_t rest = idx >> 8;
        const int kc = (int)(rest % KC);
        const int nt = (int)(rest / KC);
        int n, k;
        if (nt_width == 32) {
            n = nt * 32 + (l & 31);
            k = kc * 8 + 4 * (l >> 5) + e;
        } else {
            n = nt * 16 + (l & 15);
            k = kc * 16 + 4 * (l >> 4) + e;
        }
        if (first_perm) k = pd_first_col_all(k);
        float v = (n < Nout && k < K) ? W[(size_t)n * K + k] : 0.0f;
        if (colscale && k < K) v *= colscale[k];
        Wp[idx] = v;
    }
}

// b'[n] = b[n] + sum_k W[n][k] beta[k]   (the LayerNorm shift folded into the following bias)
__global__ void pd_fold_bias_kernel(const float *__restrict__ W, const float *__restrict__ beta, const float *__restrict__ b,
                                    int Nout, int K, float *__restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Nout) return;
    float a = 0.0f;
    for (int k = 0; k < K; ++k) a = fmaf(W[(size_t)n * K + k], beta[k], a);
    out[n] = b[n] + a;
}

// _first's STEP rows for the streamed path (>= PD_STREAM_MIN_ROWS token rows): [harmonic(x) (180) | x (9) | pivot | 0 0] = KFIRST_D
// columns (piece PD_FIRST_D of pd_denoiser_dev.h), one wave per row, written once per step and read by pd_gemm_dma like any activation
// (denoiser.py:60-68; the same expressions as the AMODE 2 staging of the small-batch pd_gemm_kernel).  z and t_emb never enter the loop: their products
// are hoisted (pd_denoiser_prepare, pd_first_ttab_kernel).
__global__ __launch_bounds__(256) void pd_embed_rows_kernel(const float *__restrict__ x, int n_frames, int M, float *__restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float4 *dst = (float4 *)(out + (size_t)row * KFIRST_D);
    float xv[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) xv[d] = x[(size_t)row * 9 + d];
    if (lane < 45) {                                            // harmonic: 180 values = 45 float4 at [0, 45)
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = 4 * lane + e, s = idx / 90, rem = idx - s * 90, d = rem / 10, kk = rem - d * 10;
            float xd = xv[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) xd = (d == q) ? xv[q] : xd;
            const float a = xd * (float)(1 << kk);
            o[e] = sinf(s ? a + 1.5707963267948966f : a);
        }
        dst[lane] = make_float4(o[0], o[1], o[2], o[3]);
    } else if (lane == 45) {
        dst[45] = make_float4(xv[0], xv[1], xv[2], xv[3]);
    } else if (lane == 46) {
        dst[46] = make_float4(xv[4], xv[5], xv[6], xv[7]);
    } else if (lane == 47) {
        dst[47] = make_float4(xv[8], (row % n_frames == 0) ? 1.0f : 0.0f, 0.0f, 0.0f);   // pivot one-hot on frame 0, padding
    }
}
// a piece of W_first [512, 702] -> row-major [512, Kdst] in the engine's column order of that piece (pd_first_col)
__global__ void pd_first_rowmajor_kernel(const float *__restrict__ W, float *__restrict__ Wf, int piece, int Kdst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= DM * Kdst) return;
    const int n = idx / Kdst, k = pd_first_col(piece, idx - n * Kdst);
    Wf[idx] = k < KFIRST ? W[(size_t)n * KFIRST + k] : 0.0f;
}
// the time piece of _first: ttab[t][n] = sum_k W_first[n][189 + k] t_emb(t)[k], an fmaf chain over the 128 columns (one block per t)
__global__ __launch_bounds__(DM) void pd_first_ttab_kernel(const float *__restrict__ W, const float *__restrict__ t_table, float *__restrict__ ttab) {
    __shared__ float te[128];
    const int t = blockIdx.x, n = threadIdx.x;
    if (n < 128) te[n] = t_table[(size_t)t * 128 + n];
    __syncthreads();
    const float *w = W + (size_t)n * KFIRST + pd_first_col(PD_FIRST_T, 0);
    float a = 0.0f;
    for (int k = 0; k < 128; ++k) a = fmaf(te[k], w[k], a);
    ttab[(size_t)t * DM + n] = a;
}

// time-step embedding (util/embedding.py:28-37) of one timestep value t: 128 threads, thread i owns output i
__device__ __forceinline__ float pd_time_embed_one(float t, const float *__restrict__ w0, const float *__restrict__ b0,
                                                   const float *__restrict__ w2, const float *__restrict__ b2, float *emb, float *hid) {
    const int i = threadIdx.x;
    // freqs = exp(-ln(10000) * arange(128, fp32) / 128)  (embedding.py:24-26), args = t * freqs
    const float freq = expf((-9.210340371976184f * (float)i) / 128.0f);
    const float arg = t * freq;
    emb[i] = cosf(arg);
    emb[128 + i] = sinf(arg);
    __syncthreads();
    float a = b0[i];
    for (int k = 0; k < 256; ++k) a = fmaf(emb[k], w0[i * 256 + k], a);
    hid[i] = a / (1.0f + expf(-a));   // SiLU
    __syncthreads();
    float o = b2[i];
    for (int k = 0; k < 128; ++k) o = fmaf(hid[k], w2[i * 128 + k], o);
    return o;
}
// the engine's table: one block per step t = 0 .. T-1
__global__ void pd_time_table_kernel(const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w2,
                                     const float *__restrict__ b2, float *__restrict__ table) {
    __shared__ float emb[256];
    __shared__ float hid[128];
    table[blockIdx.x * 128 + threadIdx.x] = pd_time_embed_one((float)blockIdx.x, w0, b0, w2, b2, emb, hid);
}
// TimeStepEmbedding.forward for arbitrary timesteps (pd_time_embedding): the same arithmetic, one block per entry of tvals
__global__ void pd_time_embed_kernel(const float *__restrict__ tvals, const float *__restrict__ w0, const float *__restrict__ b0,
                                     const float *__restrict__ w2, const float *__restrict__ b2, float *__restrict__ out) {
    __shared__ float emb[256];
    __shared__ float hid[128];
    out[(size_t)blockIdx.x * 128 + threadIdx.x] = pd_time_embed_one(tvals[blockIdx.x], w0, b0, w2, b2, emb, hid);
}
// PoseEmbedding.forward = pytorch3d HarmonicEmbedding(n = 10, append_input = True) of rows [rows, dim] (pd_pose_embedding):
// out [rows, 21 dim] = [sin(x_d 2^k) (d-major, k = 0..9) | sin(x_d 2^k + pi / 2) | x] -- the expressions of pd_embed_rows_kernel and
// of the AMODE 2 staging, in the reference's own column order
__global__ void pd_harmonic_rows_kernel(const float *__restrict__ x, long long rows, int dim, float *__restrict__ out) {
    const int per = 21 * dim;
    const long long total = rows * per;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / per;
        const int c = (int)(idx - row * per);
        float v;
        if (c >= 20 * dim) {
            v = x[row * dim + (c - 20 * dim)];
        } else {
            const int s = c / (10 * dim), rem = c - s * 10 * dim, d = rem / 10, kk = rem - d * 10;
            const float a = x[row * dim + d] * (float)(1 << kk);
            v = sinf(s ? a + 1.5707963267948966f : a);
        }
        out[idx] = v;
    }
}

// --------------------------------------------------------------------------------------------
// fused 32x32-tile GEMM:  C[m, n] = epi( sum_k A'[m, k] * W[n, k] + bias[n] )
//   AMODE 0: A' = A                      (plain rows of a [M, K] activation)
//   AMODE 1: A' = LayerNorm(A) (K = 512) (norm_first encoder layer, eps 1e-5)
//   AMODE 2: A' = [z | t_emb | harmonic(x) | x | pivot | 0 0]  (K = 704, denoiser.py:56-68; the engine's column order pd_first_col_all)
//   EPI   0: + bias     1: relu(+ bias)     2: + bias + residual (in place on C)
// --------------------------------------------------------------------------------------------
// -DPD_DEN_STAMPS (tools/den_small_legs.py; never in the product build): every launch of the small-batch chain records, from lane 0 of
// wave 0 of its block 0, the constant 100 MHz clock (s_memrealtime: comparable across kernels and CUs) at the legs of its latency chain
#ifdef PD_DEN_STAMPS
#define PD_STAMP(ptr, i)                                                                          \
    do {                                                                                          \
        if ((ptr) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) (ptr)[i] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define PD_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define PD_STAMP(ptr, i) do { } while (0)
#define PD_STAMP_DRAIN() do { } while (0)
#endif
struct GemmArgs {
#ifdef PD_DEN_STAMPS
    long long *stamps;     // [8] of this launch, or null
#endif
    const float *A;        // [M, K] (AMODE 0/1)
    const float *Wp;       // packed weights
    const float *bias;     // [Nout]
    float *C;              // [M, Nout]
    // AMODE 2
    const float *x, *z, *temb;   // x [M,9], z [M,384], temb [128] (row of the table for this t)
    int n_frames;
    int M, Nout;
    int MT;                // number of 32-row M tiles (XCD-aware block mapping)
};

template <int K, int AMODE, int EPI, int NT>
__global__ __launch_bounds__(256) void pd_gemm_kernel(GemmArgs g) {
    constexpr int LDA = K + 4;            // padded row stride (floats): conflict-free ds_read_b128
    constexpr int CW = (NT == 32) ? 8 : 16;   // k-chunk width per float4 fragment load
    constexpr int KC = K / CW;
    constexpr int CPW = KC / 4;           // chunks per wave (split-K over the 4 waves)
    constexpr int NB = (CPW > 16) ? 2 : 1;   // weight batches held in registers
    constexpr int BATCH = CPW / NB;
    constexpr int NACC = (NT == 32) ? 16 : 8;
    static_assert(KC % 4 == 0 && CPW % NB == 0, "chunk batching");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                      // [32][LDA]; later aliased by the cross-wave reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PD_DEN_STAMPS
    long long *const stamps = g.stamps;
    PD_STAMP(stamps, 0);                  // entered
#endif
    // XCD-aware tile mapping (guide T1): the dispatcher places block id on XCD id % 8; all M-tiles that share
    // an N-tile are given ids with the same id % 8, so each weight tile is fetched into ONE L2 once and the
    // other M-tile workgroups hit it there (the naive (m + MT*n) order spread them over MT different XCDs
    // and re-fetched every weight byte MT times).  Needs (Nout / NT) % 8 == 0 -- true for every layer here.
    const int bid = blockIdx.x, slot = bid >> 3;
    const int ntile = (bid & 7) + 8 * (slot / g.MT), mtile = slot % g.MT;
    const int m0 = mtile * 32, n0 = ntile * NT;
    const float4 *wp = (const float4 *)g.Wp + ((size_t)ntile * KC + (size_t)wave * CPW) * 64 + lane;

    // ---- weights first: the whole first batch of this wave's fragments goes in flight before the
    // activation staging, so the HBM/MALL latency of the weight stream hides under it --------------
    float4 w0[BATCH];
#pragma unroll
    for (int c = 0; c < BATCH; ++c) w0[c] = wp[(size_t)c * 64];
    // ... and the bias the epilogue adds (requested behind the two barriers below it is a dependent L2 round trip at the very end: -3.5 us per
    // step at B = 1).  The residual values (EPI 2) stay where they are: requested up here they cost +4 us per kernel (measured, tools/den_ab.py).
    constexpr int RPW = NACC / 4;          // accumulator registers finished per wave
    const int col = n0 + ((NT == 32) ? (lane & 31) : (lane & 15));
    const float bias = g.bias[col];

    // ---- stage the 32 activation rows (fused LN / embedding); no predicated loads ---------------
    {
        const int r = tid >> 3, sub = tid & 7;
        const int m = m0 + r;
        const bool live = m < g.M;
        const int mr = live ? m : g.M - 1;   // clamp: padded rows load a valid row and are zeroed
        float *dst = As + r * LDA;
        if constexpr (AMODE == 2) {
            // engine column order (pd_first_col_all): z | t_emb | harmonic | x | pivot | pad
            const float4 *zr = (const float4 *)(g.z + (size_t)mr * ZD);
            const float4 *te = (const float4 *)g.temb;
            float4 zv[ZD / 32], tv[4];
#pragma unroll
            for (int i = 0; i < ZD / 32; ++i) zv[i] = zr[sub + 8 * i];
#pragma unroll
            for (int i = 0; i < 4; ++i) tv[i] = te[sub + 8 * i];
            float xv[9];
#pragma unroll
            for (int d = 0; d < 9; ++d) xv[d] = g.x[(size_t)mr * 9 + d];
            const float keep = live ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < ZD / 32; ++i) {
                float4 v = zv[i];
                v.x *= keep; v.y *= keep; v.z *= keep; v.w *= keep;
                *(float4 *)(dst + 4 * (sub + 8 * i)) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 v = tv[i];
                v.x *= keep; v.y *= keep; v.z *= keep; v.w *= keep;
                *(float4 *)(dst + 384 + 4 * (sub + 8 * i)) = v;
            }
            // harmonic embedding: idx = s*90 + d*10 + k -> sin(x_d * 2^k + s * pi/2)  (pytorch3d 0.7.x)
            for (int idx = sub; idx < 180; idx += 8) {
                const int s = idx / 90, rem = idx - s * 90, d = rem / 10, kk = rem - d * 10;
                float xd = xv[0];
#pragma unroll
                for (int q = 1; q < 9; ++q) xd = (d == q) ? xv[q] : xd;
                const float e = xd * (float)(1 << kk);
                dst[512 + idx] = keep * sinf(s ? e + 1.5707963267948966f : e);
            }
            if (sub == 0) {
#pragma unroll
                for (int d = 0; d < 9; ++d) dst[692 + d] = keep * xv[d];
                dst[701] = (live && (m % g.n_frames == 0)) ? 1.0f : 0.0f;   // pivot one-hot on frame 0
                dst[702] = 0.0f;
                dst[703] = 0.0f;
            }
        } else if constexpr (AMODE == 1) {
            // LayerNorm without affine: gamma is folded into the packed weights, beta into the bias
            static_assert(AMODE != 1 || K == 512, "LayerNorm staging is built for d_model = 512");
            float4 v[K / 32];
            const float4 *src = (const float4 *)(g.A + (size_t)mr * K);
#pragma unroll
            for (int i = 0; i < K / 32; ++i) v[i] = src[sub + 8 * i];
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            const float mean = pd_sum8(s) * (1.0f / K);
            float q = 0.0f;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) {
                const float a = v[i].x - mean, b2 = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b2 * b2) + (c * c + d * d);
            }
            const float rstd = live ? 1.0f / sqrtf(pd_sum8(q) * (1.0f / K) + 1e-5f) : 0.0f;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) {
                float4 o;
                o.x = (v[i].x - mean) * rstd;
                o.y = (v[i].y - mean) * rstd;
                o.z = (v[i].z - mean) * rstd;
                o.w = (v[i].w - mean) * rstd;
                *(float4 *)(dst + 4 * (sub + 8 * i)) = o;
            }
        } else {
            const float4 *src = (const float4 *)(g.A + (size_t)mr * K);
            const float keep = live ? 1.0f : 0.0f;
            constexpr int NV = K / 32;
            constexpr int VB = NV < 16 ? NV : 16;        // loads in flight per pass
            static_assert(NV % VB == 0, "passes of VB float4 per thread");
#pragma unroll
            for (int i0 = 0; i0 < NV; i0 += VB) {
                float4 v[VB];
#pragma unroll
                for (int i = 0; i < VB; ++i) v[i] = src[sub + 8 * (i0 + i)];
#pragma unroll
                for (int i = 0; i < VB; ++i) {
                    float4 o = v[i];
                    o.x *= keep; o.y *= keep; o.z *= keep; o.w *= keep;
                    *(float4 *)(dst + 4 * (sub + 8 * (i0 + i))) = o;
                }
            }
        }
    }
    PD_STAMP(stamps, 1);                  // this thread's share of the A rows loaded (arrived from L2 / MALL), normalised, written to LDS
    __syncthreads();
    PD_STAMP(stamps, 2);                  // every wave's share staged
#ifdef PD_DEN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PD_STAMP(stamps, 3);                  // ... and the first batch of weight fragments + the bias have landed (stamps build only: the wait)
#endif

    // ---- split-K MFMA loop: wave w owns k-chunks [w*CPW, (w+1)*CPW) --------------------------
    float4 w1[NB == 2 ? BATCH : 1];
    if constexpr (NB == 2) {
#pragma unroll
        for (int c = 0; c < BATCH; ++c) w1[c] = wp[(size_t)(BATCH + c) * 64];
        __builtin_amdgcn_sched_barrier(0);   // keep the second batch's loads ahead of the first MFMAs
    }
    float accv[NACC];
    if constexpr (NT == 32) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const float *arow = As + (lane & 31) * LDA + wave * CPW * 8 + 4 * (lane >> 5);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const float4 wf = (c < BATCH) ? w0[c < BATCH ? c : 0] : w1[(NB == 2 && c >= BATCH) ? c - BATCH : 0];
            const float4 af = *(const float4 *)(arow + c * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, wf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, wf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, wf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, wf.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) accv[i] = acc[i];
    } else {
        // two 16x16 tiles (rows 0-15, 16-31) share each weight fragment; independent accumulators
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float *arow = As + (lane & 15) * LDA + wave * CPW * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const float4 wf = (c < BATCH) ? w0[c < BATCH ? c : 0] : w1[(NB == 2 && c >= BATCH) ? c - BATCH : 0];
            const float4 a0 = *(const float4 *)(arow + c * 16);
            const float4 a1 = *(const float4 *)(arow + 16 * LDA + c * 16);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wf.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wf.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wf.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wf.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wf.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wf.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wf.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wf.w, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            accv[i] = acc0[i];
            accv[4 + i] = acc1[i];
        }
    }
    PD_STAMP(stamps, 4);                  // this wave's MFMA chain issued (its results are awaited by the stores below)
    __syncthreads();   // every wave is done reading As; reuse it for the reduction

    // ---- cross-wave reduction in fixed order + fused epilogue ---------------------------------
    float *red = lds;   // [4][NACC][64]
#pragma unroll
    for (int i = 0; i < NACC; ++i) red[(wave * NACC + i) * 64 + lane] = accv[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int reg = wave * RPW + i;
        float v = red[(0 * NACC + reg) * 64 + lane];
        v += red[(1 * NACC + reg) * 64 + lane];
        v += red[(2 * NACC + reg) * 64 + lane];
        v += red[(3 * NACC + reg) * 64 + lane];
        v += bias;
        int row;
        if constexpr (NT == 32) row = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        else row = m0 + 16 * (reg >> 2) + 4 * (lane >> 4) + (reg & 3);
        if (row < g.M) {
            float *cp = g.C + (size_t)row * g.Nout + col;
            if constexpr (EPI == 1) v = pd_relu(v);
            if constexpr (EPI == 2) v += *cp;
            *cp = v;
        }
    }
    PD_STAMP(stamps, 5);                  // reduced + epilogue issued
    PD_STAMP_DRAIN();
    PD_STAMP(stamps, 6);                  // the stores have left the CU (stamps build only: the wait)
}

// --------------------------------------------------------------------------------------------
// attention core: softmax(q k^T / sqrt(dh)) v for one (sequence, head), N <= 64 frames, no mask
// (nn.MultiheadAttention inside the encoder layer).  grid = (B*heads, ceil(N/4)): every
// workgroup stages K and V of its (sequence, head) and each of its 4 waves owns ONE query row:
// lane j scores key j, softmax is a wave reduction, lanes then own 2 of the 128 output dims.
// --------------------------------------------------------------------------------------------
// SPLIT_OUT: ctx is written as split words {bf16 hi | bf16 lo << 16} for pd_gemm_split (the fast mode)
template <bool SPLIT_OUT>
__global__ __launch_bounds__(256) void pd_attn_kernel(const float *__restrict__ qkv, float *__restrict__ ctx, int N
#ifdef PD_DEN_STAMPS
                                                      , long long *stamps
#endif
) {
    PD_STAMP(stamps, 0);
    constexpr int LD = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Kk = lds, *V = Kk + N * LD, *Q = V + N * LD, *P = Q + 4 * LD;   // P [4][64]
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.y * 4 + wave;        // this wave's query row
    const float scale = 0.08838834764831845f;   // 1/sqrt(128)
    const float *base = qkv + (size_t)b * N * (3 * DM) + h * DH;
    for (int idx = tid; idx < N * (DH / 4); idx += 256) {
        const int j = idx / (DH / 4), d4 = idx % (DH / 4);
        const float *row = base + (size_t)j * (3 * DM) + d4 * 4;
        *(float4 *)(Kk + j * LD + d4 * 4) = *(const float4 *)(row + DM);
        *(float4 *)(V + j * LD + d4 * 4) = *(const float4 *)(row + 2 * DM);
    }
    if (lane < DH / 4) {
        const int ii = i < N ? i : N - 1;
        float4 q = *(const float4 *)(base + (size_t)ii * (3 * DM) + lane * 4);
        q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
        *(float4 *)(Q + wave * LD + lane * 4) = q;
    }
    __syncthreads();
    PD_STAMP(stamps, 2);                  // K, V, Q staged
    const int jj = lane < N ? lane : N - 1;
    const float4 *qa = (const float4 *)(Q + wave * LD), *kb = (const float4 *)(Kk + jj * LD);
    float s = 0.0f;
#pragma unroll 8
    for (int d = 0; d < DH / 4; ++d) {
        const float4 a = qa[d], c = kb[d];
        s = fmaf(a.x, c.x, s);
        s = fmaf(a.y, c.y, s);
        s = fmaf(a.z, c.z, s);
        s = fmaf(a.w, c.w, s);
    }
    const float sv = lane < N ? s : -INFINITY;
    const float mx = pd_wave_max(sv);
    const float e = lane < N ? expf(sv - mx) : 0.0f;
    const float inv = 1.0f / pd_wave_sum(e);
    P[wave * 64 + lane] = e * inv;
    __syncthreads();
    if (i < N) {
        const float *p = P + wave * 64;
        float o0 = 0.0f, o1 = 0.0f;
        for (int j = 0; j < N; ++j) {
            const float pj = p[j];
            o0 = fmaf(pj, V[j * LD + lane], o0);
            o1 = fmaf(pj, V[j * LD + 64 + lane], o1);
        }
        float *out = ctx + (size_t)(b * N + i) * DM + h * DH;
        if constexpr (SPLIT_OUT) {
            ((unsigned *)out)[lane] = pd_split_word(o0);
            ((unsigned *)out)[64 + lane] = pd_split_word(o1);
        } else {
            out[lane] = o0;
            out[64 + lane] = o1;
        }
    }
    PD_STAMP(stamps, 5);
    PD_STAMP_DRAIN();
    PD_STAMP(stamps, 6);
}

// The same attention for large batches: ONE workgroup per (sequence, head) stages K, V and all N query rows once (pd_attn_kernel
// stages K and V ceil(N / 4) times, once per group of four query rows: 5 120 workgroups per layer at the bench shape, 9 - 15 % of the
// denoiser's kernel time for ~1 % of its FLOPs), and every wave works on PD_ATTN_RPW query rows AT ONCE: one K (V) read from LDS serves
// all of them and their serial fmaf chains (128 deep for a score) interleave -- a wave with one row at a time is bound by exactly that
// chain's latency.  Per row the arithmetic is pd_attn_kernel's, operation for operation: the same bits.
#define PD_ATTN_RPW 5
// SPLIT_OUT: 0 fp32, 1 bf16 split words, 2 fp16 split words of ctx * out_scale (pd_split_word_as)
template <int SPLIT_OUT>
__global__ __launch_bounds__(256) void pd_attn_seq_kernel(const float *__restrict__ qkv, float *__restrict__ ctx, int N, float out_scale) {
    constexpr int LD = DH + 4, R = PD_ATTN_RPW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Kk = lds, *V = Kk + N * LD, *Q = V + N * LD, *P = Q + N * LD;   // P [4 waves][R][64]
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float scale = 0.08838834764831845f;   // 1/sqrt(128)
    const float *base = qkv + (size_t)b * N * (3 * DM) + h * DH;
    for (int idx = tid; idx < N * (DH / 4); idx += 256) {
        const int j = idx / (DH / 4), d4 = idx % (DH / 4);
        const float *row = base + (size_t)j * (3 * DM) + d4 * 4;
        float4 q = *(const float4 *)row;
        q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
        *(float4 *)(Q + j * LD + d4 * 4) = q;
        *(float4 *)(Kk + j * LD + d4 * 4) = *(const float4 *)(row + DM);
        *(float4 *)(V + j * LD + d4 * 4) = *(const float4 *)(row + 2 * DM);
    }
    __syncthreads();
    const int jj = lane < N ? lane : N - 1;
    const float4 *kb = (const float4 *)(Kk + jj * LD);
    float *pw = P + wave * (R * 64);
    for (int i0 = 0; i0 < N; i0 += 4 * R) {       // rows i0 + wave * R + t, t < R; every wave takes part in every round (workgroup barriers)
        const int ib = i0 + wave * R;
        const float4 *qa[R];
        float s[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            qa[t] = (const float4 *)(Q + min(ib + t, N - 1) * LD);
            s[t] = 0.0f;
        }
#pragma unroll 4
        for (int d = 0; d < DH / 4; ++d) {
            const float4 c = kb[d];
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const float4 a = qa[t][d];
                s[t] = fmaf(a.x, c.x, s[t]);
                s[t] = fmaf(a.y, c.y, s[t]);
                s[t] = fmaf(a.z, c.z, s[t]);
                s[t] = fmaf(a.w, c.w, s[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const float sv = lane < N ? s[t] : -INFINITY;
            const float mx = pd_wave_max(sv);
            const float e = lane < N ? expf(sv - mx) : 0.0f;
            const float inv = 1.0f / pd_wave_sum(e);
            pw[t * 64 + lane] = e * inv;
        }
        __syncthreads();
        float o0[R], o1[R];
#pragma unroll
        for (int t = 0; t < R; ++t) o0[t] = o1[t] = 0.0f;
        for (int j = 0; j < N; ++j) {
            const float v0 = V[j * LD + lane], v1 = V[j * LD + 64 + lane];
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const float pj = pw[t * 64 + j];
                o0[t] = fmaf(pj, v0, o0[t]);
                o1[t] = fmaf(pj, v1, o1[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int i = ib + t;
            if (i < N) {
                float *out = ctx + (size_t)(b * N + i) * DM + h * DH;
                if constexpr (SPLIT_OUT != 0) {
                    ((unsigned *)out)[lane] = pd_split_word_as<SPLIT_OUT>(o0[t], out_scale);
                    ((unsigned *)out)[64 + lane] = pd_split_word_as<SPLIT_OUT>(o1[t], out_scale);
                } else {
                    out[lane] = o0[t];
                    out[64 + lane] = o1[t];
                }
            }
        }
        __syncthreads();                          // P is rewritten by the next round
    }
}
// The same attention on the matrix pipe, for sequences of <= 32 frames (round 3): pd_attn_seq_kernel keeps 20 of 64 lanes busy in
// its score loop (lane = key) and is compute-bound at 19 - 20 us per layer against a ~10 us floor for moving 31 MB of QKV.  Here
// S = (Q / sqrt(dh)) K^T is four 16 x 16 tiles, one per wave, on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation, K = 128:
// 32 instructions per wave); softmax runs 8 lanes per row over S in LDS (max, expf, sum: the same formulas); O = P V is 2 x 8 tiles of
// 16 x 16, four per wave, K = 32 (32 instructions).  Rows and keys beyond N are zero / masked.  Same mathematics as pd_attn_kernel;
// the sums are MFMA-ordered instead of fmaf chains, so results agree to fp32 rounding (tests/test_gpu_parity_r3.py), not bit for bit.
template <int SPLIT_OUT>
__global__ __launch_bounds__(256) void pd_attn_mma_kernel(const float *__restrict__ qkv, float *__restrict__ ctx, int N, float out_scale) {
    constexpr int LD = DH + 4, LS = 36;
    extern __shared__ __attribute__((aligned(16))) float sm[];                   // Q, K, V: N + 1 rows each (row N is zero: every row / key
    const int NR = N + 1;                                                       // index beyond N reads it), S [32][36]: scores, then probabilities
    float *Q = sm, *Kk = Q + NR * LD, *V = Kk + NR * LD, *S = V + NR * LD;
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float scale = 0.08838834764831845f;   // 1/sqrt(128)
    const float *base = qkv + (size_t)b * N * (3 * DM) + h * DH;
    for (int idx = tid; idx < NR * (DH / 4); idx += 256) {
        const int j = idx / (DH / 4), d4 = idx % (DH / 4);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), k = q, v = q;
        if (j < N) {
            const float *row = base + (size_t)j * (3 * DM) + d4 * 4;
            q = *(const float4 *)row;
            k = *(const float4 *)(row + DM);
            v = *(const float4 *)(row + 2 * DM);
            q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
        }
        *(float4 *)(Q + j * LD + d4 * 4) = q;
        *(float4 *)(Kk + j * LD + d4 * 4) = k;
        *(float4 *)(V + j * LD + d4 * 4) = v;
    }
    __syncthreads();
    {   // scores: wave w owns the tile rows 16 (w >> 1) .., keys 16 (w & 1) ..; lane = (row or key) % 16 + 16 g feeds k = 16 c + 4 g + e
        const float *qa = Q + min(16 * (wave >> 1) + (lane & 15), N) * LD + 4 * (lane >> 4);
        const float *kb = Kk + min(16 * (wave & 1) + (lane & 15), N) * LD + 4 * (lane >> 4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DH / 16; ++c) {
            const float4 a = *(const float4 *)(qa + 16 * c), k = *(const float4 *)(kb + 16 * c);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, k.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, k.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, k.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, k.w, acc, 0, 0, 0);
        }
        const int j = 16 * (wave & 1) + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(16 * (wave >> 1) + 4 * (lane >> 4) + e) * LS + j] = acc[e];
    }
    __syncthreads();
    {   // softmax: 8 lanes per row, 4 keys per lane
        const int i = tid >> 3, sub = tid & 7;
        float4 sv = *(const float4 *)(S + i * LS + 4 * sub);
        const int j0 = 4 * sub;
        sv.x = j0 + 0 < N ? sv.x : -INFINITY;
        sv.y = j0 + 1 < N ? sv.y : -INFINITY;
        sv.z = j0 + 2 < N ? sv.z : -INFINITY;
        sv.w = j0 + 3 < N ? sv.w : -INFINITY;
        float mx = fmaxf(fmaxf(sv.x, sv.y), fmaxf(sv.z, sv.w));
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        float4 e;
        e.x = j0 + 0 < N ? expf(sv.x - mx) : 0.0f;
        e.y = j0 + 1 < N ? expf(sv.y - mx) : 0.0f;
        e.z = j0 + 2 < N ? expf(sv.z - mx) : 0.0f;
        e.w = j0 + 3 < N ? expf(sv.w - mx) : 0.0f;
        const float inv = 1.0f / pd_sum8((e.x + e.y) + (e.z + e.w));
        e.x *= inv; e.y *= inv; e.z *= inv; e.w *= inv;
        *(float4 *)(S + i * LS + 4 * sub) = e;
    }
    __syncthreads();
    {   // O = P V: wave w owns the output columns [32 w, 32 w + 32) (two tiles) of both row tiles; k = key j = 16 c + 4 g + e
        f32x4 acc[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *pa = S + (lane & 15) * LS + 4 * (lane >> 4);
        const float *vb = V + 32 * wave + (lane & 15);
        const int jg = 4 * (lane >> 4);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 p0 = *(const float4 *)(pa + 16 * c), p1 = *(const float4 *)(pa + 16 * LS + 16 * c);
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = vb[min(16 * c + jg + e, N) * LD];
                v1[e] = vb[min(16 * c + jg + e, N) * LD + 16];
            }
            const float a0[4] = {p0.x, p0.y, p0.z, p0.w}, a1[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], v0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], v1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], v0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], v1[e], acc[1][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 16 * rt + 4 * (lane >> 4) + e;
                if (i < N) {
                    float *out = ctx + (size_t)(b * N + i) * DM + h * DH + 32 * wave + (lane & 15);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if constexpr (SPLIT_OUT != 0) ((unsigned *)out)[16 * ct] = pd_split_word_as<SPLIT_OUT>(acc[rt][ct][e], out_scale);
                        else out[16 * ct] = acc[rt][ct][e];
                    }
                }
            }
    }
}
static size_t attn_mma_lds(int N) { return ((size_t)3 * (N + 1) * (DH + 4) + 32 * 36) * sizeof(float); }
static size_t attn_seq_lds(int N) { return ((size_t)3 * N * (DH + 4) + 4 * PD_ATTN_RPW * 64) * sizeof(float); }

// --------------------------------------------------------------------------------------------
// tail of the head: LayerNorm(128) -> ReLU -> Linear(128 -> 9) (denoiser.py:51,74 `_last.1..3`)
// fused with predict_start_from_noise / q_posterior / the sample update
// (gaussian_diffuser.py:190-209, :280).  One wave per token; lane holds 2 of the 128 hidden values.
// --------------------------------------------------------------------------------------------
struct HeadArgs {
    const float *hid;      // [M, 128] = _last.0 output (bias included)
    const float *lnw, *lnb, *w3, *b3;
    const float *x;        // [M, 9] current sample
    const float *noise;    // [M, 9] or null
    float *eps_out, *mean_out, *x0_out, *xnext_out;   // each [M, 9] or null
    float c_recip, c_recipm1, coef1, coef2, sigma;
    int M;
    int pred_x0;           // objective "pred_x0": the model output is x_start (gaussian_diffuser.py:225-227)
#ifdef PD_DEN_STAMPS
    long long *stamps;
#endif
};

__global__ __launch_bounds__(256) void pd_tail_kernel(HeadArgs g) {
#ifdef PD_DEN_STAMPS
    long long *const stamps = g.stamps;
#endif
    PD_STAMP(stamps, 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= g.M) return;
    const float *row = g.hid + (size_t)m * HID;
    const float v0 = row[lane], v1 = row[64 + lane];
    // everything the last nine lanes add at the end is requested now (clamped lane: no predicated loads), not behind the reductions
    const int l9 = lane < 9 ? lane : 8;
    const size_t at = (size_t)m * 9 + l9;
    const float b3v = g.b3[l9], xv = g.x[at], nz = g.noise ? g.noise[at] : 0.0f;
    const float mean = pd_wave_sum(v0 + v1) * (1.0f / HID);
    const float d0 = v0 - mean, d1 = v1 - mean;
    const float rstd = 1.0f / sqrtf(pd_wave_sum(d0 * d0 + d1 * d1) * (1.0f / HID) + 1e-5f);
    const float a0 = pd_relu(d0 * rstd * g.lnw[lane] + g.lnb[lane]);
    const float a1 = pd_relu(d1 * rstd * g.lnw[64 + lane] + g.lnb[64 + lane]);
    float e = 0.0f;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const float part = pd_wave_sum(fmaf(a0, g.w3[o * HID + lane], a1 * g.w3[o * HID + 64 + lane]));
        e = (lane == o) ? part : e;
    }
    if (lane < 9) {
        e += b3v;
        const float x0 = g.pred_x0 ? e : g.c_recip * xv - g.c_recipm1 * e;   // gaussian_diffuser.py:190-194, :221-227
        const float mu = g.coef1 * x0 + g.coef2 * xv;               // :201-205
        if (g.eps_out) g.eps_out[at] = e;
        if (g.x0_out) g.x0_out[at] = x0;
        if (g.mean_out) g.mean_out[at] = mu;
        if (g.xnext_out) g.xnext_out[at] = g.noise ? mu + g.sigma * nz : mu;   // :280
    }
    PD_STAMP(stamps, 5);
    PD_STAMP_DRAIN();
    PD_STAMP(stamps, 6);
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
static int dev_alloc(PdDenoiserDev *d, float **p, size_t n_floats) {
    PD_HIP_CHECK(hipMalloc((void **)p, n_floats * sizeof(float)));
    d->allocs.push_back(*p);
    return PD_OK;
}
static int dev_copy(PdDenoiserDev *d, float **dst, const float *src, size_t n) {
    if (!src) {
        pd_set_error("pd_engine_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    int rc = dev_alloc(d, dst, n);
    if (rc) return rc;
    PD_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    return PD_OK;
}
// pack W for tile width nt (32 or 16); gamma (nullable) is folded in as a column scale
static int dev_pack(PdDenoiserDev *d, float **dst, const float *W, int Nout, int K, int Kpad, int nt, int first_perm = 0,
                    const float *gamma = nullptr) {
    if (!W) {
        pd_set_error("pd_engine_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    const int NT = (Nout + nt - 1) / nt, KC = Kpad / (nt == 32 ? 8 : 16);
    const size_t total = (size_t)NT * KC * 256;
    int rc = dev_alloc(d, dst, total);
    if (rc) return rc;
    hipLaunchKernelGGL(pd_repack_kernel, dim3(512), dim3(256), 0, 0, W, Nout, K, KC, *dst, total, first_perm, nt, gamma);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
// row-major copy with gamma (nullable) folded in as a column scale, for pd_gemm_stream
static int dev_rowmajor(PdDenoiserDev *d, float **dst, const float *W, int Nout, int K, const float *gamma) {
    if (!W) {
        pd_set_error("pd_engine_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    const size_t total = (size_t)Nout * K;
    int rc = dev_alloc(d, dst, total);
    if (rc) return rc;
    hipLaunchKernelGGL(pd_scale_cols_kernel, dim3(512), dim3(256), 0, 0, W, gamma, K, total, *dst);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
// b' = b + W beta
static int dev_fold_bias(PdDenoiserDev *d, float **dst, const float *W, const float *beta, const float *b, int Nout, int K) {
    if (!W || !beta || !b) {
        pd_set_error("pd_engine_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    int rc = dev_alloc(d, dst, Nout);
    if (rc) return rc;
    hipLaunchKernelGGL(pd_fold_bias_kernel, dim3((Nout + 127) / 128), dim3(128), 0, 0, W, beta, b, Nout, K, *dst);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

#define PD_TRY(expr)        \
    do {                    \
        int _rc = (expr);   \
        if (_rc) return _rc; \
    } while (0)

template <typename KernelT>
static int set_lds(KernelT kern, size_t bytes) {
    PD_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PD_OK;
}

int pd_denoiser_create(pd_engine *eng, const pd_weights *w) {
    if (w->d_model != DM || w->nhead != NH || w->dim_ff != DFF || w->z_dim != ZD || w->n_harmonic != 10 ||
        w->t_emb_dim != 256 || w->mlp_hidden != HID || w->num_layers < 1 || w->num_layers > PD_MAX_LAYERS) {
        pd_set_error("pd_engine_create: unsupported denoiser shape (built for d_model=512 nhead=4 ff=1024 z=384 "
                     "harmonics=10 t_emb=256 hidden=128, 1..%d layers)", PD_MAX_LAYERS);
        return PD_ERR_UNSUPPORTED;
    }
    PdDenoiserDev *d = new PdDenoiserDev();
    eng->den = d;
    d->num_layers = w->num_layers;
    d->timesteps = w->timesteps;
    d->m_cap = ((eng->max_B * eng->max_N + 31) / 32) * 32;
    // time embedding table
    {
        float *w0, *b0, *w2, *b2;
        PD_TRY(dev_copy(d, &w0, w->time_w0, 128 * 256));
        PD_TRY(dev_copy(d, &b0, w->time_b0, 128));
        PD_TRY(dev_copy(d, &w2, w->time_w2, 128 * 128));
        PD_TRY(dev_copy(d, &b2, w->time_b2, 128));
        PD_TRY(dev_alloc(d, &d->t_table, (size_t)w->timesteps * 128));
        hipLaunchKernelGGL(pd_time_table_kernel, dim3(w->timesteps), dim3(128), 0, 0, w0, b0, w2, b2, d->t_table);
        PD_HIP_CHECK(hipGetLastError());
    }
    for (int v = 0; v < 2; ++v) {   // v = 0: 32-wide tiles, v = 1: 16-wide tiles
        const int nt = v ? 16 : 32;
        PD_TRY(dev_pack(d, &d->first_wp[v], w->first_w, DM, KFIRST, KFIRST_PAD, nt, 1));
        PD_TRY(dev_pack(d, &d->last0_wp[v], w->last0_w, HID, DM, DM, nt));
        for (int l = 0; l < w->num_layers; ++l) {
            const pd_layer_weights &s = w->layers[l];
            PdLayerDev &L = d->layers[l];
            // LayerNorm affine folded: W' = W diag(gamma) (packed), b' = b + W beta
            PD_TRY(dev_pack(d, &L.qkv_wp[v], s.in_proj_w, 3 * DM, DM, DM, nt, 0, s.norm1_w));
            PD_TRY(dev_pack(d, &L.out_wp[v], s.out_proj_w, DM, DM, DM, nt));
            PD_TRY(dev_pack(d, &L.ff1_wp[v], s.linear1_w, DFF, DM, DM, nt, 0, s.norm2_w));
            PD_TRY(dev_pack(d, &L.ff2_wp[v], s.linear2_w, DM, DFF, DFF, nt));
        }
    }
    PD_TRY(dev_copy(d, &d->first_b, w->first_b, DM));
    for (int l = 0; l < w->num_layers; ++l) {
        const pd_layer_weights &s = w->layers[l];
        PdLayerDev &L = d->layers[l];
        if (!s.norm1_w || !s.norm2_w) {
            pd_set_error("pd_engine_create: a LayerNorm weight pointer is NULL");
            return PD_ERR_INVALID_ARG;
        }
        PD_TRY(dev_fold_bias(d, &L.qkv_b, s.in_proj_w, s.norm1_b, s.in_proj_b, 3 * DM, DM));
        PD_TRY(dev_copy(d, &L.out_b, s.out_proj_b, DM));
        PD_TRY(dev_fold_bias(d, &L.ff1_b, s.linear1_w, s.norm2_b, s.linear1_b, DFF, DM));
        PD_TRY(dev_copy(d, &L.ff2_b, s.linear2_b, DM));
        PD_TRY(dev_rowmajor(d, &L.qkv_wf, s.in_proj_w, 3 * DM, DM, s.norm1_w));
        PD_TRY(dev_rowmajor(d, &L.out_wf, s.out_proj_w, DM, DM, nullptr));
        PD_TRY(dev_rowmajor(d, &L.ff1_wf, s.linear1_w, DFF, DM, s.norm2_w));
        PD_TRY(dev_rowmajor(d, &L.ff2_wf, s.linear2_w, DM, DFF, nullptr));
    }
    PD_TRY(dev_copy(d, &d->last0_b, w->last0_b, HID));
    PD_TRY(dev_copy(d, &d->last_ln_w, w->last_ln_w, HID));
    PD_TRY(dev_copy(d, &d->last_ln_b, w->last_ln_b, HID));
    PD_TRY(dev_copy(d, &d->last3_w, w->last3_w, 9 * HID));
    PD_TRY(dev_copy(d, &d->last3_b, w->last3_b, 9));
    const size_t rows = (size_t)d->m_cap;
    PD_TRY(dev_alloc(d, &d->h, rows * DM));
    PD_TRY(dev_alloc(d, &d->qkv, rows * 3 * DM));
    PD_TRY(dev_alloc(d, &d->ctx, rows * DM));
    PD_TRY(dev_alloc(d, &d->ff, rows * DFF));
    PD_TRY(dev_alloc(d, &d->hid, rows * HID));
    if (rows >= PD_STREAM_MIN_ROWS) PD_TRY(dev_alloc(d, &d->hn, rows * DM));
    // _first's input rows (materialised by pd_embed_rows_kernel on the streamed path, formerly also by the parked persistent kernel) and the
    // row-major _first / _last.0 the two paths pack from
    if (d->hn) {      // the streamed path evaluates _first in three pieces (pd_denoiser_dev.h): two of them outside the diffusion steps
        PD_TRY(dev_alloc(d, &d->emb, rows * KFIRST_D));
        PD_TRY(dev_alloc(d, &d->zproj, rows * DM));
        PD_TRY(dev_alloc(d, &d->first_df, (size_t)DM * KFIRST_D));
        PD_TRY(dev_alloc(d, &d->first_zf, (size_t)DM * ZD));
        hipLaunchKernelGGL(pd_first_rowmajor_kernel, dim3((DM * KFIRST_D + 255) / 256), dim3(256), 0, 0, w->first_w, d->first_df, PD_FIRST_D, KFIRST_D);
        hipLaunchKernelGGL(pd_first_rowmajor_kernel, dim3((DM * ZD + 255) / 256), dim3(256), 0, 0, w->first_w, d->first_zf, PD_FIRST_Z, ZD);
        PD_HIP_CHECK(hipGetLastError());
        // the time piece of _first for every step: ttab[t] = W_t t_emb(t)
        PD_TRY(dev_alloc(d, &d->ttab, (size_t)w->timesteps * DM));
        hipLaunchKernelGGL(pd_first_ttab_kernel, dim3(w->timesteps), dim3(DM), 0, 0, w->first_w, d->t_table, d->ttab);
        PD_HIP_CHECK(hipGetLastError());
    }
    PD_TRY(dev_rowmajor(d, &d->last0_wf, w->last0_w, HID, DM, nullptr));
    PD_TRY(set_lds(pd_gemm_kernel<KFIRST_PAD, 2, 0, 32>, 32 * (KFIRST_PAD + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<KFIRST_PAD, 2, 0, 16>, 32 * (KFIRST_PAD + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 1, 0, 32>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 1, 0, 16>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 1, 1, 32>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 1, 1, 16>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 0, 2, 32>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 0, 2, 16>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DFF, 0, 2, 32>, 32 * (DFF + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DFF, 0, 2, 16>, 32 * (DFF + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 0, 0, 32>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 0, 0, 16>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_attn_kernel<false>, ((2 * 64 + 4) * (DH + 4) + 4 * 64) * 4));
    PD_TRY(set_lds(pd_attn_kernel<true>, ((2 * 64 + 4) * (DH + 4) + 4 * 64) * 4));
    PD_TRY(set_lds(pd_attn_seq_kernel<0>, attn_seq_lds(64)));
    PD_TRY(set_lds(pd_attn_seq_kernel<1>, attn_seq_lds(64)));
    PD_TRY(set_lds(pd_attn_seq_kernel<2>, attn_seq_lds(64)));
    PD_TRY(set_lds(pd_attn_mma_kernel<2>, attn_mma_lds(32)));
    PD_TRY(set_lds((pd_qkv_attn_kernel<0, PD_QA_DEEP_DEFAULT != 0>), 160 * 1024));
#ifdef PD_DEV_KNOBS
    PD_TRY(set_lds((pd_qkv_attn_kernel<0, PD_QA_DEEP_DEFAULT == 0>), 160 * 1024));
    PD_TRY(set_lds(pd_qkv_attn_kernel<1>, 160 * 1024));
    PD_TRY(set_lds(pd_qkv_attn_kernel<2>, 160 * 1024));
    PD_TRY(set_lds(pd_qkv_attn_kernel<3>, 160 * 1024));
    PD_TRY(set_lds(pd_qkv_attn_kernel<4>, 160 * 1024));
    PD_TRY(set_lds(pd_qkv_attn_kernel<5>, 160 * 1024));
#endif
    PD_HIP_CHECK(hipDeviceSynchronize());
    return PD_OK;
}

// The fast mode's weights: every encoder Linear split into bf16 hi / lo in MFMA fragment order (LayerNorm scale folded as for
// the other packings).  Built when the mode is first switched on, from the row-major fp32 copies kept for the streamed GEMMs.
// mode 2, fp16 planes: fp16 keeps 11 bits and five exponent bits, so every operand gets a POWER-OF-TWO scale (exact to apply and
// to undo) fixed here from bounds that hold for every input:
//   * LayerNorm output without affine: sum of squares <= D, so |x^| <= sqrt(D) = 22.6;           scale 2^9  (<= 11 585)
//   * a Linear fed by it:  |x^ . w + b| <= sqrt(D) ||w||_2 + |b|   (Cauchy-Schwarz) -- the V rows the attention averages
//     (a convex combination: same bound) and the FF hidden rows after ReLU;                       scale 2^floor(log2(32768 / bound))
//   * weights: 2^floor(log2(16384 / max |w|)).
// Nothing can overflow (fp16 max 65 504), and hi + lo keeps 22 bits for every value above 2^-18 of its bound.
static int floor_log2_ratio(double cap, double v) {       // v finite (pd_denoiser_build_scales rejects anything else)
    if (!(v > 0.0)) return 0;
    double e = floor(log2(cap / v));                       // clamped as a double: the cast below is always defined
    e = e < -60.0 ? -60.0 : (e > 60.0 ? 60.0 : e);
    return (int)e;
}
static int pd_denoiser_build_scales(pd_engine *eng) {
    PdDenoiserDev *d = eng->den;
    if (d->scales_ready) return PD_OK;
    std::vector<float> w, b;
    auto fetch = [&](const float *Wf, const float *bias, int Nout, int K) -> int {
        w.resize((size_t)Nout * K);
        b.resize(Nout);
        PD_HIP_CHECK(hipMemcpy(w.data(), Wf, w.size() * sizeof(float), hipMemcpyDeviceToHost));
        PD_HIP_CHECK(hipMemcpy(b.data(), bias, b.size() * sizeof(float), hipMemcpyDeviceToHost));
        return PD_OK;
    };
    // a checkpoint with inf / NaN has no static bound: the fp16-plane mode is refused (PD_ERR_INVALID_ARG) and the engine stays on
    // the exact-fp32 kernels, which propagate the values like the reference does
    bool finite = true;
    auto max_abs = [&]() { double m = 0; for (float v : w) { finite = finite && isfinite(v); m = fmax(m, fabs((double)v)); } return m; };
    auto row_bound = [&](int r0, int r1, int K) {            // max over rows of sqrt(D) ||w_r||_2 + |b_r|
        double bound = 0;
        for (int r = r0; r < r1; ++r) {
            double q = 0;
            for (int k = 0; k < K; ++k) q += (double)w[(size_t)r * K + k] * w[(size_t)r * K + k];
            bound = fmax(bound, sqrt((double)DM) * sqrt(q) + fabs((double)b[r]));
            finite = finite && isfinite(q) && isfinite(b[r]);
        }
        return bound;
    };
    const int e_ln = 9;
    PD_HIP_CHECK(hipDeviceSynchronize());
    for (int l = 0; l < d->num_layers; ++l) {
        PdLayerDev &L = d->layers[l];
        PD_TRY(fetch(L.qkv_wf, L.qkv_b, 3 * DM, DM));
        const int e_ctx = floor_log2_ratio(32768.0, row_bound(2 * DM, 3 * DM, DM));
        L.e_wqkv = floor_log2_ratio(16384.0, max_abs());
        PD_TRY(fetch(L.out_wf, L.out_b, DM, DM));
        L.e_wo = floor_log2_ratio(16384.0, max_abs());
        PD_TRY(fetch(L.ff1_wf, L.ff1_b, DFF, DM));
        const int e_ff = floor_log2_ratio(32768.0, row_bound(0, DFF, DM));
        L.e_w1 = floor_log2_ratio(16384.0, max_abs());
        PD_TRY(fetch(L.ff2_wf, L.ff2_b, DM, DFF));
        L.e_w2 = floor_log2_ratio(16384.0, max_abs());
        L.qkv_cs = ldexpf(1.0f, -(e_ln + L.e_wqkv));
        L.out_cs = ldexpf(1.0f, -(e_ctx + L.e_wo));
        L.ff1_cs = ldexpf(1.0f, -(e_ln + L.e_w1));
        L.ff2_cs = ldexpf(1.0f, -(e_ff + L.e_w2));
        L.ctx_scale = ldexpf(1.0f, e_ctx);
        L.ff_scale = ldexpf(1.0f, e_ff);
        if (!finite) {
            d->non_finite = true;
            pd_set_error("denoiser: encoder layer %d holds non-finite weights or biases: no static operand bound exists, the fp16-plane "
                         "mode (PD_OPT_DENOISER_SPLIT = 2) is not available for these weights", l);
            return PD_ERR_INVALID_ARG;
        }
    }
    d->scales_ready = true;
    return PD_OK;
}
static int pd_denoiser_build_split_h(pd_engine *eng) {
    PdDenoiserDev *d = eng->den;
    if (d->split_h_ready) return PD_OK;
    PD_TRY(pd_denoiser_build_scales(eng));
    auto split = [&](unsigned **dst, const float *Wf, int Nout, int K, int ew) -> int {
        float *p = nullptr;
        int rc = dev_alloc(d, &p, (size_t)Nout * K);
        if (rc) return rc;
        *dst = (unsigned *)p;
        const size_t total = (size_t)(Nout / 32) * (K / 16) * 64;
        hipLaunchKernelGGL(vit_frag_split_kernel, dim3(512), dim3(256), 0, 0, Wf, (const float *)nullptr, K, total, (uint4 *)p, 1, ldexpf(1.0f, ew));
        PD_HIP_CHECK(hipGetLastError());
        return PD_OK;
    };
    for (int l = 0; l < d->num_layers; ++l) {
        PdLayerDev &L = d->layers[l];
        PD_TRY(split(&L.qkv_wh, L.qkv_wf, 3 * DM, DM, L.e_wqkv));
        PD_TRY(split(&L.out_wh, L.out_wf, DM, DM, L.e_wo));
        PD_TRY(split(&L.ff1_wh, L.ff1_wf, DFF, DM, L.e_w1));
        PD_TRY(split(&L.ff2_wh, L.ff2_wf, DM, DFF, L.e_w2));
    }
    PD_HIP_CHECK(hipDeviceSynchronize());
    d->split_h_ready = true;
    return PD_OK;
}
bool pd_denoiser_has_streamed_path(const pd_engine *eng) { return eng->den && eng->den->hn; }
bool pd_denoiser_weights_non_finite(const pd_engine *eng) { return eng->den && eng->den->non_finite; }

int pd_denoiser_build_split(pd_engine *eng, int mode) {
    PdDenoiserDev *d = eng->den;
    if (!d->hn) {
        pd_set_error("split-precision denoiser: the engine was created for fewer than %d token rows (max_B x max_N); the mode "
                     "applies to the streamed large-batch path only", PD_STREAM_MIN_ROWS);
        return PD_ERR_UNSUPPORTED;
    }
    if (mode == 2) return pd_denoiser_build_split_h(eng);
    if (d->split_ready) return PD_OK;
    if (!d->hn) {
        pd_set_error("split-precision denoiser: the engine was created for fewer than %d token rows (max_B x max_N); the mode "
                     "applies to the streamed large-batch path only", PD_STREAM_MIN_ROWS);
        return PD_ERR_UNSUPPORTED;
    }
    auto split = [&](unsigned **dst, const float *Wf, int Nout, int K) -> int {
        float *p = nullptr;
        int rc = dev_alloc(d, &p, (size_t)Nout * K);
        if (rc) return rc;
        *dst = (unsigned *)p;
        const size_t total = (size_t)(Nout / 32) * (K / 16) * 64;
        hipLaunchKernelGGL(vit_frag_split_kernel, dim3(512), dim3(256), 0, 0, Wf, (const float *)nullptr, K, total, (uint4 *)p);   // gamma is already folded into Wf
        PD_HIP_CHECK(hipGetLastError());
        return PD_OK;
    };
    for (int l = 0; l < d->num_layers; ++l) {
        PdLayerDev &L = d->layers[l];
        PD_TRY(split(&L.qkv_ws, L.qkv_wf, 3 * DM, DM));
        PD_TRY(split(&L.out_ws, L.out_wf, DM, DM));
        PD_TRY(split(&L.ff1_ws, L.ff1_wf, DFF, DM));
        PD_TRY(split(&L.ff2_ws, L.ff2_wf, DM, DFF));
    }
    PD_HIP_CHECK(hipDeviceSynchronize());
    d->split_ready = true;
    return PD_OK;
}

void pd_denoiser_destroy(pd_engine *eng) {
    if (!eng->den) return;
    for (void *p : eng->den->allocs) (void)hipFree(p);
    delete eng->den;
    eng->den = nullptr;
}

// one GEMM launch; the tile width is chosen per problem: 16-wide tiles double the workgroup count (and
// halve each wave's serial MFMA chain) whenever 32-wide tiles would leave most of the 256 CUs idle
#ifdef PD_DEN_STAMPS
static long long *g_den_stamps = nullptr;      // [PD_DEN_STAMP_SLOTS][8], device memory; the slot of the next small-batch launch
static int g_den_stamp_slot = 0;
#define PD_DEN_STAMP_SLOTS 256
static long long *next_stamp_slot() {
    if (!g_den_stamps) {
        if (hipMalloc((void **)&g_den_stamps, sizeof(long long) * 8 * PD_DEN_STAMP_SLOTS) != hipSuccess) return nullptr;
        (void)hipMemset(g_den_stamps, 0, sizeof(long long) * 8 * PD_DEN_STAMP_SLOTS);
    }
    long long *p = g_den_stamps + 8 * (g_den_stamp_slot % PD_DEN_STAMP_SLOTS);
    g_den_stamp_slot += 1;
    return p;
}
// out[n_slots][8]: the stamps of the last launches (slot = launch index mod 256); restarts the slot counter
extern "C" int pd_debug_den_stamps(long long *out, int n_slots) {
    if (!out || n_slots <= 0 || n_slots > PD_DEN_STAMP_SLOTS || !g_den_stamps) return PD_ERR_INVALID_ARG;
    PD_HIP_CHECK(hipDeviceSynchronize());
    PD_HIP_CHECK(hipMemcpy(out, g_den_stamps, sizeof(long long) * 8 * n_slots, hipMemcpyDeviceToHost));
    PD_HIP_CHECK(hipMemset(g_den_stamps, 0, sizeof(long long) * 8 * PD_DEN_STAMP_SLOTS));
    g_den_stamp_slot = 0;
    return PD_OK;
}
#endif
template <int K, int AMODE, int EPI>
static void launch_gemm(GemmArgs &g, float *const wp[2], int MT, int wide_min, hipStream_t s) {
    const int tiles32 = MT * (g.Nout / 32);
    g.MT = MT;
#ifdef PD_DEN_STAMPS
    g.stamps = next_stamp_slot();
#endif
    // the XCD-aware block mapping of pd_gemm_kernel needs a multiple of 8 N-tiles (128-wide _last.0 has only 4 of 32)
    if (tiles32 >= wide_min && (g.Nout / 32) % 8 == 0) {
        g.Wp = wp[0];
        hipLaunchKernelGGL((pd_gemm_kernel<K, AMODE, EPI, 32>), dim3(MT * (g.Nout / 32)), dim3(256), 32 * (K + 4) * 4, s, g);
    } else {
        g.Wp = wp[1];
        hipLaunchKernelGGL((pd_gemm_kernel<K, AMODE, EPI, 16>), dim3(MT * (g.Nout / 16)), dim3(256), 32 * (K + 4) * 4, s, g);
    }
}

// The step-invariant piece of _first (models/denoiser.py:56-70: z and the pivot flag do not change over the T steps; here the z columns):
// zproj[m] = z[m] W_z^T + b_first, once per sampling call.  Every step then adds its row of the time table and its 192-column GEMM.
int pd_denoiser_prepare(pd_engine *eng, const float *z, int B, int N, hipStream_t s) {
    PdDenoiserDev *d = eng->den;
    if (!z || B <= 0 || N <= 0 || B > eng->max_B || N > eng->max_N) {
        pd_set_error("denoiser: invalid arguments (B=%d N=%d; max_B=%d max_N=%d)", B, N, eng->max_B, eng->max_N);
        return PD_ERR_INVALID_ARG;
    }
    const int M = B * N;
    // (below PD_STREAM_MIN_ROWS token rows _first stays ONE fused launch -- embedding staged in the GEMM's A rows, K = 704: a step there is a
    // chain of 43 latency-bound launches in which the shorter K buys 1 %, and the small-batch results stay bitwise those of rounds 1-4)
    if (M >= PD_STREAM_MIN_ROWS && d->hn) pd_gemm_dma<0>(z, ZD, d->first_zf, ZD, d->first_b, d->zproj, M, DM, s);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

// z_prepared: pd_denoiser_prepare ran for this z (the sampling loop calls it once); otherwise it is issued here (the step-level API)
int pd_denoiser_launch(pd_engine *eng, const float *x, const float *z, int t, int B, int N, float *eps_out,
                       float *mean_out, float *x0_out, const float *noise, float *x_next_out, hipStream_t s, bool z_prepared) {
    PdDenoiserDev *d = eng->den;
    if (!x || !z || B <= 0 || N <= 0 || B > eng->max_B || N > eng->max_N || N > 64 || t < 0 || t >= d->timesteps) {
        pd_set_error("denoiser: invalid arguments (B=%d N=%d t=%d; max_B=%d max_N=%d, N <= 64, 0 <= t < %d)", B, N, t,
                     eng->max_B, eng->max_N, d->timesteps);
        return PD_ERR_INVALID_ARG;
    }
    if (!z_prepared) {
        int rc = pd_denoiser_prepare(eng, z, B, N, s);
        if (rc) return rc;
    }
    const int M = B * N, MT = (M + 31) / 32;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = M;
    const bool streamed = M >= PD_STREAM_MIN_ROWS && d->hn;
    if (streamed) {
        // _first = zproj (z piece + bias, hoisted) + ttab[t] (time piece, a table) + the step piece, K = 192
        hipLaunchKernelGGL(pd_embed_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, N, M, d->emb);
        pd_gemm_dma<4>(d->emb, KFIRST_D, d->first_df, KFIRST_D, d->ttab + (size_t)t * DM, d->h, M, DM, s, nullptr, d->zproj);
    } else {
        // _first with the embedding fused into the A staging
        g.bias = d->first_b; g.C = d->h; g.Nout = DM;
        g.x = x; g.z = z; g.temb = d->t_table + (size_t)t * 128; g.n_frames = N;
        launch_gemm<KFIRST_PAD, 2, 0>(g, d->first_wp, MT, eng->gemm_wide_min_tiles, s);
    }
    // >= 1024 token rows (52 sequences of 20 frames): the encoder GEMMs are large enough for 64 x 64 tiles streamed through LDS
    // (pd_gemm_stream.h; same sums in another order than the 32-row split-K tiles below, i.e. rounding-level differences
    // between small and large batches).  LayerNorm is fused into the A staging there too (pre-pass per workgroup; affine folded
    // into the weights, as below).
    for (int l = 0; l < d->num_layers; ++l) {
        const PdLayerDev &L = d->layers[l];
        if (streamed && eng->den_split == 2 && d->split_h_ready) {
            // fp16-plane mode: the fast mode's kernels with fp16 halves and the static power-of-two scales of pd_denoiser_build_split_h
            // (22 mantissa bits, fp32 accumulation: fp32-grade results at the three-product rate)
            // GEMM kernel per shape: the strip kernel (A by LDS-DMA, no weight fragment fetched twice) or the round-1 two-plane kernel;
            // bitwise the same results (tools/split3_probe.hip).  PD_DEN_STRIP = bit mask {QKV, out, FF1, FF2} (development A / B)
            static const int strip = pd_dev_knob("PD_DEN_STRIP", 15);
            hipLaunchKernelGGL((pd_ln_rows_kernel<DM, 2>), dim3((M + 3) / 4), dim3(256), 0, s, d->h, d->hn, M, 1e-5f, 512.0f);
            static const int attn_mma = pd_dev_knob("PD_DEN_ATTN_MMA", 1);     // development A / B
            // the fused kernel holds a CU for ~38 us whatever the batch (one workgroup per 4 sequences and head): it wins when its workgroups
            // fill the chip's rounds (256 sequences = 256 workgroups: -76 us per step), not at 103 sequences (104 workgroups: +2 %)
            const int qa_wgs = ((B + pd_qkv_attn_group(N > 32 ? 1 : N) - 1) / pd_qkv_attn_group(N > 32 ? 1 : N)) * NH, cus = eng->num_cus > 0 ? eng->num_cus : 256;
            const bool qa_fills = 4 * qa_wgs >= 3 * ((qa_wgs + cus - 1) / cus) * cus;
            if (N <= 32 && eng->den_fused_attn == 1 ? qa_fills : (N <= 32 && eng->den_fused_attn == 2)) {
                // in_proj + attention of a head for a group of whole sequences in one workgroup, Q / K / V in LDS only (pd_qkv_attn.h):
                // bitwise the two launches of the else branch
                pd_qkv_attn((const unsigned *)d->hn, L.qkv_wh, L.qkv_b, (unsigned *)d->ctx, B, N, L.qkv_cs, L.ctx_scale, s);
            } else {
                if (strip & 1) pd_gemm_strip<0, 2, true, 1, PD_STRIP_K64>((const unsigned *)d->hn, DM, L.qkv_wh, DM, L.qkv_b, d->qkv, M, 3 * DM, s, L.qkv_cs);
                else pd_gemm_split<0, 1, 2, true>((const unsigned *)d->hn, DM, L.qkv_wh, DM, L.qkv_b, d->qkv, M, 3 * DM, s, L.qkv_cs);
                if (N <= 32 && attn_mma) hipLaunchKernelGGL(pd_attn_mma_kernel<2>, dim3(B * NH), dim3(256), attn_mma_lds(N), s, d->qkv, d->ctx, N, L.ctx_scale);
                else hipLaunchKernelGGL(pd_attn_seq_kernel<2>, dim3(B * NH), dim3(256), attn_seq_lds(N), s, d->qkv, d->ctx, N, L.ctx_scale);
            }
            // 512-wide outputs: 96-row tiles where 64-row tiles would give the busiest CUs two tiles and most CUs one (5 120 rows: 320 tiles on 256 CUs ->
            // 216 tiles of 1.5 x the work: the launch is as long as its busiest CU).  Same sums in the same order: bitwise the same C.
            const bool rt3 = PD_STRIP_RT3 && (((M + 63) / 64) * (DM / 128)) > cus && (((M + 95) / 96) * (DM / 128)) <= cus;
#if PD_STRIP_RT1
            if ((strip & 2) && rt3) pd_gemm_strip<2, 1, true, 1, PD_STRIP_K64>((const unsigned *)d->ctx, DM, L.out_wh, DM, L.out_b, d->h, M, DM, s, L.out_cs);
            else
#endif
            if ((strip & 2) && rt3) pd_gemm_strip<2, 3, true, 1, PD_STRIP_K64>((const unsigned *)d->ctx, DM, L.out_wh, DM, L.out_b, d->h, M, DM, s, L.out_cs);
            else if (strip & 2) pd_gemm_strip<2, 2, true, 1, PD_STRIP_K64>((const unsigned *)d->ctx, DM, L.out_wh, DM, L.out_b, d->h, M, DM, s, L.out_cs);
            else pd_gemm_split<2, 1, 1, true>((const unsigned *)d->ctx, DM, L.out_wh, DM, L.out_b, d->h, M, DM, s, L.out_cs);
            hipLaunchKernelGGL((pd_ln_rows_kernel<DM, 2>), dim3((M + 3) / 4), dim3(256), 0, s, d->h, d->hn, M, 1e-5f, 512.0f);
#if PD_STRIP_RT1 == 2
            if ((strip & 4) && rt3) pd_gemm_strip<4, 1, true, 1, PD_STRIP_K64>((const unsigned *)d->hn, DM, L.ff1_wh, DM, L.ff1_b, d->ff, M, DFF, s, L.ff1_cs, L.ff_scale);
            else
#endif
            if ((strip & 4) && PD_STRIP_RT3_FF1 && rt3) pd_gemm_strip<4, 3, true, 1, PD_STRIP_K64>((const unsigned *)d->hn, DM, L.ff1_wh, DM, L.ff1_b, d->ff, M, DFF, s, L.ff1_cs, L.ff_scale);
            else if (strip & 4) pd_gemm_strip<4, 2, true, 1, PD_STRIP_K64>((const unsigned *)d->hn, DM, L.ff1_wh, DM, L.ff1_b, d->ff, M, DFF, s, L.ff1_cs, L.ff_scale);
            else pd_gemm_split<4, 1, 2, true>((const unsigned *)d->hn, DM, L.ff1_wh, DM, L.ff1_b, d->ff, M, DFF, s, L.ff1_cs, L.ff_scale);
#if PD_STRIP_RT1
            if ((strip & 8) && rt3) pd_gemm_strip<2, 1, true, 1, PD_STRIP_K64>((const unsigned *)d->ff, DFF, L.ff2_wh, DFF, L.ff2_b, d->h, M, DM, s, L.ff2_cs);
            else
#endif
            if ((strip & 8) && rt3) pd_gemm_strip<2, 3, true, 1, PD_STRIP_K64>((const unsigned *)d->ff, DFF, L.ff2_wh, DFF, L.ff2_b, d->h, M, DM, s, L.ff2_cs);
            else if (strip & 8) pd_gemm_strip<2, 2, true, 1, PD_STRIP_K64>((const unsigned *)d->ff, DFF, L.ff2_wh, DFF, L.ff2_b, d->h, M, DM, s, L.ff2_cs);
            else pd_gemm_split<2, 1, 1, true>((const unsigned *)d->ff, DFF, L.ff2_wh, DFF, L.ff2_b, d->h, M, DM, s, L.ff2_cs);
            continue;
        }
        if (streamed && eng->den_split == 1 && d->split_ready) {
            // fast mode: the four encoder GEMMs on the bf16 matrix pipe in split precision (pd_gemm_split.h); activations
            // between them as split words -- LayerNorm, attention and the FF1 epilogue write them in place of fp32
            hipLaunchKernelGGL((pd_ln_rows_kernel<DM, 1>), dim3((M + 3) / 4), dim3(256), 0, s, d->h, d->hn, M, 1e-5f, 1.0f);
            pd_gemm_split<0, 1, 2>((const unsigned *)d->hn, DM, L.qkv_ws, DM, L.qkv_b, d->qkv, M, 3 * DM, s);
            hipLaunchKernelGGL(pd_attn_seq_kernel<1>, dim3(B * NH), dim3(256), attn_seq_lds(N), s, d->qkv, d->ctx, N, 1.0f);
            pd_gemm_split<2, 1, 1>((const unsigned *)d->ctx, DM, L.out_ws, DM, L.out_b, d->h, M, DM, s);
            hipLaunchKernelGGL((pd_ln_rows_kernel<DM, 1>), dim3((M + 3) / 4), dim3(256), 0, s, d->h, d->hn, M, 1e-5f, 1.0f);
            pd_gemm_split<4, 1, 2>((const unsigned *)d->hn, DM, L.ff1_ws, DM, L.ff1_b, d->ff, M, DFF, s);
            pd_gemm_split<2, 1, 1>((const unsigned *)d->ff, DFF, L.ff2_ws, DFF, L.ff2_b, d->h, M, DM, s);
            continue;
        }
        if (streamed) {
            float2 *stats = (float2 *)d->hn;           // (mean, rstd) per token row; applied in the A staging of the next GEMM
            hipLaunchKernelGGL(pd_ln_stats_kernel<DM>, dim3((M + 3) / 4), dim3(256), 0, s, d->h, stats, M, 1e-5f);
            pd_gemm_dma<0, true>(d->h, DM, L.qkv_wf, DM, L.qkv_b, d->qkv, M, 3 * DM, s, stats);                // LayerNorm-1 at the fragment reads
            hipLaunchKernelGGL(pd_attn_seq_kernel<0>, dim3(B * NH), dim3(256), attn_seq_lds(N), s, d->qkv, d->ctx, N, 1.0f);
            pd_gemm_dma<2>(d->ctx, DM, L.out_wf, DM, L.out_b, d->h, M, DM, s);
            hipLaunchKernelGGL(pd_ln_stats_kernel<DM>, dim3((M + 3) / 4), dim3(256), 0, s, d->h, stats, M, 1e-5f);
            pd_gemm_dma<1, true>(d->h, DM, L.ff1_wf, DM, L.ff1_b, d->ff, M, DFF, s, stats);                     // LayerNorm-2 likewise
            pd_gemm_dma<2>(d->ff, DFF, L.ff2_wf, DFF, L.ff2_b, d->h, M, DM, s);
            continue;
        }
        // x += MHA(LN1(x))
        g.A = d->h; g.bias = L.qkv_b; g.C = d->qkv; g.Nout = 3 * DM;
        launch_gemm<DM, 1, 0>(g, L.qkv_wp, MT, eng->gemm_wide_min_tiles, s);
#ifdef PD_DEN_STAMPS
        hipLaunchKernelGGL(pd_attn_kernel<false>, dim3(B * NH, (N + 3) / 4), dim3(256), ((2 * N + 4) * (DH + 4) + 4 * 64) * 4, s, d->qkv, d->ctx, N, next_stamp_slot());
#else
        hipLaunchKernelGGL(pd_attn_kernel<false>, dim3(B * NH, (N + 3) / 4), dim3(256), ((2 * N + 4) * (DH + 4) + 4 * 64) * 4, s, d->qkv, d->ctx, N);
#endif
        g.A = d->ctx; g.bias = L.out_b; g.C = d->h; g.Nout = DM;
        launch_gemm<DM, 0, 2>(g, L.out_wp, MT, eng->gemm_wide_min_tiles, s);
        // x += W2 relu(W1 LN2(x))
        g.A = d->h; g.bias = L.ff1_b; g.C = d->ff; g.Nout = DFF;
        launch_gemm<DM, 1, 1>(g, L.ff1_wp, MT, eng->gemm_wide_min_tiles, s);
        g.A = d->ff; g.bias = L.ff2_b; g.C = d->h; g.Nout = DM;
        launch_gemm<DFF, 0, 2>(g, L.ff2_wp, MT, eng->gemm_wide_min_tiles, s);
    }
    // _last.0 as a plain tile GEMM, then the fused LN/ReLU/Linear(128->9)/DDPM tail
    if (streamed) {
        pd_gemm_dma<0>(d->h, DM, d->last0_wf, DM, d->last0_b, d->hid, M, HID, s);
    } else {
        g.A = d->h; g.bias = d->last0_b; g.C = d->hid; g.Nout = HID;
        launch_gemm<DM, 0, 0>(g, d->last0_wp, MT, eng->gemm_wide_min_tiles, s);
    }
    HeadArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.hid = d->hid; ha.lnw = d->last_ln_w; ha.lnb = d->last_ln_b;
    ha.w3 = d->last3_w; ha.b3 = d->last3_b; ha.x = x; ha.noise = noise;
    ha.eps_out = eps_out; ha.mean_out = mean_out; ha.x0_out = x0_out; ha.xnext_out = x_next_out;
    ha.c_recip = eng->c_recip[t]; ha.c_recipm1 = eng->c_recipm1[t]; ha.coef1 = eng->coef1[t]; ha.coef2 = eng->coef2[t];
    ha.sigma = expf(0.5f * eng->logvar[t]);
    ha.M = M;
    ha.pred_x0 = eng->pred_x0;
#ifdef PD_DEN_STAMPS
    ha.stamps = streamed ? nullptr : next_stamp_slot();
#endif
    hipLaunchKernelGGL(pd_tail_kernel, dim3((M + 3) / 4), dim3(256), 0, s, ha);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

// ---- rows D2 / D3 as stand-alone operators (the reference's util/embedding.py modules called piecewise) -------------------------------
extern "C" int pd_time_embedding(const float *w0, const float *b0, const float *w2, const float *b2, const float *timesteps, int n,
                                 float *out, void *stream) {
    if (n == 0) return PD_OK;                       // an empty batch: nothing to read or write (the pointers may be NULL)
    if (!w0 || !b0 || !w2 || !b2 || !timesteps || !out || n < 0) {
        pd_set_error("pd_time_embedding: invalid arguments (n=%d)", n);
        return PD_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(pd_time_embed_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, timesteps, w0, b0, w2, b2, out);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
extern "C" int pd_pose_embedding(const float *x, long long rows, int dim, float *out, void *stream) {
    if (rows == 0 && dim >= 1) return PD_OK;        // an empty batch (the pointers may be NULL)
    if (!x || !out || rows < 0 || dim < 1 || dim > 4096) {
        pd_set_error("pd_pose_embedding: invalid arguments (rows=%lld dim=%d)", rows, dim);
        return PD_ERR_INVALID_ARG;
    }
    const long long total = rows * 21 * dim;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(pd_harmonic_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, dim, out);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

// ---- probe: fp16-subnormal operands on the fp16 matrix pipe (pd_engine.h pd_debug_mfma_f16_subnormal) ----------------------------
__global__ __launch_bounds__(64) void pd_mfma_f16_subnormal_kernel(float *out) {
    const float av[4] = {9.5367431640625e-07f, 1024.0f, 9.5367431640625e-07f, 1.0f};      // 2^-20 is an fp16 subnormal (min normal 2^-14)
    const float bv[4] = {1024.0f, 9.5367431640625e-07f, 0.0625f, 1.0f};
    for (int c = 0; c < 4; ++c) {
        f16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] = (_Float16)av[c];
            b[e] = (_Float16)bv[c];
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (threadIdx.x == 0) {
            out[c] = acc[0];
            out[4 + c] = (float)a[0];       // what the conversion itself kept of the operand
        }
    }
}
extern "C" int pd_debug_mfma_f16_subnormal(float *out4_host, void *stream) {
    if (!out4_host) return PD_ERR_INVALID_ARG;
    float *d = nullptr;
    PD_HIP_CHECK(hipMalloc((void **)&d, 8 * sizeof(float)));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pd_mfma_f16_subnormal_kernel, dim3(1), dim3(64), 0, s, d);
    float h[8];
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) {
        pd_set_error("pd_debug_mfma_f16_subnormal: %s", hipGetErrorString(e));
        return PD_ERR_HIP;
    }
    for (int i = 0; i < 4; ++i) out4_host[i] = h[i];
    if (h[4] == 0.0f) out4_host[0] = -1.0f;   // the fp32 -> fp16 conversion itself flushed 2^-20 (would make the probe meaningless)
    return PD_OK;
}

