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
#include "pd_internal.h"

#include <math.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DM 512          // d_model
#define NH 4            // heads
#define DH 128          // head dim
#define DFF 1024        // feed-forward dim
#define ZD 384          // z_dim
#define KFIRST 702      // 189 + 128 + 384 + 1   (denoiser.py:39)
#define KFIRST_PAD 704
// The engine permutes the K axis of _first so the wide pieces land 16-byte aligned in LDS:
//   engine column k' : [0,384) z | [384,512) t_emb | [512,692) harmonic | [692,701) x | 701 pivot | 702,703 pad
//   reference column : [0,180) harmonic | [180,189) x | [189,317) t_emb | [317,701) z | 701 pivot  (denoiser.py:68)
__host__ __device__ inline int pd_first_col(int kp) {
    if (kp < 384) return 317 + kp;
    if (kp < 512) return 189 + (kp - 384);
    if (kp < 692) return kp - 512;
    if (kp < 701) return 180 + (kp - 692);
    return kp;   // 701 pivot; 702/703 are padding (>= KFIRST -> zero)
}
#define HID 128         // mlp_hidden_dim

struct PdLayerDev {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    float *qkv_wp, *qkv_b;     // packed [1536/32][512/8][64][4]
    float *out_wp, *out_b;
    float *ff1_wp, *ff1_b;
    float *ff2_wp, *ff2_b;
};

struct PdDenoiserDev {
    int num_layers = 0, timesteps = 0, m_cap = 0;
    float *t_table = nullptr;          // [T,128] time embeddings
    float *first_wp = nullptr, *first_b = nullptr;
    PdLayerDev layers[PD_MAX_LAYERS];
    float *last0_wp = nullptr, *last0_b = nullptr, *last_ln_w = nullptr, *last_ln_b = nullptr;
    float *last3_w = nullptr, *last3_b = nullptr;   // [9,128] plain
    float *h = nullptr, *qkv = nullptr, *ctx = nullptr, *ff = nullptr, *hid = nullptr;   // activations [m_cap, .]
    std::vector<void *> allocs;
};

// --------------------------------------------------------------------------------------------
// weight repack: W[Nout][K] row-major  ->  Wp[nt][kc][lane][4], lane l holds
// W[nt*32 + (l & 31)][kc*8 + 4*(l >> 5) + 0..3]   (zero padded)
// --------------------------------------------------------------------------------------------
__global__ void pd_repack_kernel(const float *__restrict__ W, int Nout, int K, int KC, float *__restrict__ Wp, size_t total,
                                 int first_perm) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int l = (idx >> 2) & 63;
        const size_t rest = idx >> 8;
        const int kc = (int)(rest % KC);
        const int nt = (int)(rest / KC);
        const int n = nt * 32 + (l & 31);
        int k = kc * 8 + 4 * (l >> 5) + e;
        if (first_perm) k = pd_first_col(k);
        Wp[idx] = (n < Nout && k < K) ? W[(size_t)n * K + k] : 0.0f;
    }
}

// time-step embedding table (util/embedding.py:28-37): one block per step t
__global__ void pd_time_table_kernel(const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w2,
                                     const float *__restrict__ b2, float *__restrict__ table) {
    __shared__ float emb[256];
    __shared__ float hid[128];
    const int t = blockIdx.x, i = threadIdx.x;   // 128 threads
    // freqs = exp(-ln(10000) * arange(128, fp32) / 128)  (embedding.py:24-26), args = t * freqs
    const float freq = expf((-9.210340371976184f * (float)i) / 128.0f);
    const float arg = (float)t * freq;
    emb[i] = cosf(arg);
    emb[128 + i] = sinf(arg);
    __syncthreads();
    float a = b0[i];
    for (int k = 0; k < 256; ++k) a = fmaf(emb[k], w0[i * 256 + k], a);
    hid[i] = a / (1.0f + expf(-a));   // SiLU
    __syncthreads();
    float o = b2[i];
    for (int k = 0; k < 128; ++k) o = fmaf(hid[k], w2[i * 128 + k], o);
    table[t * 128 + i] = o;
}

// --------------------------------------------------------------------------------------------
// fused 32x32-tile GEMM:  C[m, n] = epi( sum_k A'[m, k] * W[n, k] + bias[n] )
//   AMODE 0: A' = A                      (plain rows of a [M, K] activation)
//   AMODE 1: A' = LayerNorm(A) (K = 512) (norm_first encoder layer, eps 1e-5)
//   AMODE 2: A' = [harmonic(x) | t_emb | z | pivot | 0 0]  (K = 704, denoiser.py:56-68)
//   EPI   0: + bias     1: relu(+ bias)     2: + bias + residual (in place on C)
// --------------------------------------------------------------------------------------------
struct GemmArgs {
    const float *A;        // [M, K] (AMODE 0/1)
    const float *Wp;       // packed weights
    const float *bias;     // [Nout]
    float *C;              // [M, Nout]
    const float *ln_w, *ln_b;
    // AMODE 2
    const float *x, *z, *temb;   // x [M,9], z [M,384], temb [128] (row of the table for this t)
    int n_frames;
    int M, Nout;
};

template <int K, int AMODE, int EPI>
__global__ __launch_bounds__(256) void pd_gemm_kernel(GemmArgs g) {
    constexpr int LDA = K + 4;            // padded row stride (floats): conflict-free ds_read_b128
    constexpr int KC = K / 8;             // 8-wide k chunks
    constexpr int CPW = KC / 4;           // chunks per wave (split-K over the 4 waves)
    constexpr int NB = (CPW > 16) ? 2 : 1;   // weight batches held in registers
    constexpr int BATCH = CPW / NB;
    static_assert(CPW % NB == 0, "chunk batching");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                      // [32][LDA]; later aliased by the cross-wave reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const float4 *wp = (const float4 *)g.Wp + ((size_t)blockIdx.y * KC + (size_t)wave * CPW) * 64 + lane;

    // ---- weights first: the whole first batch of this wave's fragments goes in flight before the
    // activation staging, so the HBM/MALL latency of the weight stream hides under it --------------
    float4 w0[BATCH];
#pragma unroll
    for (int c = 0; c < BATCH; ++c) w0[c] = wp[(size_t)c * 64];

    // ---- stage the 32 activation rows (fused LN / embedding); no predicated loads ---------------
    {
        const int r = tid >> 3, sub = tid & 7;
        const int m = m0 + r;
        const bool live = m < g.M;
        const int mr = live ? m : g.M - 1;   // clamp: padded rows load a valid row and are zeroed
        float *dst = As + r * LDA;
        if constexpr (AMODE == 2) {
            // engine column order (pd_first_col): z | t_emb | harmonic | x | pivot | pad
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
            static_assert(AMODE != 1 || K == 512, "LayerNorm staging is built for d_model = 512");
            float4 v[K / 32];
            const float4 *src = (const float4 *)(g.A + (size_t)mr * K);
#pragma unroll
            for (int i = 0; i < K / 32; ++i) v[i] = src[sub + 8 * i];
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            const float mean = s * (1.0f / K);
            float q = 0.0f;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) {
                const float a = v[i].x - mean, b2 = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b2 * b2) + (c * c + d * d);
            }
            q += __shfl_xor(q, 1, 64);
            q += __shfl_xor(q, 2, 64);
            q += __shfl_xor(q, 4, 64);
            const float rstd = live ? 1.0f / sqrtf(q * (1.0f / K) + 1e-5f) : 0.0f;
            const float4 *gw = (const float4 *)g.ln_w, *gb = (const float4 *)g.ln_b;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) {
                const float4 w = gw[sub + 8 * i], bb = gb[sub + 8 * i];
                float4 o;
                o.x = (v[i].x - mean) * rstd * w.x + (live ? bb.x : 0.0f);
                o.y = (v[i].y - mean) * rstd * w.y + (live ? bb.y : 0.0f);
                o.z = (v[i].z - mean) * rstd * w.z + (live ? bb.z : 0.0f);
                o.w = (v[i].w - mean) * rstd * w.w + (live ? bb.w : 0.0f);
                *(float4 *)(dst + 4 * (sub + 8 * i)) = o;
            }
        } else {
            const float4 *src = (const float4 *)(g.A + (size_t)mr * K);
            const float keep = live ? 1.0f : 0.0f;
            constexpr int NV = K / 32;
            constexpr int VB = 16;        // loads in flight per pass
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
    __syncthreads();

    // ---- split-K MFMA loop: wave w owns k-chunks [w*CPW, (w+1)*CPW) --------------------------
    float4 w1[NB == 2 ? BATCH : 1];
    if constexpr (NB == 2) {
#pragma unroll
        for (int c = 0; c < BATCH; ++c) w1[c] = wp[(size_t)(BATCH + c) * 64];
        __builtin_amdgcn_sched_barrier(0);   // keep the second batch's loads ahead of the first MFMAs
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float *arow = As + (lane & 31) * LDA + wave * CPW * 8 + 4 * (lane >> 5);
#pragma unroll
    for (int c = 0; c < BATCH; ++c) {
        const float4 af = *(const float4 *)(arow + c * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, w0[c].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, w0[c].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, w0[c].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, w0[c].w, acc, 0, 0, 0);
    }
    if constexpr (NB == 2) {
#pragma unroll
        for (int c = 0; c < BATCH; ++c) {
            const float4 af = *(const float4 *)(arow + (BATCH + c) * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, w1[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, w1[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, w1[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, w1[c].w, acc, 0, 0, 0);
        }
    }
    __syncthreads();   // every wave is done reading As; reuse it for the reduction

    // ---- cross-wave reduction in fixed order + fused epilogue ---------------------------------
    float *red = lds;   // [4][16][64]
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
    const int col = n0 + (lane & 31);
    const float bias = g.bias[col];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int reg = wave * 4 + i;
        float v = red[(0 * 16 + reg) * 64 + lane];
        v += red[(1 * 16 + reg) * 64 + lane];
        v += red[(2 * 16 + reg) * 64 + lane];
        v += red[(3 * 16 + reg) * 64 + lane];
        v += bias;
        const int row = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < g.M) {
            float *cp = g.C + (size_t)row * g.Nout + col;
            if constexpr (EPI == 1) v = fmaxf(v, 0.0f);
            if constexpr (EPI == 2) v += *cp;
            *cp = v;
        }
    }
}

// --------------------------------------------------------------------------------------------
// attention core: softmax(q k^T / sqrt(dh)) v for one (sequence, head), N <= 64 frames, no mask
// (nn.MultiheadAttention inside the encoder layer).  grid = (B*heads, ceil(N/4)): every
// workgroup stages K and V of its (sequence, head) and each of its 4 waves owns ONE query row:
// lane j scores key j, softmax is a wave reduction, lanes then own 2 of the 128 output dims.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float pd_wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float pd_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void pd_attn_kernel(const float *__restrict__ qkv, float *__restrict__ ctx, int N) {
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
        out[lane] = o0;
        out[64 + lane] = o1;
    }
}

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
};

__global__ __launch_bounds__(256) void pd_tail_kernel(HeadArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= g.M) return;
    const float *row = g.hid + (size_t)m * HID;
    const float v0 = row[lane], v1 = row[64 + lane];
    const float mean = pd_wave_sum(v0 + v1) * (1.0f / HID);
    const float d0 = v0 - mean, d1 = v1 - mean;
    const float rstd = 1.0f / sqrtf(pd_wave_sum(d0 * d0 + d1 * d1) * (1.0f / HID) + 1e-5f);
    const float a0 = fmaxf(d0 * rstd * g.lnw[lane] + g.lnb[lane], 0.0f);
    const float a1 = fmaxf(d1 * rstd * g.lnw[64 + lane] + g.lnb[64 + lane], 0.0f);
    float e = 0.0f;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const float part = pd_wave_sum(fmaf(a0, g.w3[o * HID + lane], a1 * g.w3[o * HID + 64 + lane]));
        e = (lane == o) ? part : e;
    }
    if (lane < 9) {
        e += g.b3[lane];
        const size_t at = (size_t)m * 9 + lane;
        const float xv = g.x[at];
        const float x0 = g.c_recip * xv - g.c_recipm1 * e;          // gaussian_diffuser.py:190-194
        const float mu = g.coef1 * x0 + g.coef2 * xv;               // :201-205
        if (g.eps_out) g.eps_out[at] = e;
        if (g.x0_out) g.x0_out[at] = x0;
        if (g.mean_out) g.mean_out[at] = mu;
        if (g.xnext_out) g.xnext_out[at] = g.noise ? mu + g.sigma * g.noise[at] : mu;   // :280
    }
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
static int dev_pack(PdDenoiserDev *d, float **dst, const float *W, int Nout, int K, int Kpad, int first_perm = 0) {
    if (!W) {
        pd_set_error("pd_engine_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    const int NT = (Nout + 31) / 32, KC = Kpad / 8;
    const size_t total = (size_t)NT * KC * 256;
    int rc = dev_alloc(d, dst, total);
    if (rc) return rc;
    hipLaunchKernelGGL(pd_repack_kernel, dim3(512), dim3(256), 0, 0, W, Nout, K, KC, *dst, total, first_perm);
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
    PD_TRY(dev_pack(d, &d->first_wp, w->first_w, DM, KFIRST, KFIRST_PAD, 1));
    PD_TRY(dev_copy(d, &d->first_b, w->first_b, DM));
    for (int l = 0; l < w->num_layers; ++l) {
        const pd_layer_weights &s = w->layers[l];
        PdLayerDev &L = d->layers[l];
        PD_TRY(dev_copy(d, &L.ln1_w, s.norm1_w, DM));
        PD_TRY(dev_copy(d, &L.ln1_b, s.norm1_b, DM));
        PD_TRY(dev_copy(d, &L.ln2_w, s.norm2_w, DM));
        PD_TRY(dev_copy(d, &L.ln2_b, s.norm2_b, DM));
        PD_TRY(dev_pack(d, &L.qkv_wp, s.in_proj_w, 3 * DM, DM, DM));
        PD_TRY(dev_copy(d, &L.qkv_b, s.in_proj_b, 3 * DM));
        PD_TRY(dev_pack(d, &L.out_wp, s.out_proj_w, DM, DM, DM));
        PD_TRY(dev_copy(d, &L.out_b, s.out_proj_b, DM));
        PD_TRY(dev_pack(d, &L.ff1_wp, s.linear1_w, DFF, DM, DM));
        PD_TRY(dev_copy(d, &L.ff1_b, s.linear1_b, DFF));
        PD_TRY(dev_pack(d, &L.ff2_wp, s.linear2_w, DM, DFF, DFF));
        PD_TRY(dev_copy(d, &L.ff2_b, s.linear2_b, DM));
    }
    PD_TRY(dev_pack(d, &d->last0_wp, w->last0_w, HID, DM, DM));
    PD_TRY(dev_copy(d, &d->last0_b, w->last0_b, HID));
    PD_TRY(dev_copy(d, &d->last_ln_w, w->last_ln_w, HID));
    PD_TRY(dev_copy(d, &d->last_ln_b, w->last_ln_b, HID));
    PD_TRY(dev_copy(d, &d->last3_w, w->last3_w, 9 * HID));
    PD_TRY(dev_copy(d, &d->last3_b, w->last3_b, 9));
    PD_TRY(dev_alloc(d, &d->h, (size_t)d->m_cap * DM));
    PD_TRY(dev_alloc(d, &d->qkv, (size_t)d->m_cap * 3 * DM));
    PD_TRY(dev_alloc(d, &d->ctx, (size_t)d->m_cap * DM));
    PD_TRY(dev_alloc(d, &d->ff, (size_t)d->m_cap * DFF));
    PD_TRY(dev_alloc(d, &d->hid, (size_t)d->m_cap * HID));
    PD_TRY(set_lds(pd_gemm_kernel<KFIRST_PAD, 2, 0>, 32 * (KFIRST_PAD + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 1, 0>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 1, 1>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 0, 2>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DFF, 0, 2>, 32 * (DFF + 4) * 4));
    PD_TRY(set_lds(pd_gemm_kernel<DM, 0, 0>, 32 * (DM + 4) * 4));
    PD_TRY(set_lds(pd_attn_kernel, ((2 * 64 + 4) * (DH + 4) + 4 * 64) * 4));
    PD_HIP_CHECK(hipDeviceSynchronize());
    return PD_OK;
}

void pd_denoiser_destroy(pd_engine *eng) {
    if (!eng->den) return;
    for (void *p : eng->den->allocs) (void)hipFree(p);
    delete eng->den;
    eng->den = nullptr;
}

int pd_denoiser_launch(pd_engine *eng, const float *x, const float *z, int t, int B, int N, float *eps_out,
                       float *mean_out, float *x0_out, const float *noise, float *x_next_out, hipStream_t s) {
    PdDenoiserDev *d = eng->den;
    if (!x || !z || B <= 0 || N <= 0 || B > eng->max_B || N > eng->max_N || N > 64 || t < 0 || t >= d->timesteps) {
        pd_set_error("denoiser: invalid arguments (B=%d N=%d t=%d; max_B=%d max_N=%d, N <= 64, 0 <= t < %d)", B, N, t,
                     eng->max_B, eng->max_N, d->timesteps);
        return PD_ERR_INVALID_ARG;
    }
    const int M = B * N, MT = (M + 31) / 32;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = M;
    // _first with the embedding fused into the A staging
    g.Wp = d->first_wp; g.bias = d->first_b; g.C = d->h; g.Nout = DM;
    g.x = x; g.z = z; g.temb = d->t_table + (size_t)t * 128; g.n_frames = N;
    hipLaunchKernelGGL((pd_gemm_kernel<KFIRST_PAD, 2, 0>), dim3(MT, DM / 32), dim3(256), 32 * (KFIRST_PAD + 4) * 4, s, g);
    for (int l = 0; l < d->num_layers; ++l) {
        const PdLayerDev &L = d->layers[l];
        // x += MHA(LN1(x))
        g.A = d->h; g.Wp = L.qkv_wp; g.bias = L.qkv_b; g.C = d->qkv; g.Nout = 3 * DM; g.ln_w = L.ln1_w; g.ln_b = L.ln1_b;
        hipLaunchKernelGGL((pd_gemm_kernel<DM, 1, 0>), dim3(MT, 3 * DM / 32), dim3(256), 32 * (DM + 4) * 4, s, g);
        hipLaunchKernelGGL(pd_attn_kernel, dim3(B * NH, (N + 3) / 4), dim3(256), ((2 * N + 4) * (DH + 4) + 4 * 64) * 4, s, d->qkv, d->ctx, N);
        g.A = d->ctx; g.Wp = L.out_wp; g.bias = L.out_b; g.C = d->h; g.Nout = DM;
        hipLaunchKernelGGL((pd_gemm_kernel<DM, 0, 2>), dim3(MT, DM / 32), dim3(256), 32 * (DM + 4) * 4, s, g);
        // x += W2 relu(W1 LN2(x))
        g.A = d->h; g.Wp = L.ff1_wp; g.bias = L.ff1_b; g.C = d->ff; g.Nout = DFF; g.ln_w = L.ln2_w; g.ln_b = L.ln2_b;
        hipLaunchKernelGGL((pd_gemm_kernel<DM, 1, 1>), dim3(MT, DFF / 32), dim3(256), 32 * (DM + 4) * 4, s, g);
        g.A = d->ff; g.Wp = L.ff2_wp; g.bias = L.ff2_b; g.C = d->h; g.Nout = DM;
        hipLaunchKernelGGL((pd_gemm_kernel<DFF, 0, 2>), dim3(MT, DM / 32), dim3(256), 32 * (DFF + 4) * 4, s, g);
    }
    // _last.0 as a plain tile GEMM (4 N-tiles), then the fused LN/ReLU/Linear(128->9)/DDPM tail
    g.A = d->h; g.Wp = d->last0_wp; g.bias = d->last0_b; g.C = d->hid; g.Nout = HID;
    hipLaunchKernelGGL((pd_gemm_kernel<DM, 0, 0>), dim3(MT, HID / 32), dim3(256), 32 * (DM + 4) * 4, s, g);
    HeadArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.hid = d->hid; ha.lnw = d->last_ln_w; ha.lnb = d->last_ln_b;
    ha.w3 = d->last3_w; ha.b3 = d->last3_b; ha.x = x; ha.noise = noise;
    ha.eps_out = eps_out; ha.mean_out = mean_out; ha.x0_out = x0_out; ha.xnext_out = x_next_out;
    ha.c_recip = eng->c_recip[t]; ha.c_recipm1 = eng->c_recipm1[t]; ha.coef1 = eng->coef1[t]; ha.coef2 = eng->coef2[t];
    ha.sigma = expf(0.5f * eng->logvar[t]);
    ha.M = M;
    hipLaunchKernelGGL(pd_tail_kernel, dim3((M + 3) / 4), dim3(256), 0, s, ha);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
