// pd_vit.hip -- the multi-scale DINO ViT-S/16 image feature extractor.
//
// Replaces (paths relative to /root/reference/pose_diffusion/):
//   models/image_feature_extractor.py:57-87   ImageNet normalisation, bilinear rescaling by every scale factor,
//                                             backbone at every scale, average of the CLS features
//   models/image_feature_extractor.py:40-42   the backbone: torch.hub "facebookresearch/dino:main" dino_vits16 -- third-party
//                                             code that is NOT in the reference tree; restated from the published algorithm
//                                             (vision_transformer.py: patch embedding conv 16x16/16, CLS token, bicubically
//                                             resampled position grid, 12 pre-norm blocks [LayerNorm eps 1e-6, 6-head
//                                             attention, GELU MLP 384-1536-384], final LayerNorm, CLS output)
// Everything is fp32 on the exact-fp32 matrix instruction (the features feed the denoiser, whose parity contract is 1e-4).
// Structure per scale, batched over all frames:  normalise+resize -> patch GEMM (im2col in the A staging, position embedding
// and token placement in the epilogue) -> 12 x [LN+QKV GEMM, attention, proj GEMM + residual, LN+FC1 GEMM + GELU,
// FC2 GEMM + residual (two 768-deep halves)] -> LayerNorm of the CLS rows, accumulated over the scales.
// The GEMM is the denoiser's small-tile kernel shape (32 x 32 output tile, 4 waves split K, activation tile staged once in
// LDS with the LayerNorm fused, weights pre-packed in MFMA fragment order) generalised to K = 384 / 768.
#include "pd_internal.h"

#include <math.h>
#include <string.h>
#include <vector>

#include "pd_gemm_stream.h"
#include "pd_gemm_split.h"

#define VD 384          // embedding dim
#define VH 6            // heads
#define VDH 64          // head dim
#define VFF 1536        // MLP hidden
#define VP 16           // patch size
#define VKP 768         // 3 * 16 * 16
#define VT_MAX 1056     // tokens per image: the attention kernel keeps one 32 x T score block in LDS (images up to 512 x 512)
#define VDEPTH_MAX 16

struct pd_vit {
    int device = 0, depth = 0, grid0 = 0;       // grid0: side of the trained position grid (14)
    int exact_fp32 = 0;                         // PD_VIT_OPT_EXACT_FP32: 0 fp16 planes (default at >= 1 024 rows), 1 exact fp32, 2 bf16 planes (rounds 1-5)
    bool h_ready = false;                       // the fp16-plane weights and their static scales exist (finite weights)
    float ln_scale = 1.0f;                      // 2^e of the LayerNorm operand (|x^| <= sqrt(384))
    float *patch_wp = nullptr, *patch_b = nullptr, *cls = nullptr, *pos = nullptr;   // pos [1 + grid0^2, 384]
    struct Layer {
        float *qkv_wp, *qkv_b, *proj_wp, *proj_b, *fc1_wp, *fc1_b, *fc2a_wp, *fc2b_wp, *fc2_b;
        float *qkv_wf, *proj_wf, *fc1_wf, *fc2_wf;      // row-major copies (LayerNorm scale folded) for the streamed GEMM
        unsigned *qkv_ws, *proj_ws, *fc1_ws, *fc2_ws;   // split into bf16 hi / lo, in MFMA fragment order (vit_frag_split_kernel)
        unsigned *qkv_wh, *proj_wh, *fc1_wh, *fc2_wh;   // fp16 hi / lo planes of w * 2^e (round 6: the denoiser's fp16-plane mode, pd_gemm_strip_kernel<.., F16>)
        float qkv_cs, proj_cs, fc1_cs, fc2_cs;          // accumulator scales 2^-(e_operand + e_weight)
        float ctx_scale, hid_scale;                     // 2^e of the attention output / the GELU hidden rows (split-word operands of proj / fc2)
    } L[VDEPTH_MAX];
    float *norm_w = nullptr, *norm_b = nullptr, *zero_b = nullptr;
    // workspaces, sized at the first forward / grown on demand
    size_t cap_tokens = 0, cap_pixels = 0;
    float *x = nullptr, *xn = nullptr, *qkv = nullptr, *ctx = nullptr, *hid = nullptr, *img = nullptr;
    std::vector<void *> allocs;
};

// ---- small kernels -----------------------------------------------------------------------------------
// W[Nout][K] (row stride ldw, column offset koff) -> fragment order for v_mfma_f32_32x32x2_f32, optional per-column scale
__global__ void vit_repack_kernel(const float *__restrict__ W, int Nout, int K, int ldw, int koff, float *__restrict__ Wp, size_t total,
                                  const float *__restrict__ colscale) {
    const int KC = K / 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3, l = (idx >> 2) & 63;
        const size_t rest = idx >> 8;
        const int kc = (int)(rest % KC), nt = (int)(rest / KC);
        const int n = nt * 32 + (l & 31), k = kc * 8 + 4 * (l >> 5) + e;
        float v = (n < Nout) ? W[(size_t)n * ldw + koff + k] : 0.0f;
        if (colscale) v *= colscale[k];
        Wp[idx] = v;
    }
}
// b'[n] = b[n] + sum_k W[n][k] beta[k]
__global__ void vit_fold_bias_kernel(const float *__restrict__ W, const float *__restrict__ beta, const float *__restrict__ b, int Nout,
                                     int K, float *__restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Nout) return;
    float a = 0.0f;
    for (int k = 0; k < K; ++k) a = fmaf(W[(size_t)n * K + k], beta[k], a);
    out[n] = b[n] + a;
}

// (image - mean) / std (image_feature_extractor.py:62-63), then F.interpolate(scale_factor, bilinear, align_corners=False)
// (:86-87): source index (dst + 0.5) / scale_factor - 0.5 clamped at 0.  [n,3,H,W] -> [n,3,Hs,Ws]
__global__ void vit_prep_kernel(const float *__restrict__ in, int n, int H, int W, int Hs, int Ws, float inv_sf, int identity,
                                float *__restrict__ out) {
    const size_t total = (size_t)n * 3 * Hs * Ws;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % Ws), oy = (int)((idx / Ws) % Hs), c = (int)((idx / ((size_t)Ws * Hs)) % 3);
    const size_t im = idx / ((size_t)3 * Hs * Ws);
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f), sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    const float *src = in + (im * 3 + c) * (size_t)H * W;
    float v;
    if (identity) {
        v = (src[(size_t)oy * W + ox] - mean) / sd;
    } else {
        const float sy = fmaxf(inv_sf * ((float)oy + 0.5f) - 0.5f, 0.0f), sx = fmaxf(inv_sf * ((float)ox + 0.5f) - 0.5f, 0.0f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float a = (src[(size_t)y0 * W + x0] - mean) / sd, b = (src[(size_t)y0 * W + x1] - mean) / sd;
        const float d = (src[(size_t)y1 * W + x0] - mean) / sd, e = (src[(size_t)y1 * W + x1] - mean) / sd;
        v = hy * (hx * a + lx * b) + ly * (hx * d + lx * e);
    }
    out[idx] = v;
}

// CLS rows: x[img * T] = cls_token + pos[0]
__global__ void vit_cls_kernel(const float *__restrict__ cls, const float *__restrict__ pos0, int n_img, int T, float *__restrict__ x) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_img * VD) return;
    const int im = idx / VD, d = idx - im * VD;
    x[(size_t)im * T * VD + d] = cls[d] + pos0[d];
}

// final LayerNorm (eps 1e-6) of the CLS rows; z (+)= LN(x[img * T]) * weight: one wave per image, 6 values per lane
__global__ __launch_bounds__(64) void vit_final_kernel(const float *__restrict__ x, int T, const float *__restrict__ w, const float *__restrict__ b,
                                                       float scale, int accumulate, float *__restrict__ z) {
    const int im = blockIdx.x, lane = threadIdx.x;
    const float *row = x + (size_t)im * T * VD;
    float v[6], s = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v[i] = row[lane + 64 * i];
        s += v[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * (1.0f / VD);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) q += (v[i] - mean) * (v[i] - mean);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / VD) + 1e-6f);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = lane + 64 * i;
        const float o = ((v[i] - mean) * rstd * w[d] + b[d]) * scale;
        z[(size_t)im * VD + d] = accumulate ? z[(size_t)im * VD + d] + o : o;
    }
}

// ---- GEMM -----------------------------------------------------------------------------------------------
//   C[m, n] = epi( sum_k A'[m, k] W[n, k] + bias[n] ),  32 x 32 tile per workgroup, split-K over the 4 waves
//   AMODE 0: A' = A rows (row stride lda)      1: A' = LayerNorm(A) without affine (folded into W / bias), eps 1e-6
//   AMODE 3: A' = im2col of the prepared images: row = (image, patch), k = c * 256 + ky * 16 + kx
//   EPI   0: + bias   2: + bias + C (residual, in place)   3: gelu(+ bias)   4: + bias + pos[1 + patch], row -> token row
struct VitGemmArgs {
    const float *A, *Wp, *bias, *pos, *img;
    float *C;
    int M, Nout, lda, T, gh, gw, Hs, Ws;
};

template <int K, int AMODE, int EPI>
__global__ __launch_bounds__(256) void vit_gemm_kernel(VitGemmArgs g) {
    constexpr int LDA = K + 4;
    constexpr int KC = K / 8, CPW = KC / 4;        // 8-deep fragments, chunks per wave
    constexpr int BATCH = CPW / 2;                 // two register batches (12 / 24 float4 each)
    static_assert(KC % 4 == 0 && CPW % 2 == 0, "chunk batching");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int MT = (g.M + 31) / 32;
    const int ntile = blockIdx.x / MT, mtile = blockIdx.x % MT;     // consecutive blocks share the weight tile (L2)
    const int m0 = mtile * 32, n0 = ntile * 32;
    const float4 *wp = (const float4 *)g.Wp + ((size_t)ntile * KC + (size_t)wave * CPW) * 64 + lane;
    float4 w0[BATCH];
#pragma unroll
    for (int c = 0; c < BATCH; ++c) w0[c] = wp[(size_t)c * 64];
    {
        const int r = tid >> 3, sub = tid & 7;
        const int m = m0 + r;
        const bool live = m < g.M;
        const int mr = live ? m : g.M - 1;
        float *dst = As + r * LDA;
        if constexpr (AMODE == 3) {
            // one patch row: 3 channels x 16 lines of 16 pixels; thread `sub` takes lines sub, sub + 8 of every channel
            const int P = g.gh * g.gw;
            const int im = mr / P, p = mr - im * P, py = p / g.gw, px = p - py * g.gw;
            const float keep = live ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int ky = sub + 8 * hh;
                    const float *src = g.img + (((size_t)im * 3 + c) * g.Hs + (py * VP + ky)) * g.Ws + px * VP;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);   // rows may be unaligned (Ws odd)
                        v.x *= keep; v.y *= keep; v.z *= keep; v.w *= keep;
                        *(float4 *)(dst + c * 256 + ky * 16 + 4 * q) = v;
                    }
                }
        } else if constexpr (AMODE == 1) {
            float4 v[K / 32];
            const float4 *src = (const float4 *)(g.A + (size_t)mr * g.lda);
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
                const float a = v[i].x - mean, b2 = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b2 * b2) + (cc * cc + d * d);
            }
            q += __shfl_xor(q, 1, 64);
            q += __shfl_xor(q, 2, 64);
            q += __shfl_xor(q, 4, 64);
            const float rstd = live ? 1.0f / sqrtf(q * (1.0f / K) + 1e-6f) : 0.0f;
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
            const float4 *src = (const float4 *)(g.A + (size_t)mr * g.lda);
            const float keep = live ? 1.0f : 0.0f;
#pragma unroll
            for (int i0 = 0; i0 < K / 32; i0 += 12) {
                float4 v[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = src[sub + 8 * (i0 + i)];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    float4 o = v[i];
                    o.x *= keep; o.y *= keep; o.z *= keep; o.w *= keep;
                    *(float4 *)(dst + 4 * (sub + 8 * (i0 + i))) = o;
                }
            }
        }
    }
    __syncthreads();
    float4 w1[BATCH];
#pragma unroll
    for (int c = 0; c < BATCH; ++c) w1[c] = wp[(size_t)(BATCH + c) * 64];
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float *arow = As + (lane & 31) * LDA + wave * CPW * 8 + 4 * (lane >> 5);
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const float4 wf = (c < BATCH) ? w0[c < BATCH ? c : 0] : w1[c >= BATCH ? c - BATCH : 0];
        const float4 af = *(const float4 *)(arow + c * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, wf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, wf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, wf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, wf.w, acc, 0, 0, 0);
    }
    __syncthreads();
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
            if constexpr (EPI == 4) {
                const int P = g.gh * g.gw, im = row / P, p = row - im * P;
                g.C[((size_t)im * g.T + 1 + p) * g.Nout + col] = v + g.pos[(size_t)(1 + p) * g.Nout + col];
            } else {
                float *cp = g.C + (size_t)row * g.Nout + col;
                if constexpr (EPI == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // nn.GELU(), exact form
                if constexpr (EPI == 2) v += *cp;
                *cp = v;
            }
        }
    }
}

// ---- the same GEMM in split precision: x ~= hi + lo (two bf16), x.w ~= hi.hi + hi.lo + lo.hi on the bf16 matrix instruction ----
// (16 x the rate of the f32 instruction per product, three products; fp32 accumulation).  Measured deviation of the CLS
// features from the fp32 network: 8e-6 .. 1e-5 of max|z| (oracle-side simulation and GPU tests), a tenth of the 1e-4 contract.
//   activations between kernels: one 32-bit word per element {hi | lo << 16} (stores keep the fp32 pattern); the A staging
//   un-zips 8 words into an 8 x hi and an 8 x lo fragment on their way to LDS.  Weights are split and grouped at creation:
//   per row, per 8 k: 8 hi then 8 lo (32 B), so their staging is a plain copy.  LDS rows as in the f32 kernel (36 dwords).
// 8 consecutive fp32 (two float4) * scale -> an 8 x hi and an 8 x lo bf16 fragment
__device__ __forceinline__ void vit_split8(const float4 &a, const float4 &c, float scale, bf16x8 &h, bf16x8 &l) {
    const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, c.x * scale, c.y * scale, c.z * scale, c.w * scale};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = (__bf16)v[e];
        l[e] = (__bf16)(v[e] - (float)h[e]);
    }
}

// ---- attention on the matrix cores: one workgroup per (image, head, block of 32 query rows) --------------------------
//   S = (Q / 8) K^T : wave w takes the key tiles w, w + 4, ... (32 keys each); Q and K fragments come straight from global
//                     memory (L2) in the MFMA operand layout, the S tile goes to LDS [32][nkt * 32 + 4] (-inf past T)
//   P = softmax(S)  : 8 threads per row, in place
//   O = P V         : waves = 2 column tiles (32 of the 64 head dims) x 2 halves of the keys; V fragments from global memory
//                     (32 consecutive dims per half wave), the two halves summed through LDS
// SPLIT: 0 = exact-fp32 QK^T, fp32 ctx; 1 = QK^T on bf16 hi + lo planes, ctx as bf16 split words (rounds 1-5); 2 = exact-fp32 QK^T, ctx as fp16 split
// words of ctx * out_scale (round 6: the default at >= 1 024 rows -- only the four Linear layers run on fp16 planes, as in the denoiser)
template <int SPLIT>
__global__ __launch_bounds__(256) void vit_attn_kernel(const float *__restrict__ qkv, float *__restrict__ ctx, int T, int nqb, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nkt = (T + 31) >> 5, LP = nkt * 32 + 4;
    float *S = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int b = blockIdx.x;
    const int qb = b % nqb;
    b /= nqb;
    const int h = b % VH, im = b / VH, q0 = qb * 32;
    const float *base = qkv + (size_t)im * T * (3 * VD) + h * VDH;
    if constexpr (SPLIT == 1) {
        // S = (Q / 8) K^T in split precision: 8 consecutive head dims per lane and 16-k step (lanes 0-31 the even groups of 8,
        // lanes 32-63 the odd ones), each fp32 fragment split into bf16 hi + lo in registers, three bf16 products per step
        bf16x8 qh[4], ql[4];
        {
            const float4 *src = (const float4 *)(base + (size_t)min(q0 + l31, T - 1) * (3 * VD)) + 2 * hi;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float4 a = src[4 * st], c = src[4 * st + 1];
                vit_split8(a, c, 0.125f, qh[st], ql[st]);                                   // head_dim ** -0.5, exact
            }
        }
        float4 ka[4], kb[4];
        if (wave < nkt) {
            const float4 *src = (const float4 *)(base + (size_t)min(wave * 32 + l31, T - 1) * (3 * VD) + VD) + 2 * hi;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                ka[st] = src[4 * st];
                kb[st] = src[4 * st + 1];
            }
        }
        for (int kt = wave; kt < nkt; kt += 4) {
            bf16x8 kh8[4], kl8[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) vit_split8(ka[st], kb[st], 1.0f, kh8[st], kl8[st]);
            if (kt + 4 < nkt) {          // next tile's loads fly during this tile's products
                const float4 *src = (const float4 *)(base + (size_t)min((kt + 4) * 32 + l31, T - 1) * (3 * VD) + VD) + 2 * hi;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    ka[st] = src[4 * st];
                    kb[st] = src[4 * st + 1];
                }
            }
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql[st], kh8[st], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[st], kl8[st], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh[st], kh8[st], acc, 0, 0, 0);
            }
            const bool valid = kt * 32 + l31 < T;
#pragma unroll
            for (int i = 0; i < 16; ++i) S[((i & 3) + 8 * (i >> 2) + 4 * hi) * LP + kt * 32 + l31] = valid ? acc[i] : -INFINITY;
        }
    } else {
    float4 qf[8];
    {
        const int qr = min(q0 + l31, T - 1);
        const float4 *src = (const float4 *)(base + (size_t)qr * (3 * VD)) + hi;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 v = src[2 * c];
            qf[c] = make_float4(v.x * 0.125f, v.y * 0.125f, v.z * 0.125f, v.w * 0.125f);      // head_dim ** -0.5, exact
        }
    }
    float4 kf[8];
    if (wave < nkt) {
        const float4 *src = (const float4 *)(base + (size_t)min(wave * 32 + l31, T - 1) * (3 * VD) + VD) + hi;
#pragma unroll
        for (int c = 0; c < 8; ++c) kf[c] = src[2 * c];
    }
    for (int kt = wave; kt < nkt; kt += 4) {
        float4 kn[8];
        const bool more = kt + 4 < nkt;
        if (more) {
            const float4 *src = (const float4 *)(base + (size_t)min((kt + 4) * 32 + l31, T - 1) * (3 * VD) + VD) + hi;
#pragma unroll
            for (int c = 0; c < 8; ++c) kn[c] = src[2 * c];
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].x, kf[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].y, kf[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].z, kf[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].w, kf[c].w, acc, 0, 0, 0);
        }
        const bool valid = kt * 32 + l31 < T;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[((i & 3) + 8 * (i >> 2) + 4 * hi) * LP + kt * 32 + l31] = valid ? acc[i] : -INFINITY;
        if (more) {
#pragma unroll
            for (int c = 0; c < 8; ++c) kf[c] = kn[c];
        }
    }
    }
    // V fragments of this wave's first 16 key chunks: issued now, they land while the softmax runs
    const int nt = wave & 1, kh = wave >> 1;
    const int nk8 = (T + 7) >> 3, c_begin = kh ? (nk8 >> 1) : 0, c_end = kh ? nk8 : (nk8 >> 1);
    const float *vbase = base + 2 * VD + nt * 32 + l31;
    float vpre[16][4];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int k0 = min(c_begin + u, max(c_end - 1, 0)) * 8 + 4 * hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) vpre[u][e] = vbase[(size_t)min(k0 + e, T - 1) * (3 * VD)];          // P is 0 past T
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    {
        float *row = S + (tid >> 3) * LP;
        const int sub = tid & 7, ncol = nkt * 32;
        float mx = -INFINITY;
        for (int j = sub; j < ncol; j += 8) mx = fmaxf(mx, row[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        float sum = 0.0f;
        for (int j = sub; j < ncol; j += 8) {
            const float e = expf(row[j] - mx);
            row[j] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        for (int j = sub; j < ncol; j += 8) row[j] = row[j] / sum;
    }
    __syncthreads();
    const float *prow = S + l31 * LP + 4 * hi;
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.0f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (c_begin + u < c_end) {
            const float4 pf = *(const float4 *)(prow + (c_begin + u) * 8);
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.x, vpre[u][0], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.y, vpre[u][1], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.z, vpre[u][2], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.w, vpre[u][3], o, 0, 0, 0);
        }
    }
    for (int c4 = c_begin + 16; c4 < c_end; c4 += 4) {     // more than 256 keys: 4 chunks of 8 keys per trip, 16 loads in flight
        float vv[4][4];
        float4 pf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = min(c4 + u, c_end - 1), k0 = c * 8 + 4 * hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[u][e] = vbase[(size_t)min(k0 + e, T - 1) * (3 * VD)];      // P is 0 past T
            pf[u] = *(const float4 *)(prow + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c4 + u < c_end) {
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[u].x, vv[u][0], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[u].y, vv[u][1], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[u].z, vv[u][2], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[u].w, vv[u][3], o, 0, 0, 0);
            }
        }
    }
    __syncthreads();
    float *red = lds;      // [2][16][64]
    if (kh) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(nt * 16 + i) * 64 + lane] = o[i];
    }
    __syncthreads();
    if (!kh) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int qr = q0 + (i & 3) + 8 * (i >> 2) + 4 * hi;
            if (qr < T) {
                const float v = o[i] + red[(nt * 16 + i) * 64 + lane];
                float *dst = ctx + ((size_t)im * T + qr) * VD + h * VDH + nt * 32 + l31;
                if constexpr (SPLIT == 1)
                    *(unsigned *)dst = pd_split_word(v);       // feeds vit_gemm_split_kernel
                else if constexpr (SPLIT == 2)
                    *(unsigned *)dst = pd_split_word_h(v * out_scale);     // feeds pd_gemm_strip_kernel<.., F16>
                else
                    *dst = v;
            }
        }
    }
}
static size_t vit_attn_lds(int T) {
    const size_t s = (size_t)32 * (((T + 31) / 32) * 32 + 4);
    return (s > 2048 ? s : 2048) * sizeof(float);
}

// ---- host ---------------------------------------------------------------------------------------------------
#define VIT_TRY(expr)        \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

static int vit_alloc(pd_vit *v, float **p, size_t n) {
    PD_HIP_CHECK(hipMalloc((void **)p, n * sizeof(float)));
    v->allocs.push_back(*p);
    return PD_OK;
}
static int vit_copy(pd_vit *v, float **dst, const float *src, size_t n) {
    if (!src) {
        pd_set_error("pd_vit_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    VIT_TRY(vit_alloc(v, dst, n));
    PD_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice));
    return PD_OK;
}
static int vit_pack(pd_vit *v, float **dst, const float *W, int Nout, int K, int ldw, int koff, const float *gamma) {
    if (!W) {
        pd_set_error("pd_vit_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    const size_t total = (size_t)(Nout / 32) * (K / 8) * 256;
    VIT_TRY(vit_alloc(v, dst, total));
    hipLaunchKernelGGL(vit_repack_kernel, dim3(512), dim3(256), 0, 0, W, Nout, K, ldw, koff, *dst, total, gamma);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
static int vit_fold(pd_vit *v, float **dst, const float *W, const float *beta, const float *b, int Nout, int K) {
    if (!W || !beta || !b) {
        pd_set_error("pd_vit_create: a weight pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    VIT_TRY(vit_alloc(v, dst, Nout));
    hipLaunchKernelGGL(vit_fold_bias_kernel, dim3((Nout + 127) / 128), dim3(128), 0, 0, W, beta, b, Nout, K, *dst);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

static int vit_rowmajor(pd_vit *v, float **dst, const float *W, int Nout, int K, const float *gamma) {
    const size_t total = (size_t)Nout * K;
    VIT_TRY(vit_alloc(v, dst, total));
    hipLaunchKernelGGL(pd_scale_cols_kernel, dim3(512), dim3(256), 0, 0, W, gamma, K, total, *dst);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

static int vit_grouped(pd_vit *v, unsigned **dst, const float *W, int Nout, int K, const float *gamma) {
    const size_t total = (size_t)(Nout / 32) * (K / 16) * 64;        // one thread per (32-column tile, 16-k step, lane)
    float *p = nullptr;
    VIT_TRY(vit_alloc(v, &p, (size_t)Nout * K));
    *dst = (unsigned *)p;
    hipLaunchKernelGGL(vit_frag_split_kernel, dim3(512), dim3(256), 0, 0, W, gamma, K, total, (uint4 *)p);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

template <typename KernelT>
static int vit_set_lds(KernelT kern, size_t bytes) {
    PD_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PD_OK;
}

extern "C" void pd_vit_destroy(pd_vit *v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipDeviceSynchronize();
    for (void *p : v->allocs) (void)hipFree(p);
    delete v;
}

// The fp16-plane mode of the four Linear layers (round 6; the denoiser's pd_denoiser_build_scales, for dim 384): fp16 keeps 11 bits and five
// exponent bits, so every operand gets a POWER-OF-TWO scale (exact to apply and to undo) fixed here from bounds that hold for every input:
//   * LayerNorm output without affine: sum of squares <= D, so |x^| <= sqrt(384) = 19.6;                       scale 2^floor(log2(32768 / 19.6)) = 2^10
//   * a Linear fed by it: |x^ . w + b| <= sqrt(D) ||w||_2 + |b| (Cauchy-Schwarz) -- the V rows the attention averages (a convex combination:
//     same bound) and the FC1 rows, of which GELU keeps |gelu(v)| <= |v|;                                          scale 2^floor(log2(32768 / bound))
//   * weights (LayerNorm gamma folded): 2^floor(log2(16384 / max |w|)).
// Nothing can overflow (fp16 max 65 504); hi + lo keeps 22 bits.  Non-finite weights have no bound: h_ready stays false and the engine
// runs the exact-fp32 kernels (which propagate inf / NaN like the reference).
static int vit_floor_log2_ratio(double cap, double v) {
    if (!(v > 0.0)) return 0;
    double e = floor(log2(cap / v));
    e = e < -60.0 ? -60.0 : (e > 60.0 ? 60.0 : e);
    return (int)e;
}
static int vit_build_planes_h(pd_vit *v) {
    std::vector<float> w, b;
    bool finite = true;
    auto fetch = [&](const float *Wf, const float *bias, int Nout, int K) -> int {
        w.resize((size_t)Nout * K);
        b.resize(Nout);
        PD_HIP_CHECK(hipMemcpy(w.data(), Wf, w.size() * sizeof(float), hipMemcpyDeviceToHost));
        PD_HIP_CHECK(hipMemcpy(b.data(), bias, b.size() * sizeof(float), hipMemcpyDeviceToHost));
        return PD_OK;
    };
    auto max_abs = [&]() { double m = 0; for (float x : w) { finite = finite && isfinite(x); m = fmax(m, fabs((double)x)); } return m; };
    auto row_bound = [&](int r0, int r1, int K) {            // max over rows of sqrt(D) ||w_r||_2 + |b_r|
        double bound = 0;
        for (int r = r0; r < r1; ++r) {
            double q = 0;
            for (int k = 0; k < K; ++k) q += (double)w[(size_t)r * K + k] * w[(size_t)r * K + k];
            bound = fmax(bound, sqrt((double)VD) * sqrt(q) + fabs((double)b[r]));
            finite = finite && isfinite(q) && isfinite(b[r]);
        }
        return bound;
    };
    auto planes = [&](unsigned **dst, const float *Wf, int Nout, int K, int ew) -> int {
        float *p = nullptr;
        VIT_TRY(vit_alloc(v, &p, (size_t)Nout * K));
        *dst = (unsigned *)p;
        const size_t total = (size_t)(Nout / 32) * (K / 16) * 64;
        hipLaunchKernelGGL(vit_frag_split_kernel, dim3(512), dim3(256), 0, 0, Wf, (const float *)nullptr, K, total, (uint4 *)p, 1, ldexpf(1.0f, ew));   // gamma is in Wf
        PD_HIP_CHECK(hipGetLastError());
        return PD_OK;
    };
    PD_HIP_CHECK(hipDeviceSynchronize());
    const int e_ln = vit_floor_log2_ratio(32768.0, sqrt((double)VD));
    struct E { int qkv, proj, fc1, fc2, ctx, hid; } e[VDEPTH_MAX];
    for (int l = 0; l < v->depth; ++l) {
        pd_vit::Layer &L = v->L[l];
        VIT_TRY(fetch(L.qkv_wf, L.qkv_b, 3 * VD, VD));
        e[l].ctx = vit_floor_log2_ratio(32768.0, row_bound(2 * VD, 3 * VD, VD));
        e[l].qkv = vit_floor_log2_ratio(16384.0, max_abs());
        VIT_TRY(fetch(L.proj_wf, L.proj_b, VD, VD));
        e[l].proj = vit_floor_log2_ratio(16384.0, max_abs());
        VIT_TRY(fetch(L.fc1_wf, L.fc1_b, VFF, VD));
        e[l].hid = vit_floor_log2_ratio(32768.0, row_bound(0, VFF, VD));
        e[l].fc1 = vit_floor_log2_ratio(16384.0, max_abs());
        VIT_TRY(fetch(L.fc2_wf, L.fc2_b, VD, VFF));
        e[l].fc2 = vit_floor_log2_ratio(16384.0, max_abs());
        if (!finite) return PD_OK;                 // (not an error: the exact-fp32 kernels take such a network)
    }
    for (int l = 0; l < v->depth; ++l) {
        pd_vit::Layer &L = v->L[l];
        VIT_TRY(planes(&L.qkv_wh, L.qkv_wf, 3 * VD, VD, e[l].qkv));
        VIT_TRY(planes(&L.proj_wh, L.proj_wf, VD, VD, e[l].proj));
        VIT_TRY(planes(&L.fc1_wh, L.fc1_wf, VFF, VD, e[l].fc1));
        VIT_TRY(planes(&L.fc2_wh, L.fc2_wf, VD, VFF, e[l].fc2));
        L.qkv_cs = ldexpf(1.0f, -(e_ln + e[l].qkv));
        L.proj_cs = ldexpf(1.0f, -(e[l].ctx + e[l].proj));
        L.fc1_cs = ldexpf(1.0f, -(e_ln + e[l].fc1));
        L.fc2_cs = ldexpf(1.0f, -(e[l].hid + e[l].fc2));
        L.ctx_scale = ldexpf(1.0f, e[l].ctx);
        L.hid_scale = ldexpf(1.0f, e[l].hid);
    }
    v->ln_scale = ldexpf(1.0f, e_ln);
    v->h_ready = true;
    return PD_OK;
}

extern "C" int pd_vit_create(const pd_vit_weights *w, pd_vit **out) {
    if (!w || !out) {
        pd_set_error("pd_vit_create: NULL argument");
        return PD_ERR_INVALID_ARG;
    }
    if (w->dim != VD || w->num_heads != VH || w->mlp_hidden != VFF || w->patch_size != VP || w->depth < 1 || w->depth > VDEPTH_MAX ||
        w->pos_grid < 1 || w->pos_grid > 15) {
        pd_set_error("pd_vit_create: unsupported ViT shape (built for dim 384, 6 heads, MLP 1536, patch 16, <= %d blocks, position grid <= 15)",
                     VDEPTH_MAX);
        return PD_ERR_UNSUPPORTED;
    }
    pd_vit *v = new pd_vit();
    *out = nullptr;
    PD_HIP_CHECK(hipGetDevice(&v->device));
    v->depth = w->depth;
    v->grid0 = w->pos_grid;
    int rc = PD_OK;
    do {
        if ((rc = vit_pack(v, &v->patch_wp, w->patch_w, VD, VKP, VKP, 0, nullptr))) break;
        if ((rc = vit_copy(v, &v->patch_b, w->patch_b, VD))) break;
        if ((rc = vit_copy(v, &v->cls, w->cls_token, VD))) break;
        if ((rc = vit_copy(v, &v->pos, w->pos_embed, (size_t)(1 + w->pos_grid * w->pos_grid) * VD))) break;
        if ((rc = vit_copy(v, &v->norm_w, w->norm_w, VD))) break;
        if ((rc = vit_copy(v, &v->norm_b, w->norm_b, VD))) break;
        if ((rc = vit_alloc(v, &v->zero_b, VD))) break;
        if (hipMemset(v->zero_b, 0, VD * sizeof(float)) != hipSuccess) {
            rc = PD_ERR_HIP;
            break;
        }
        for (int l = 0; l < w->depth && !rc; ++l) {
            const pd_vit_layer_weights &s = w->layers[l];
            pd_vit::Layer &L = v->L[l];
            // LayerNorm affine folded: W' = W diag(gamma), b' = b + W beta
            if ((rc = vit_pack(v, &L.qkv_wp, s.qkv_w, 3 * VD, VD, VD, 0, s.norm1_w))) break;
            if ((rc = vit_fold(v, &L.qkv_b, s.qkv_w, s.norm1_b, s.qkv_b, 3 * VD, VD))) break;
            if ((rc = vit_pack(v, &L.proj_wp, s.proj_w, VD, VD, VD, 0, nullptr))) break;
            if ((rc = vit_copy(v, &L.proj_b, s.proj_b, VD))) break;
            if ((rc = vit_pack(v, &L.fc1_wp, s.fc1_w, VFF, VD, VD, 0, s.norm2_w))) break;
            if ((rc = vit_fold(v, &L.fc1_b, s.fc1_w, s.norm2_b, s.fc1_b, VFF, VD))) break;
            if ((rc = vit_pack(v, &L.fc2a_wp, s.fc2_w, VD, VKP, VFF, 0, nullptr))) break;     // K columns [0, 768)
            if ((rc = vit_pack(v, &L.fc2b_wp, s.fc2_w, VD, VKP, VFF, VKP, nullptr))) break;   // K columns [768, 1536)
            if ((rc = vit_copy(v, &L.fc2_b, s.fc2_b, VD))) break;
            if ((rc = vit_rowmajor(v, &L.qkv_wf, s.qkv_w, 3 * VD, VD, s.norm1_w))) break;
            if ((rc = vit_rowmajor(v, &L.proj_wf, s.proj_w, VD, VD, nullptr))) break;
            if ((rc = vit_rowmajor(v, &L.fc1_wf, s.fc1_w, VFF, VD, s.norm2_w))) break;
            if ((rc = vit_rowmajor(v, &L.fc2_wf, s.fc2_w, VD, VFF, nullptr))) break;
            if ((rc = vit_grouped(v, &L.qkv_ws, s.qkv_w, 3 * VD, VD, s.norm1_w))) break;
            if ((rc = vit_grouped(v, &L.proj_ws, s.proj_w, VD, VD, nullptr))) break;
            if ((rc = vit_grouped(v, &L.fc1_ws, s.fc1_w, VFF, VD, s.norm2_w))) break;
            if ((rc = vit_grouped(v, &L.fc2_ws, s.fc2_w, VD, VFF, nullptr))) break;
        }
        if (rc) break;
        if ((rc = vit_set_lds(vit_gemm_kernel<VKP, 3, 4>, 32 * (VKP + 4) * 4))) break;
        if ((rc = vit_set_lds(vit_gemm_kernel<VD, 1, 0>, 32 * (VD + 4) * 4))) break;
        if ((rc = vit_set_lds(vit_gemm_kernel<VD, 0, 2>, 32 * (VD + 4) * 4))) break;
        if ((rc = vit_set_lds(vit_gemm_kernel<VD, 1, 3>, 32 * (VD + 4) * 4))) break;
        if ((rc = vit_set_lds(vit_gemm_kernel<VKP, 0, 2>, 32 * (VKP + 4) * 4))) break;
        if ((rc = vit_set_lds(vit_attn_kernel<0>, vit_attn_lds(VT_MAX)))) break;
        if ((rc = vit_set_lds(vit_attn_kernel<1>, vit_attn_lds(VT_MAX)))) break;
        if ((rc = vit_set_lds(vit_attn_kernel<2>, vit_attn_lds(VT_MAX)))) break;
        if ((rc = vit_build_planes_h(v))) break;
        if (hipDeviceSynchronize() != hipSuccess) rc = PD_ERR_HIP;
    } while (0);
    if (rc) {
        pd_vit_destroy(v);
        return rc;
    }
    *out = v;
    return PD_OK;
}

static int vit_reserve(pd_vit *v, size_t tokens, size_t pixels) {
    auto grow = [&](float **p, size_t n) -> int {
        if (*p) {
            (void)hipFree(*p);
            for (auto &q : v->allocs)
                if (q == *p) q = nullptr;
        }
        return vit_alloc(v, p, n);
    };
    if (tokens > v->cap_tokens) {
        PD_HIP_CHECK(hipDeviceSynchronize());
        VIT_TRY(grow(&v->x, tokens * VD));
        VIT_TRY(grow(&v->xn, tokens * VD));
        VIT_TRY(grow(&v->qkv, tokens * 3 * VD));
        VIT_TRY(grow(&v->ctx, tokens * VD));
        VIT_TRY(grow(&v->hid, tokens * VFF));
        v->cap_tokens = tokens;
    }
    if (pixels > v->cap_pixels) {
        PD_HIP_CHECK(hipDeviceSynchronize());
        VIT_TRY(grow(&v->img, pixels));
        v->cap_pixels = pixels;
    }
    return PD_OK;
}

template <int K, int AMODE, int EPI>
static void vit_gemm(const VitGemmArgs &g, hipStream_t s) {
    const int MT = (g.M + 31) / 32;
    hipLaunchKernelGGL((vit_gemm_kernel<K, AMODE, EPI>), dim3(MT * (g.Nout / 32)), dim3(256), 32 * (K + 4) * 4, s, g);
}

// split precision (pd_gemm_split.h): 128 x 128 tiles where the column count gives enough of them, 128 x 64 for the 384-wide outputs
#define vit_gemm_split pd_gemm_split

extern "C" int pd_vit_set_option(pd_vit *v, int option, int value) {
    if (!v || option != PD_VIT_OPT_EXACT_FP32 || value < 0 || value > 2) {
        pd_set_error("pd_vit_set_option: unknown option %d / value %d", option, value);
        return PD_ERR_INVALID_ARG;
    }
    v->exact_fp32 = value;
    return PD_OK;
}

// one scale: images [n,3,H,W] in [0,1] -> z (+)= norm(ViT(prep(images)))[:, 0] * weight.  pos_scaled: the position table of
// this token grid [1 + gh*gw, 384] (DEVICE), resampled by the caller (bicubic, as DINO's interpolate_pos_encoding does).
extern "C" int pd_vit_forward_scale(pd_vit *v, const float *images, int n_img, int H, int W, double scale_factor,
                                    const float *pos_scaled, float weight, int accumulate, float *z_out, void *stream) {
    if (!v || !images || !z_out || n_img <= 0 || H < VP || W < VP || !(scale_factor > 0.0)) {
        pd_set_error("pd_vit_forward_scale: invalid arguments (n=%d H=%d W=%d scale=%g)", n_img, H, W, scale_factor);
        return PD_ERR_INVALID_ARG;
    }
    const bool identity = scale_factor == 1.0;
    const int Hs = identity ? H : (int)floor((double)H * scale_factor), Ws = identity ? W : (int)floor((double)W * scale_factor);   // torch: floor(size * scale) in double
    const int gh = Hs / VP, gw = Ws / VP, P = gh * gw, T = P + 1;
    if (gh < 1 || gw < 1 || T > VT_MAX) {
        pd_set_error("pd_vit_forward_scale: %d x %d pixels give %d tokens per image (1..%d supported)", Hs, Ws, T, VT_MAX);
        return PD_ERR_UNSUPPORTED;
    }
    const bool native = (gh == v->grid0 && gw == v->grid0);
    if (!native && !pos_scaled) {
        pd_set_error("pd_vit_forward_scale: a %d x %d token grid needs the resampled position table (pos_scaled)", gh, gw);
        return PD_ERR_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t tokens = (size_t)n_img * T;
    VIT_TRY(vit_reserve(v, tokens, (size_t)n_img * 3 * Hs * Ws));
    const float *pos = native && !pos_scaled ? v->pos : pos_scaled;
    {
        const size_t total = (size_t)n_img * 3 * Hs * Ws;
        hipLaunchKernelGGL(vit_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, images, n_img, H, W, Hs, Ws,
                           (float)(1.0 / scale_factor), identity ? 1 : 0, v->img);      // torch: float(1 / scale_factor) as the coordinate scale
    }
    VitGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.T = T; g.gh = gh; g.gw = gw; g.Hs = Hs; g.Ws = Ws;
    // patch embedding + position, CLS rows
    g.img = v->img; g.Wp = v->patch_wp; g.bias = v->patch_b; g.pos = pos; g.C = v->x; g.M = n_img * P; g.Nout = VD; g.lda = VKP;
    vit_gemm<VKP, 3, 4>(g, s);
    hipLaunchKernelGGL(vit_cls_kernel, dim3((n_img * VD + 255) / 256), dim3(256), 0, s, v->cls, pos, n_img, T, v->x);
    g.M = (int)tokens;
    const size_t attn_lds = vit_attn_lds(T);
    const int nqb = (T + 31) / 32;
    const bool streamed = (int)tokens >= PD_STREAM_MIN_ROWS;
    for (int l = 0; l < v->depth; ++l) {
        const pd_vit::Layer &L = v->L[l];
        if (streamed && v->exact_fp32 == 0 && v->h_ready) {
            // fp16-plane mode (round 6, the default): LayerNorm, softmax, QK^T, PV, residuals in fp32; the four Linear layers multiply fp16 hi + lo
            // operand pairs (22 bits, static power-of-two scales) with fp32 accumulation; exact erf GELU in FC1's epilogue
            const int M = (int)tokens;
            hipLaunchKernelGGL((pd_ln_rows_kernel<VD, 2>), dim3((M + 3) / 4), dim3(256), 0, s, v->x, v->xn, M, 1e-6f, v->ln_scale);
            pd_gemm_strip<0, 2, true, 1, true>((const unsigned *)v->xn, VD, L.qkv_wh, VD, L.qkv_b, v->qkv, M, 3 * VD, s, L.qkv_cs);
            hipLaunchKernelGGL(vit_attn_kernel<2>, dim3(n_img * VH * nqb), dim3(256), attn_lds, s, v->qkv, v->ctx, T, nqb, L.ctx_scale);
            pd_gemm_strip<2, 2, true, 1, true>((const unsigned *)v->ctx, VD, L.proj_wh, VD, L.proj_b, v->x, M, VD, s, L.proj_cs);
            hipLaunchKernelGGL((pd_ln_rows_kernel<VD, 2>), dim3((M + 3) / 4), dim3(256), 0, s, v->x, v->xn, M, 1e-6f, v->ln_scale);
            pd_gemm_strip<3, 2, true, 1, true>((const unsigned *)v->xn, VD, L.fc1_wh, VD, L.fc1_b, v->hid, M, VFF, s, L.fc1_cs, L.hid_scale);
            pd_gemm_strip<2, 2, true, 1, true>((const unsigned *)v->hid, VFF, L.fc2_wh, VFF, L.fc2_b, v->x, M, VD, s, L.fc2_cs);
            continue;
        }
        if (streamed && v->exact_fp32 == 2) {
            const int M = (int)tokens;
            hipLaunchKernelGGL((pd_ln_rows_kernel<VD, 1>), dim3((M + 3) / 4), dim3(256), 0, s, v->x, v->xn, M, 1e-6f, 1.0f);
            vit_gemm_split<0, 2, 2>((const unsigned *)v->xn, VD, L.qkv_ws, VD, L.qkv_b, v->qkv, M, 3 * VD, s);
            hipLaunchKernelGGL(vit_attn_kernel<1>, dim3(n_img * VH * nqb), dim3(256), attn_lds, s, v->qkv, v->ctx, T, nqb, 1.0f);
            vit_gemm_split<2, 2, 1>((const unsigned *)v->ctx, VD, L.proj_ws, VD, L.proj_b, v->x, M, VD, s);
            hipLaunchKernelGGL((pd_ln_rows_kernel<VD, 1>), dim3((M + 3) / 4), dim3(256), 0, s, v->x, v->xn, M, 1e-6f, 1.0f);
            vit_gemm_split<3, 2, 2>((const unsigned *)v->xn, VD, L.fc1_ws, VD, L.fc1_b, v->hid, M, VFF, s);
            vit_gemm_split<2, 2, 1>((const unsigned *)v->hid, VFF, L.fc2_ws, VFF, L.fc2_b, v->x, M, VD, s);
            continue;
        }
        if (streamed) {
            const int M = (int)tokens;
            hipLaunchKernelGGL((pd_ln_rows_kernel<VD, 0>), dim3((M + 3) / 4), dim3(256), 0, s, v->x, v->xn, M, 1e-6f, 1.0f);
            pd_gemm_stream<0>(v->xn, VD, L.qkv_wf, VD, L.qkv_b, v->qkv, M, 3 * VD, s);
            hipLaunchKernelGGL(vit_attn_kernel<0>, dim3(n_img * VH * nqb), dim3(256), attn_lds, s, v->qkv, v->ctx, T, nqb, 1.0f);
            pd_gemm_stream<2>(v->ctx, VD, L.proj_wf, VD, L.proj_b, v->x, M, VD, s);
            hipLaunchKernelGGL((pd_ln_rows_kernel<VD, 0>), dim3((M + 3) / 4), dim3(256), 0, s, v->x, v->xn, M, 1e-6f, 1.0f);
            pd_gemm_stream<3>(v->xn, VD, L.fc1_wf, VD, L.fc1_b, v->hid, M, VFF, s);
            pd_gemm_stream<2>(v->hid, VFF, L.fc2_wf, VFF, L.fc2_b, v->x, M, VD, s);
            continue;
        }
        g.A = v->x; g.lda = VD; g.Wp = L.qkv_wp; g.bias = L.qkv_b; g.C = v->qkv; g.Nout = 3 * VD;
        vit_gemm<VD, 1, 0>(g, s);
        hipLaunchKernelGGL(vit_attn_kernel<0>, dim3(n_img * VH * nqb), dim3(256), attn_lds, s, v->qkv, v->ctx, T, nqb, 1.0f);
        g.A = v->ctx; g.lda = VD; g.Wp = L.proj_wp; g.bias = L.proj_b; g.C = v->x; g.Nout = VD;
        vit_gemm<VD, 0, 2>(g, s);
        g.A = v->x; g.lda = VD; g.Wp = L.fc1_wp; g.bias = L.fc1_b; g.C = v->hid; g.Nout = VFF;
        vit_gemm<VD, 1, 3>(g, s);
        g.A = v->hid; g.lda = VFF; g.Wp = L.fc2a_wp; g.bias = L.fc2_b; g.C = v->x; g.Nout = VD;
        vit_gemm<VKP, 0, 2>(g, s);
        g.A = v->hid + VKP; g.Wp = L.fc2b_wp; g.bias = v->zero_b;
        vit_gemm<VKP, 0, 2>(g, s);
    }
    hipLaunchKernelGGL(vit_final_kernel, dim3(n_img), dim3(64), 0, s, v->x, T, v->norm_w, v->norm_b, weight, accumulate, z_out);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
