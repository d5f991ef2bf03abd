// pd_denoiser_dev.h -- shapes, device-side weight tables and small wave helpers shared by the
// denoiser kernels (pd_denoiser.hip, pd_gemm_stream.h).
#pragma once
#include "pd_internal.h"

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DM 512          // d_model
#define NH 4            // heads
#define DH 128          // head dim
#define DFF 1024        // feed-forward dim
#define ZD 384          // z_dim
#define KFIRST 702      // 189 + 128 + 384 + 1   (denoiser.py:39)
// At >= PD_STREAM_MIN_ROWS token rows `_first` (denoiser.py:56-70) is evaluated in THREE pieces (round 5): of its 702 input columns only the 189 pose-embedding columns
// change from one diffusion step to the next --
//   reference column : [0,180) harmonic | [180,189) x | [189,317) t_emb | [317,701) z | 701 pivot  (denoiser.py:68)
//   PD_FIRST_Z   z columns [317,701), K = 384: zproj[m] = z[m] W_z^T + b_first, ONCE per sampling call (z is the same in all T steps)
//   PD_FIRST_T   t_emb columns [189,317): a [T, 512] table W_t t_emb(t), built at engine creation (t_emb depends on t only)
//   PD_FIRST_D   the step's own columns in the engine's order [0,180) harmonic | [180,189) x | 189 pivot | 190,191 pad: K = 192
// so a step's `_first` is h = zproj + ttab[t] + D W_d^T: K 704 -> 192 inside the loop.
#define KFIRST_PAD 704
// Below PD_STREAM_MIN_ROWS token rows _first stays one launch over all 702 columns, its K axis permuted so the wide pieces land 16-byte
// aligned in LDS:  engine column k' : [0,384) z | [384,512) t_emb | [512,692) harmonic | [692,701) x | 701 pivot | 702,703 pad
__host__ __device__ inline int pd_first_col_all(int kp) {
    if (kp < 384) return 317 + kp;
    if (kp < 512) return 189 + (kp - 384);
    if (kp < 692) return kp - 512;
    if (kp < 701) return 180 + (kp - 692);
    return kp;   // 701 pivot; 702/703 are padding (>= KFIRST -> zero)
}
#define PD_FIRST_D 1
#define PD_FIRST_Z 2
#define PD_FIRST_T 3
#define KFIRST_D 192
// column of the reference's _first.weight behind column kp of piece `piece` (>= KFIRST: padding -> zero)
__host__ __device__ inline int pd_first_col(int piece, int kp) {
    if (piece == PD_FIRST_Z) return kp < ZD ? 317 + kp : KFIRST;
    if (piece == PD_FIRST_T) return kp < 128 ? 189 + kp : KFIRST;
    if (kp < 189) return kp;
    return kp == 189 ? 701 : KFIRST;   // pivot; 190 / 191 are padding
}
#define HID 128         // mlp_hidden_dim

struct PdLayerDev {          // [0] = 32-wide-tile packing, [1] = 16-wide-tile packing of the same weights
    float *qkv_wp[2], *qkv_b;  // LayerNorm-1 gamma folded into the columns, beta into the bias
    float *out_wp[2], *out_b;
    float *ff1_wp[2], *ff1_b;  // LayerNorm-2 folded likewise
    float *ff2_wp[2], *ff2_b;
    float *qkv_wf, *out_wf, *ff1_wf, *ff2_wf;   // row-major copies (LayerNorm scale folded) for the streamed GEMM at >= 1024 token rows
    unsigned *qkv_ws, *out_ws, *ff1_ws, *ff2_ws;   // split into bf16 hi / lo in MFMA fragment order (pd_gemm_split.h): the fast mode, built on demand
    // the fp16-plane mode (PD_OPT_DENOISER_SPLIT = 2): fp16 hi / lo of w * 2^ew in the same order, and the power-of-two scales of
    // pd_denoiser_build_split: accumulator scales 2^-(ea + ew) per GEMM, operand scales of the attention output and the FF hidden rows
    unsigned *qkv_wh, *out_wh, *ff1_wh, *ff2_wh;
    float qkv_cs, out_cs, ff1_cs, ff2_cs, ctx_scale, ff_scale;
    int e_wqkv, e_wo, e_w1, e_w2;                  // the weights' scale exponents (pd_denoiser_build_scales)
};

struct PdDenoiserDev {
    int num_layers = 0, timesteps = 0, m_cap = 0;
    float *t_table = nullptr;          // [T,128] time embeddings
    float *first_wp[2] = {nullptr, nullptr}, *first_b = nullptr;   // _first packed for the small-batch kernel (K = 704, pd_first_col_all)
    float *ttab = nullptr;             // [T, 512] = W_t t_emb(t): the time piece of _first, added as the step GEMM's bias
    float *zproj = nullptr;            // [rows, 512] = z W_z^T + b_first of the sampling call in flight (pd_denoiser_prepare)
    PdLayerDev layers[PD_MAX_LAYERS];
    float *last0_wp[2] = {nullptr, nullptr}, *last0_b = nullptr, *last_ln_w = nullptr, *last_ln_b = nullptr;
    float *last3_w = nullptr, *last3_b = nullptr;   // [9,128] plain
    float *h = nullptr, *qkv = nullptr, *ctx = nullptr, *ff = nullptr, *hid = nullptr;   // activations [rows, .]
    float *hn = nullptr;               // LayerNorm(h) without affine, streamed path only
    float *emb = nullptr, *first_df = nullptr, *first_zf = nullptr, *last0_wf = nullptr;   // streamed path: _first's step rows [rows, 192], row-major _first pieces / _last.0 weights
    bool split_ready = false;          // the fast mode's split weights exist
    bool split_h_ready = false;        // the fp16-plane mode's weights exist
    bool scales_ready = false;         // the fp16-plane scales exist (pd_denoiser_build_scales)
    bool non_finite = false;           // pd_denoiser_build_scales met inf / NaN in an encoder weight or bias: the ONLY failure pd_engine_create downgrades on
    std::vector<void *> allocs;
};

// 8-lane (one activation row) sum on the DPP network: xor-1, xor-2 quad permutes + half-row mirror
template <int CTRL>
__device__ __forceinline__ float pd_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float pd_sum8(float v) {
    v = pd_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = pd_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    return pd_dpp_add<0x141>(v);   // row_half_mirror
}

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

