// pd_denoiser_xcd.hip -- the denoiser + DDPM update as ONE persistent launch for a whole range of
// diffusion steps, organised per XCD (accelerator die) of the MI355X.
//
// Same arithmetic and reference lines as pd_denoiser.hip (models/denoiser.py:53-98,
// util/embedding.py:13-50, models/gaussian_diffuser.py:190-209,:280); what changes is the schedule:
//   * sequence s lives on XCD s % 8 for the whole launch: every activation of a sequence is produced
//     and consumed by CUs that share ONE L2, so phases hand data over through that L2 (plain stores,
//     s_waitcnt, an L2 atomic counter, `buffer_inv sc0` on the consumer) -- no kernel boundaries, no
//     cross-die coherence traffic for activations.  Measured (tools/xcd_sync_probe.hip): a launch of 8*W
//     workgroups puts W on every XCD; a barrier among 32 workgroups of one XCD costs 0.72 us against
//     ~4.5 us for the cheapest dependent kernel launch inside a hipGraph;
//   * the 43 launches of a step become 43 phases of one kernel; a launch covers any number of steps
//     (the 90 unguided steps of a pass are one launch);
//   * every XCD streams the weights itself (they sit in the 256 MiB Infinity Cache); a workgroup owns
//     N-tiles w, w + W, ... of every GEMM, keeps the activation tile in LDS across its N-tiles and
//     issues the weight fragments of its first tile BEFORE it waits on the barrier.
// The price: a 20-frame sequence fills only 20 of the 32 rows of its XCD's MFMA tile.
#include "pd_denoiser_dev.h"

#include <math.h>
#include <string.h>

#define XCD_COUNT 8
#define XCD_RED_FLOATS (4 * 8 * 64)
#define XCD_LDS_BYTES ((32 * (DFF + 4) + XCD_RED_FLOATS) * 4)
#define XCD_SPIN_LIMIT (1u << 22)

struct XcdLayer {
    const float *qkv_wp, *qkv_b, *out_wp, *out_b, *ff1_wp, *ff1_b, *ff2_wp, *ff2_b;
};
struct XcdArgs {
    const float *first_wp, *first_b;
    XcdLayer L[PD_MAX_LAYERS];
    const float *last0_wp, *last0_b, *lnw, *lnb, *w3, *b3;
    const float *t_table, *sched;      // [T,128]; [T,8] = c_recip, c_recipm1, coef1, coef2, sigma
    float *h, *qkv, *ctx, *ff, *hid;   // activations, XCD x owns rows [x * cap_x, (x + 1) * cap_x)
    const float *z, *noise;            // [B,N,384]; [T+1,B,N,9]
    float *proc;                       // [T+1,B,N,9]: step k reads slot k and writes slot k + 1
    unsigned *bar, *err;               // bar: [8][32] words, [x][0] arrival counter, [x][1] its value at launch start, [x][2] tickets
    int num_layers, cap_x, B, N, W, T, step_begin, step_end, use_noise;
};

struct XcdCtx {
    int x, wx, W, Mx, N;
    unsigned *cnt, *err;
    unsigned base, phase;
};

__device__ __forceinline__ unsigned pd_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// wait until every workgroup of this XCD has finished `c.phase` phases, then drop stale L1 lines
__device__ __forceinline__ bool xcd_wait(XcdCtx &c, int *flag) {
    if (c.phase == 0) return true;   // nothing of this launch to wait for (inputs come from earlier kernels)
    if (threadIdx.x == 0) {
        const unsigned target = c.base + (unsigned)c.W * c.phase;
        int ok = 0;
        for (unsigned spin = 0; spin < XCD_SPIN_LIMIT; ++spin) {
            const unsigned v = __hip_atomic_load(c.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - target) >= 0) {
                ok = 1;
                break;
            }
            if ((spin & 1023u) == 1023u && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) atomicOr(c.err, 4u);
        *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    asm volatile("buffer_inv sc0" ::: "memory");
    return ok;
}
// every store of this workgroup has reached the L2 -> count this workgroup as done with the phase
__device__ __forceinline__ void xcd_arrive(XcdCtx &c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++c.phase;
}

// local row of this XCD -> global token index (sequence x + 8 * (row / N), frame row % N)
__device__ __forceinline__ int xcd_token(const XcdCtx &c, int row) {
    const int sl = row / c.N, n = row - sl * c.N;
    return (c.x + XCD_COUNT * sl) * c.N + n;
}

// stage 32 activation rows [m0, m0 + 32) of this XCD into LDS (row stride K + 4), same three modes as
// pd_gemm_kernel: 0 plain, 1 LayerNorm without affine (folded into the weights), 2 embedding build
template <int K, int AMODE>
__device__ __forceinline__ void xcd_stage(const XcdCtx &c, const float *A, const float *xg, const float *zg, const float *temb,
                                          int m0, float *As) {
    constexpr int LDA = K + 4;
    const int tid = threadIdx.x;
    const int r = tid >> 3, sub = tid & 7;
    const int m = m0 + r;
    const bool live = m < c.Mx;
    const int mr = live ? m : c.Mx - 1;   // clamp: padded rows load a valid row and are zeroed
    float *dst = As + r * LDA;
    if constexpr (AMODE == 2) {
        const int tok = xcd_token(c, mr);
        const float4 *zr = (const float4 *)(zg + (size_t)tok * ZD);
        const float4 *te = (const float4 *)temb;
        float4 zv[ZD / 32], tv[4];
#pragma unroll
        for (int i = 0; i < ZD / 32; ++i) zv[i] = zr[sub + 8 * i];
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = te[sub + 8 * i];
        float xv[9];
#pragma unroll
        for (int d = 0; d < 9; ++d) xv[d] = xg[(size_t)tok * 9 + d];
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
            dst[701] = (live && (mr % c.N == 0)) ? 1.0f : 0.0f;   // pivot one-hot on frame 0 (denoiser.py:62-66)
            dst[702] = 0.0f;
            dst[703] = 0.0f;
        }
    } else if constexpr (AMODE == 1) {
        static_assert(AMODE != 1 || K == 512, "LayerNorm staging is built for d_model = 512");
        float4 v[K / 32];
        const float4 *src = (const float4 *)(A + (size_t)mr * K);
#pragma unroll
        for (int i = 0; i < K / 32; ++i) v[i] = src[sub + 8 * i];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < K / 32; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = pd_sum8(s) * (1.0f / K);
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < K / 32; ++i) {
            const float a = v[i].x - mean, b2 = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b2 * b2) + (cc * cc + d * d);
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
        const float4 *src = (const float4 *)(A + (size_t)mr * K);
        const float keep = live ? 1.0f : 0.0f;
        constexpr int NV = K / 32;
        constexpr int VB = 16;
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

// one GEMM phase of this XCD:  C[m, n] = epi( sum_k A'[m, k] W[n, k] + bias[n] ) for the Mx rows of the XCD.
// 16-wide N-tiles (v_mfma_f32_16x16x4_f32, two 16-row tiles share each weight fragment), split-K over the 4 waves.
template <int K, int AMODE, int EPI>
__device__ __forceinline__ bool xcd_gemm(XcdCtx &c, const float *A, const float *Wp, const float *bias, float *C, int Nout,
                                         const float *xg, const float *zg, const float *temb, float *As, float *red, int *flag) {
    constexpr int LDA = K + 4;
    constexpr int KC = K / 16;
    constexpr int CPW = KC / 4;
    static_assert(KC % 4 == 0 && CPW <= 16, "split-K chunking");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = Nout / 16, MTx = (c.Mx + 31) / 32;
    const bool has_work = c.wx < ntiles;
    // weight fragments of the first tile go in flight before the barrier wait: they do not depend on it
    float4 w[CPW];
    {
        const float4 *wp = (const float4 *)Wp + ((size_t)(has_work ? c.wx : 0) * KC + (size_t)wave * CPW) * 64 + lane;
#pragma unroll
        for (int q = 0; q < CPW; ++q) w[q] = wp[(size_t)q * 64];
    }
    if (!xcd_wait(c, flag)) return false;
    if (has_work) {
        for (int mt = 0; mt < MTx; ++mt) {
            const int m0 = mt * 32;
            xcd_stage<K, AMODE>(c, A, xg, zg, temb, m0, As);
            __syncthreads();
            for (int nt = c.wx; nt < ntiles; nt += c.W) {
                if (mt != 0 || nt != c.wx) {
                    const float4 *wp = (const float4 *)Wp + ((size_t)nt * KC + (size_t)wave * CPW) * 64 + lane;
#pragma unroll
                    for (int q = 0; q < CPW; ++q) w[q] = wp[(size_t)q * 64];
                }
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float *arow = As + (lane & 15) * LDA + wave * CPW * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int q = 0; q < CPW; ++q) {
                    const float4 wf = w[q];
                    const float4 a0 = *(const float4 *)(arow + q * 16);
                    const float4 a1 = *(const float4 *)(arow + 16 * LDA + q * 16);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wf.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wf.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wf.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wf.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wf.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wf.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wf.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wf.w, acc1, 0, 0, 0);
                }
                // cross-wave reduction in fixed order + fused epilogue (same order as pd_gemm_kernel<.., 16>)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    red[(wave * 8 + i) * 64 + lane] = acc0[i];
                    red[(wave * 8 + 4 + i) * 64 + lane] = acc1[i];
                }
                __syncthreads();
                const int col = nt * 16 + (lane & 15);
                const float bv = bias[col];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int reg = wave * 2 + i;
                    float v = red[(0 * 8 + reg) * 64 + lane];
                    v += red[(1 * 8 + reg) * 64 + lane];
                    v += red[(2 * 8 + reg) * 64 + lane];
                    v += red[(3 * 8 + reg) * 64 + lane];
                    v += bv;
                    const int row = m0 + 16 * (reg >> 2) + 4 * (lane >> 4) + (reg & 3);
                    if (row < c.Mx) {
                        float *cp = C + (size_t)row * Nout + col;
                        if constexpr (EPI == 1) v = fmaxf(v, 0.0f);
                        if constexpr (EPI == 2) v += *cp;
                        *cp = v;
                    }
                }
                __syncthreads();   // `red` (and, for the next M tile, `As`) may be rewritten
            }
        }
    }
    xcd_arrive(c);
    return true;
}

// attention core of one (local sequence, head, 4 query rows) unit; same arithmetic as pd_attn_kernel
__device__ __forceinline__ bool xcd_attention(XcdCtx &c, const float *qkv, float *ctx, float *lds, int *flag) {
    constexpr int LD = DH + 4;
    const int N = c.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *Kk = lds, *V = Kk + N * LD, *Q = V + N * LD, *P = Q + 4 * LD;
    if (!xcd_wait(c, flag)) return false;
    const int RQ = (N + 3) / 4, ns = c.Mx / N;
    const int units = ns * NH * RQ;
    const float scale = 0.08838834764831845f;   // 1/sqrt(128)
    for (int u = c.wx; u < units; u += c.W) {
        const int rq = u % RQ, h = (u / RQ) % NH, sl = u / (RQ * NH);
        const int i = rq * 4 + wave;
        const float *base = qkv + (size_t)sl * N * (3 * DM) + h * DH;
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
            const float4 a = qa[d], k4 = kb[d];
            s = fmaf(a.x, k4.x, s);
            s = fmaf(a.y, k4.y, s);
            s = fmaf(a.z, k4.z, s);
            s = fmaf(a.w, k4.w, s);
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
            float *out = ctx + (size_t)(sl * N + i) * DM + h * DH;
            out[lane] = o0;
            out[64 + lane] = o1;
        }
        __syncthreads();   // K / V / Q / P are rewritten by the next unit
    }
    xcd_arrive(c);
    return true;
}

// LayerNorm(128) -> ReLU -> Linear(128 -> 9) + DDPM update, one wave per token (= pd_tail_kernel)
__device__ __forceinline__ bool xcd_tail(XcdCtx &c, const XcdArgs &a, const float *hid, const float *xg, const float *nz, float *xn,
                                         const float *sc, int *flag) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!xcd_wait(c, flag)) return false;
    const float c_recip = sc[0], c_recipm1 = sc[1], coef1 = sc[2], coef2 = sc[3], sigma = sc[4];
    for (int m = c.wx * 4 + wave; m < c.Mx; m += 4 * c.W) {
        const float *row = hid + (size_t)m * HID;
        const float v0 = row[lane], v1 = row[64 + lane];
        const float mean = pd_wave_sum(v0 + v1) * (1.0f / HID);
        const float d0 = v0 - mean, d1 = v1 - mean;
        const float rstd = 1.0f / sqrtf(pd_wave_sum(d0 * d0 + d1 * d1) * (1.0f / HID) + 1e-5f);
        const float a0 = fmaxf(d0 * rstd * a.lnw[lane] + a.lnb[lane], 0.0f);
        const float a1 = fmaxf(d1 * rstd * a.lnw[64 + lane] + a.lnb[64 + lane], 0.0f);
        float e = 0.0f;
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const float part = pd_wave_sum(fmaf(a0, a.w3[o * HID + lane], a1 * a.w3[o * HID + 64 + lane]));
            e = (lane == o) ? part : e;
        }
        if (lane < 9) {
            e += a.b3[lane];
            const size_t at = (size_t)xcd_token(c, m) * 9 + lane;
            const float xv = xg[at];
            const float x0 = c_recip * xv - c_recipm1 * e;          // gaussian_diffuser.py:190-194
            const float mu = coef1 * x0 + coef2 * xv;               // :201-205
            xn[at] = nz ? mu + sigma * nz[at] : mu;                 // :280 (guided steps: the mean, noise = 0, :272-276)
        }
    }
    xcd_arrive(c);
    return true;
}

__global__ __launch_bounds__(256) void pd_den_xcd_kernel(XcdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int flag;
    __shared__ unsigned base_s;
    float *As = lds, *red = lds + 32 * (DFF + 4);
    // Which XCD am I on?  A launch of 8*W workgroups puts exactly W on each XCD (round-robin dispatch), but the
    // round-robin pointer carries over from the previous launch, so workgroup b is NOT always on XCD b % 8: read the
    // hardware id and take a ticket for the index within the XCD (W consecutive tickets mod W = 0..W-1 in any launch).
    __shared__ int ids[2];
    if (threadIdx.x == 0) {
        const int x = (int)pd_xcc_id() & (XCD_COUNT - 1);
        ids[0] = x;
        ids[1] = (int)(__hip_atomic_fetch_add(a.bar + x * 32 + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % (unsigned)a.W);
        base_s = __hip_atomic_load(a.bar + x * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    XcdCtx c;
    c.x = ids[0];
    c.wx = ids[1];
    c.W = a.W;
    c.N = a.N;
    const int ns = a.B > c.x ? (a.B - c.x + XCD_COUNT - 1) / XCD_COUNT : 0;
    if (ns == 0) return;                       // no sequence lives on this XCD
    c.Mx = ns * a.N;
    c.cnt = a.bar + c.x * 32;
    c.err = a.err;
    c.phase = 0;
    c.base = base_s;
    const size_t part = (size_t)c.x * a.cap_x;
    float *h = a.h + part * DM, *qkv = a.qkv + part * 3 * DM, *ctx = a.ctx + part * DM, *ff = a.ff + part * DFF,
          *hid = a.hid + part * HID;
    const size_t bn9 = (size_t)a.B * a.N * 9;
    for (int step = a.step_begin; step < a.step_end; ++step) {
        const int t = a.T - 1 - step;                                  // reversed(range(T)), gaussian_diffuser.py:296
        const float *xg = a.proc + (size_t)step * bn9;
        float *xn = a.proc + (size_t)(step + 1) * bn9;
        const float *nz = (a.use_noise && t > 0) ? a.noise + (size_t)(step + 1) * bn9 : nullptr;   // :278
        const float *temb = a.t_table + (size_t)t * 128;
        if (!xcd_gemm<KFIRST_PAD, 2, 0>(c, nullptr, a.first_wp, a.first_b, h, DM, xg, a.z, temb, As, red, &flag)) return;
        for (int l = 0; l < a.num_layers; ++l) {
            const XcdLayer &L = a.L[l];
            if (!xcd_gemm<DM, 1, 0>(c, h, L.qkv_wp, L.qkv_b, qkv, 3 * DM, nullptr, nullptr, nullptr, As, red, &flag)) return;
            if (!xcd_attention(c, qkv, ctx, As, &flag)) return;
            if (!xcd_gemm<DM, 0, 2>(c, ctx, L.out_wp, L.out_b, h, DM, nullptr, nullptr, nullptr, As, red, &flag)) return;
            if (!xcd_gemm<DM, 1, 1>(c, h, L.ff1_wp, L.ff1_b, ff, DFF, nullptr, nullptr, nullptr, As, red, &flag)) return;
            if (!xcd_gemm<DFF, 0, 2>(c, ff, L.ff2_wp, L.ff2_b, h, DM, nullptr, nullptr, nullptr, As, red, &flag)) return;
        }
        if (!xcd_gemm<DM, 0, 0>(c, h, a.last0_wp, a.last0_b, hid, HID, nullptr, nullptr, nullptr, As, red, &flag)) return;
        if (!xcd_tail(c, a, hid, xg, nz, xn, a.sched + (size_t)t * 8, &flag)) return;
    }
    // the counter's value at the end is the next launch's base; published by one workgroup after everybody arrived
    if (c.wx == 0) {
        if (!xcd_wait(c, &flag)) return;
        if (threadIdx.x == 0)
            __hip_atomic_store(c.cnt + 1, c.base + (unsigned)c.W * c.phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- host ------------------------------------------------------------------------------------------
int pd_denoiser_xcd_init() {
    PD_HIP_CHECK(hipFuncSetAttribute((const void *)pd_den_xcd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XCD_LDS_BYTES));
    return PD_OK;
}

// can the persistent kernel run this shape with the engine's current option?
bool pd_denoiser_xcd_applicable(const pd_engine *eng, int B, int N) {
    if (eng->den_wgs_per_xcd <= 0 || !eng->den || !eng->den->xcd_bar) return false;
    const int ns = (B + XCD_COUNT - 1) / XCD_COUNT;
    return N <= 64 && ns * N <= eng->den->cap_x && ns * N <= 64;   // larger per-XCD row counts: per-launch kernels win
}

// steps [step_begin, step_end) of the sampler on the engine's own buffers (d_process / d_noise / d_z)
int pd_denoiser_xcd_launch(pd_engine *eng, int B, int N, int step_begin, int step_end, int use_noise, hipStream_t s) {
    PdDenoiserDev *d = eng->den;
    if (!pd_denoiser_xcd_applicable(eng, B, N) || step_begin < 0 || step_end > d->timesteps || step_begin >= step_end) {
        pd_set_error("denoiser (per-XCD): invalid call (B=%d N=%d steps [%d,%d) wgs_per_xcd=%d)", B, N, step_begin, step_end,
                     eng->den_wgs_per_xcd);
        return PD_ERR_INVALID_ARG;
    }
    XcdArgs a;
    memset(&a, 0, sizeof(a));
    a.first_wp = d->first_wp[1];
    a.first_b = d->first_b;
    for (int l = 0; l < d->num_layers; ++l) {
        const PdLayerDev &L = d->layers[l];
        a.L[l] = {L.qkv_wp[1], L.qkv_b, L.out_wp[1], L.out_b, L.ff1_wp[1], L.ff1_b, L.ff2_wp[1], L.ff2_b};
    }
    a.last0_wp = d->last0_wp[1];
    a.last0_b = d->last0_b;
    a.lnw = d->last_ln_w; a.lnb = d->last_ln_b; a.w3 = d->last3_w; a.b3 = d->last3_b;
    a.t_table = d->t_table; a.sched = d->sched;
    a.h = d->h; a.qkv = d->qkv; a.ctx = d->ctx; a.ff = d->ff; a.hid = d->hid;
    a.z = eng->d_z; a.noise = eng->d_noise; a.proc = eng->d_process;
    a.bar = d->xcd_bar; a.err = eng->d_err;
    a.num_layers = d->num_layers; a.cap_x = d->cap_x; a.B = B; a.N = N;
    a.W = eng->den_wgs_per_xcd > 32 ? 32 : eng->den_wgs_per_xcd;
    a.T = d->timesteps; a.step_begin = step_begin; a.step_end = step_end; a.use_noise = use_noise;
    hipLaunchKernelGGL(pd_den_xcd_kernel, dim3(XCD_COUNT * a.W), dim3(256), XCD_LDS_BYTES, s, a);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
