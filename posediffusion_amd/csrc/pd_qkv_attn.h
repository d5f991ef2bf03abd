// pd_qkv_attn.h -- the attention half of an encoder layer as ONE kernel per (group of sequences, head), Q / K / V never leaving the CU
// (round 5; models/denoiser.py:88-97 = nn.TransformerEncoderLayer's self-attention block: in_proj Linear -> scaled-dot-product attention per
// head; BASELINE.json north_star: "attention projections use MFMA tiles with LDS-staged Q/K/V").
//
// Before: pd_gemm_strip_kernel wrote QKV [M, 1536] as fp32 (31.5 MB at 5 120 rows), pd_attn_mma_kernel read it back: 63 MB of the 231 MB a
// layer moved, two launches.  Here a workgroup of 12 wavefronts owns G = 95 / N whole sequences (G N <= 95 token rows in three 32-row MFMA
// tiles; N = 20: 4 sequences, 80 rows) and ONE head:
//   1. the head's 384 in_proj columns (q | k | v, 128 each) as twelve 32-column strips, one per wave, over the three row tiles:
//      the fp16-plane product of pd_gemm_strip_kernel (hi + lo operand pairs, three v_mfma_f32_32x32x16_f16 per 16 k, fp32 accumulation;
//      the A rows -- LayerNorm output as split words -- go L2 -> LDS by LDS-DMA in 64-k chunks, un-zipped at the fragment reads; the weight
//      fragments go L2 -> registers, hand-issued half a chunk ahead; every vector-memory operation counted by hand-written s_waitcnt);
//   2. accumulator * c_scale + bias (q also * 1/sqrt(128)) -> fp32 Q, K, V in LDS (the staging buffers are dead by then: aliased);
//   3. attention per sequence by teams of four waves -- pd_attn_mma_kernel's arithmetic, operation for operation: S = Q K^T on
//      v_mfma_f32_16x16x4_f32, softmax with 8 lanes per row, O = P V -- and ctx written as the split words the out-projection GEMM reads.
// The sums are accumulated in the order of the two kernels this replaces, so the result is BITWISE theirs (tests/test_gpu_parity_r5.py
// compares the two paths); PD_OPT_DENOISER_FUSED_ATTN = 0 keeps the two-launch path (comparison / testing, and N > 32).
#pragma once
#include "pd_gemm_split.h"

#ifndef PD_QA_XCD_MAP
#define PD_QA_XCD_MAP 1
#endif
#define PD_QA_WAVES 12
#define PD_QA_THREADS (PD_QA_WAVES * 64)
#define PD_QA_ROWS 96                       // three 32-row tiles
#define PD_QA_LDR (3 * DH + 4)              // row stride of the Q | K | V image in LDS (floats): 388 = 4 mod 32 banks, like DH + 4
#define PD_QA_LS 36                         // row stride of a team's score tile
#ifndef PD_QA_DEEP_DEFAULT
#define PD_QA_DEEP_DEFAULT 0                // 1: weight fragments a whole chunk ahead (DEEP below) -- measured no faster (925 vs 926 us per step), 35 more registers
#endif

struct PdQkvAttnArgs {
    const unsigned *A;        // LayerNorm output as split words [M][DM] (pd_ln_rows_kernel<DM, 2>)
    const unsigned *W;        // in_proj weights, fp16 planes in fragment order (vit_frag_split_kernel): [1536 / 32][DM / 16][hi | lo][lane] x 16 B
    const float *bias;        // [1536] (LayerNorm shift folded in)
    unsigned *ctx;            // [M][DM] split words of ctx * out_scale
    int B, N, G;              // sequences, frames per sequence, sequences per workgroup (G N < PD_QA_ROWS)
    float c_scale, out_scale;
};
// sequences per workgroup: whole sequences in at most 95 token rows (96 rows + the zero row + the score tiles would exceed the 160 KiB of LDS by 528 B)
static inline int pd_qkv_attn_group(int N) { return N >= 1 && N <= 32 ? (PD_QA_ROWS - 1) / N : 0; }
static inline size_t pd_qkv_attn_lds(int N) {
    const int G = pd_qkv_attn_group(N);
    const size_t image = (size_t)(G * N + 1) * PD_QA_LDR * sizeof(float);          // Q | K | V rows + one zero row
    const size_t stage = (size_t)4 * PD_QA_ROWS * 32 * sizeof(unsigned);           // two 64-k chunks of A
    return (image > stage ? image : stage) + (size_t)3 * 32 * PD_QA_LS * sizeof(float);
}

// BARE (development builds only, -DPD_DEV_KNOBS + PD_QA_BARE=n; results meaningless for n > 0): what bounds the kernel -- 1 no weight-fragment loads
// inside the K loop, 2 no LDS-DMA inside it, 3 neither, 4 everything but the MFMAs, 5 the product only (no Q | K | V image, no attention)
// DEEP: the weight fragments of chunk c + 1 (both 32-k blocks: 8 KiB per wave) are requested at the START of chunk c into a second register set,
// before any MFMA of the chunk is issued (a wave issues in order: loads placed behind a block's 18 MFMAs leave only when the matrix pipe has
// taken them all) -- a whole chunk of matrix work to hide behind instead of half of one; the same MFMA order, bitwise the same result.
template <int BARE, bool DEEP = false>
__global__ __launch_bounds__(PD_QA_THREADS) void pd_qkv_attn_kernel(PdQkvAttnArgs g) {
    constexpr int KC = 32, RT = 3, TM = PD_QA_ROWS, CHA = TM * KC, LDR = PD_QA_LDR, LS = PD_QA_LS;
    extern __shared__ __attribute__((aligned(1024))) unsigned qa_lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (group of sequences, head).  XCD b % 8 runs block b.  PD_QA_XCD_MAP 0 (rounds 5 - 6): the four heads of a group are neighbours -- XCD x hosts head x % 4
    // (786 KB of weights) and touches the rows of HALF of all groups: every group's rows cross the fabric four times (55 MB per launch at 5 120 rows against 21 if
    // every operand crossed once; profiles/round6_pmc_summary.json).  1: in chunks of 16 blocks = 4 groups, XCDs 0 - 3 host heads {0, 1}, XCDs 4 - 7 heads {2, 3}, XCD x
    // the groups = x mod 4: two heads' weights (1.6 MB) and a quarter of the groups' rows per L2, every row crosses twice.  Same workgroups, same arithmetic.
    int head = blockIdx.x & 3, grp = blockIdx.x >> 2;
#if PD_QA_XCD_MAP == 1
    {
        const int ngrp = gridDim.x >> 2, c = blockIdx.x >> 4, r = blockIdx.x & 15;
        if (4 * c + 4 <= ngrp) {
            head = 2 * ((r & 7) >> 2) + (r >> 3);
            grp = 4 * c + (r & 3);
        }
    }
#elif PD_QA_XCD_MAP == 2      // (measured alternative: all four heads of a group on ONE XCD -- every row crosses once, 3.1 MB of weights per L2)
    {
        const int ngrp = gridDim.x >> 2, c = blockIdx.x >> 5, r = blockIdx.x & 31;
        if (8 * c + 8 <= ngrp) {
            head = r >> 3;
            grp = 8 * c + (r & 7);
        }
    }
#endif
    const int N = g.N, G = g.G, M = g.B * N;
    const int seq0 = grp * G, nseq = min(G, g.B - seq0), m0 = seq0 * N, rows = nseq * N;     // this workgroup's token rows [m0, m0 + rows)
    // ---- 1. the in_proj product ------------------------------------------------------------------------------------------------------
    // staging: pieces of 1 KiB = 8 rows x 32 words; a 32-k block of A has 12 of them, wave w moves piece w
    const int prow = lane >> 3, pslot = lane & 7;
    unsigned oa;
    {
        const int r = 8 * wave + prow;
        oa = (unsigned)(((size_t)min(m0 + r, M - 1) * DM + 4 * (pslot ^ ((r >> 1) & 7))) * sizeof(unsigned));
    }
    const unsigned lds_a = (unsigned)(size_t)(qa_lds + wave * 256);
    const int third = wave >> 2;                                                     // 0 q, 1 k, 2 v
    const int ntile = third * (DM / 32) + head * (DH / 32) + (wave & 3);              // this wave's 32-column tile of the 1536 in_proj rows
    constexpr int KS = DM / 16;
    const uint4 *wq = (const uint4 *)g.W + (size_t)ntile * KS * 128 + lane;
    typedef unsigned wv4 __attribute__((ext_vector_type(4)));
    f32x16 acc[RT];
#pragma unroll
    for (int mi = 0; mi < RT; ++mi)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mi][i] = 0.0f;
    auto mmaw = [](const uint4 &a, const wv4 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    };
#define PD_QA_WLOAD(w0, w1, w2, w3, b)                                                                                               \
    do {                                                                                                                             \
        const uint4 *wp_ = wq + (size_t)(b) * 256;                                                                                   \
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"                           \
                     "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"                    \
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(wp_) : "memory");                                            \
    } while (0)
#define PD_QA_WAIT(n, w0, w1, w2, w3) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : : "memory")
    auto stage64 = [&](int c, int buf) {            // both 32-k blocks of chunk c: two pieces per wave
        const unsigned da = __builtin_amdgcn_readfirstlane(lds_a + buf * 2 * CHA * 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) pd_dma_piece((const float *)(g.A + (2 * c + h) * KC), oa, da + h * CHA * 4);
    };
    // one 32-k block: fragment reads + un-zip + 18 MFMAs; w0 / w1 = k step 0 {hi | lo}, w2 / w3 = k step 1.  Per output element and k step:
    // lo x hi, hi x lo, hi x hi -- pd_gemm_strip_kernel's order
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
            for (int mi = 0; mi < RT; ++mi) acc[mi] = mmaw(al[mi], wh, acc[mi]);
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) acc[mi] = mmaw(ah[mi], wl, acc[mi]);
#pragma unroll
            for (int mi = 0; mi < RT; ++mi) acc[mi] = mmaw(ah[mi], wh, acc[mi]);
        }
    };
    {
        wv4 a0, a1, a2, a3, b0, b1, b2, b3;               // weight fragments of the chunk's first / second 32-k block
        constexpr int nk64 = DM / 64;
        stage64(0, 0);
        PD_QA_WLOAD(a0, a1, a2, a3, 0);
        PD_QA_WLOAD(b0, b1, b2, b3, 1);
        PD_QA_WAIT(0, a0, a1, a2, a3);
        PD_QA_WAIT(0, b0, b1, b2, b3);
        __syncthreads();
        if constexpr (DEEP) {
            static_assert(!DEEP || BARE == 0, "the deep form has no development variants");
            wv4 c0, c1, c2, c3, d0, d1, d2, d3;              // the second set: chunk c + 1 while a / b hold chunk c, and vice versa
            for (int c = 0; c < nk64; c += 2) {              // nk64 is even
                {
                    const unsigned *a = qa_lds + 0 * 2 * CHA + l31 * KC;
                    stage64(c + 1, 1);
                    PD_QA_WLOAD(c0, c1, c2, c3, 2 * (c + 1));
                    PD_QA_WLOAD(d0, d1, d2, d3, 2 * (c + 1) + 1);
                    block(a, a0, a1, a2, a3);
                    block(a + CHA, b0, b1, b2, b3);
                    PD_QA_WAIT(0, c0, c1, c2, c3);           // everything requested at the start of this chunk has landed
                    PD_QA_WAIT(0, d0, d1, d2, d3);
                    __syncthreads();
                }
                {
                    const int cn = min(c + 2, nk64 - 1);     // the chunk after the last is the last again (never used)
                    const unsigned *a = qa_lds + 1 * 2 * CHA + l31 * KC;
                    stage64(cn, 0);
                    PD_QA_WLOAD(a0, a1, a2, a3, 2 * cn);
                    PD_QA_WLOAD(b0, b1, b2, b3, 2 * cn + 1);
                    block(a, c0, c1, c2, c3);
                    block(a + CHA, d0, d1, d2, d3);
                    PD_QA_WAIT(0, a0, a1, a2, a3);
                    PD_QA_WAIT(0, b0, b1, b2, b3);
                    __syncthreads();
                }
            }
        } else
        for (int c = 0; c < nk64; ++c) {
            const int cn = min(c + 1, nk64 - 1);                 // the chunk after the last is the last again (never used)
            const unsigned *a = qa_lds + (c & 1) * 2 * CHA + l31 * KC;
            if constexpr (BARE == 0) {
                if (c + 1 < nk64) {
                    stage64(cn, (c + 1) & 1);                    // in flight, oldest first: the second block's weights [4, from the previous turn], this DMA [2]
                    block(a, a0, a1, a2, a3);
                    PD_QA_WLOAD(a0, a1, a2, a3, 2 * cn);         //   ... + the next chunk's first block [4]
                    PD_QA_WAIT(6, b0, b1, b2, b3);               // the second block's weights have landed; the DMA [2] and the loads just issued [4] may fly
                    block(a + CHA, b0, b1, b2, b3);
                    PD_QA_WLOAD(b0, b1, b2, b3, 2 * cn + 1);     //   ... + the next chunk's second block [4]
                    PD_QA_WAIT(4, a0, a1, a2, a3);               // the next chunk's rows and first block have landed; the second block may still fly
                } else {                                         // the last chunk requests nothing (round 6; it used to re-stage itself: 1 / 8 of the loop's L2 traffic
                    block(a, a0, a1, a2, a3);                    //   and a whole load latency at the loop's end, for operands never used)
                    PD_QA_WAIT(0, b0, b1, b2, b3);
                    block(a + CHA, b0, b1, b2, b3);
                }
            } else {                                             // development variants (see BARE): every wait drains
                if constexpr (BARE != 2 && BARE != 3) stage64(cn, (c + 1) & 1);
                if constexpr (BARE != 4) block(a, a0, a1, a2, a3);
                if constexpr (BARE != 1 && BARE != 3) PD_QA_WLOAD(a0, a1, a2, a3, 2 * cn);
                if constexpr (BARE != 4) block(a + CHA, b0, b1, b2, b3);
                if constexpr (BARE != 1 && BARE != 3) PD_QA_WLOAD(b0, b1, b2, b3, 2 * cn + 1);
                PD_QA_WAIT(0, a0, a1, a2, a3);
                PD_QA_WAIT(0, b0, b1, b2, b3);
            }
            __syncthreads();
        }
        PD_QA_WAIT(0, b0, b1, b2, b3);
    }
#undef PD_QA_WLOAD
#undef PD_QA_WAIT
    if constexpr (BARE == 5) {
        if (acc[0][0] == 123.456f) g.ctx[tid] = 1u;                // keep the product alive
        return;
    }
    // ---- 2. Q | K | V of the workgroup's rows as fp32 in LDS (every wave is past its last fragment read: the barrier that ended the loop) ----
    float *img = (float *)qa_lds;                                // [G N + 1][LDR]: columns [0, 128) q / sqrt(dh), [128, 256) k, [256, 384) v; row G N = 0
    float *Sall = img + (size_t)(G * N + 1) * LDR;               // [3 teams][32][LS]
    {
        const int col = ntile * 32 + l31;                        // in_proj row = column of the QKV matrix
        const float bias = g.bias[col];
        const int lcol = third * DH + (wave & 3) * 32 + l31;
        const float qscale = 0.08838834764831845f;               // 1/sqrt(128): pd_attn_mma_kernel scales q as it stages it
#pragma unroll
        for (int mi = 0; mi < RT; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = mi * 32 + 4 * hi + (i & 3) + 8 * (i >> 2);
                float v = fmaf(acc[mi][i], g.c_scale, bias);     // = what pd_gemm_strip_kernel stored as QKV
                if (third == 0) v *= qscale;
                if (row < rows) img[row * LDR + lcol] = v;
            }
        for (int c = tid; c < LDR; c += PD_QA_THREADS) img[G * N * LDR + c] = 0.0f;       // the row every index beyond N reads
    }
    __syncthreads();
    // ---- 3. attention: team t (four waves) takes sequences t, t + 3, ... of the workgroup; pd_attn_mma_kernel's arithmetic ---------------
    const int team = wave >> 2, tw = wave & 3, ttid = tid & 255;
    float *S = Sall + team * 32 * LS;
    const int rounds = (G + 2) / 3;
    for (int rd = 0; rd < rounds; ++rd) {
        const int s = rd * 3 + team;
        const bool active = s < nseq;
        const float *base = img + (size_t)s * N * LDR;           // local row j of the sequence: base + j LDR (j < N), the zero row otherwise
        const float *zrow = img + (size_t)G * N * LDR;
        auto rowp = [&](int j) { return j < N ? base + j * LDR : zrow; };
        if (active) {   // scores: wave tw owns the tile rows 16 (tw >> 1) .., keys 16 (tw & 1) ..; lane = (row or key) % 16 + 16 g feeds k = 16 c + 4 g + e
            const float *qa = rowp(16 * (tw >> 1) + (lane & 15)) + 4 * (lane >> 4);
            const float *kb = rowp(16 * (tw & 1) + (lane & 15)) + DH + 4 * (lane >> 4);
            f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DH / 16; ++c) {
                const float4 a = *(const float4 *)(qa + 16 * c), k = *(const float4 *)(kb + 16 * c);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, k.x, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, k.y, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, k.z, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, k.w, sc, 0, 0, 0);
            }
            const int j = 16 * (tw & 1) + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) S[(16 * (tw >> 1) + 4 * (lane >> 4) + e) * LS + j] = sc[e];
        }
        __syncthreads();
        if (active) {   // softmax: 8 lanes per row, 4 keys per lane
            const int i = ttid >> 3, sub = ttid & 7;
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
        if (active) {   // O = P V: wave tw owns the output columns [32 tw, 32 tw + 32) (two tiles) of both row tiles; k = key j = 16 c + 4 g + e
            f32x4 o[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) o[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *pa = S + (lane & 15) * LS + 4 * (lane >> 4);
            const int vcol = 2 * DH + 32 * tw + (lane & 15);
            const int jg = 4 * (lane >> 4);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4 p0 = *(const float4 *)(pa + 16 * c), p1 = *(const float4 *)(pa + 16 * LS + 16 * c);
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float *vr = rowp(16 * c + jg + e) + vcol;
                    v0[e] = vr[0];
                    v1[e] = vr[16];
                }
                const float a0[4] = {p0.x, p0.y, p0.z, p0.w}, a1[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], v0[e], o[0][0], 0, 0, 0);
                    o[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], v1[e], o[0][1], 0, 0, 0);
                    o[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], v0[e], o[1][0], 0, 0, 0);
                    o[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], v1[e], o[1][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * rt + 4 * (lane >> 4) + e;
                    if (i < N) {
                        unsigned *out = g.ctx + (size_t)((seq0 + s) * N + i) * DM + head * DH + 32 * tw + (lane & 15);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) out[16 * ct] = pd_split_word_as<2>(o[rt][ct][e], g.out_scale);
                    }
                }
        }
        __syncthreads();                                         // the team's score tile is rewritten by the next round
    }
}

// LayerNorm output (split words) -> ctx (split words) for B sequences of N <= 32 frames
static inline void pd_qkv_attn(const unsigned *hn, const unsigned *Wh, const float *bias, unsigned *ctx, int B, int N, float c_scale, float out_scale,
                               hipStream_t s) {
    const int G = pd_qkv_attn_group(N);
    PdQkvAttnArgs g{hn, Wh, bias, ctx, B, N, G, c_scale, out_scale};
#ifdef PD_DEV_KNOBS
    static const int bare = pd_dev_knob("PD_QA_BARE", 0);
    const dim3 grid(((B + G - 1) / G) * NH), blk(PD_QA_THREADS);
    const size_t lds = pd_qkv_attn_lds(N);
    static const int deep = pd_dev_knob("PD_QA_DEEP", PD_QA_DEEP_DEFAULT);
    if (bare == 0 && !deep) { hipLaunchKernelGGL((pd_qkv_attn_kernel<0, false>), grid, blk, lds, s, g); return; }
    switch (bare) {
    case 1: hipLaunchKernelGGL(pd_qkv_attn_kernel<1>, grid, blk, lds, s, g); return;
    case 2: hipLaunchKernelGGL(pd_qkv_attn_kernel<2>, grid, blk, lds, s, g); return;
    case 3: hipLaunchKernelGGL(pd_qkv_attn_kernel<3>, grid, blk, lds, s, g); return;
    case 4: hipLaunchKernelGGL(pd_qkv_attn_kernel<4>, grid, blk, lds, s, g); return;
    case 5: hipLaunchKernelGGL(pd_qkv_attn_kernel<5>, grid, blk, lds, s, g); return;
    default: break;
    }
#endif
    hipLaunchKernelGGL((pd_qkv_attn_kernel<0, PD_QA_DEEP_DEFAULT != 0>), dim3(((B + G - 1) / G) * NH), dim3(PD_QA_THREADS), pd_qkv_attn_lds(N), s, g);
}
