// pd_ggs.hip -- Geometry-Guided Sampling as ONE persistent kernel per guided diffusion step.
//
// Replaces (paths relative to /root/reference/pose_diffusion/):
//   util/geometry_guided_sampling.py:14-64   geometry_guided_sampling  (5 optimisations)
//   util/geometry_guided_sampling.py:67-126  GGS_optimize  (clipped momentum SGD, early exit)
//   util/geometry_guided_sampling.py:129-172 compute_sampson_distance
//   util/get_fundamental_matrix.py:14-51     F for the frame pairs that own matches
//   util/camera_transform.py:80-97           pose decode (quat -> R, clamp(exp(logFL + 1.8)))
// plus torch autograd's backward of all of the above, derived by hand (DESIGN.md "GGS backward";
// the same derivation in fp64 numpy is oracle/pd_oracle.py:sampson_loss_grad_analytic).
//
// Mapping to CDNA4.  The reference launches ~1850 ATen kernels per iteration and 700 iterations
// per guided step; the chain is strictly sequential, so the design goal is launch-free, sync-cheap
// iterations:
//   * one launch runs every iteration of every stage; pose parameters, momentum and all per-frame
//     state live in registers/LDS of the owning workgroup(s);
//   * matches are pair-sorted at upload; a wavefront owns one (pair, <=512 matches) work item, the
//     pair's F is wave-uniform, the per-match Sampson residual + dL/dF is accumulated per lane and
//     reduced with a 64-lane butterfly (fixed order -> bitwise reproducible);
//   * k workgroups may cooperate on one sequence (k = ceil(items / 8) when CUs are free): each
//     publishes its 12 per-item sums as tagged 8-byte granules (write-through, data-is-the-flag,
//     guide section 6 G16 R2) and every workgroup gathers all of them, then redundantly runs the
//     tiny per-frame backward + SGD update, so there is exactly ONE cross-workgroup hop per
//     iteration and no broadcast of the new parameters.  All arithmetic orders are fixed, so the
//     replicas stay bitwise identical (and k = 1 and k > 1 give identical bits).
//   * blockIdx -> (sequence, workgroup) is XCD-aware: with B % 8 == 0 all workgroups of a sequence
//     sit on one XCD (dispatcher places block b on XCD b % 8) so the exchange stays in one L2.
//     That is a speed choice only; correctness uses agent-scope granules and bounded spins.
#include "pd_internal.h"

#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>

// Floating-point contraction is decided per source expression in this file (not by the backend across statements, as
// -ffp-contract=fast allows): the GGS kernel exists in several template variants (resident / LDS-staged / register-streamed
// match pass, two-hop) whose results must agree BIT FOR BIT for the same sequence whatever the launch shape -- a backend
// that fuses a*b+c differently in two instantiations of the same source line would break that.
#pragma clang fp contract(on)

#ifndef PD_GGS_PROF12
#define PD_GGS_PROF12 0
#endif
#ifndef PD_GGS_ABLATE
#define PD_GGS_ABLATE 0   // development (tools/ab_ggs.py): bit mask of phases whose work is SKIPPED (1 P1, 2 P3a, 4 P3b, 8 P4, 16 match pass, 32 exchange
#endif                    // gather) -- garbage results, honest timing of what is left: the difference is a phase's share of the critical path

#ifndef PD_GGS_MIN_WAVES_PER_SIMD
#define PD_GGS_MIN_WAVES_PER_SIMD 2      // one 512-thread workgroup per CU; 4 = experiment: two workgroups per CU (<= 128 VGPRs)
#endif
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define PD_XCHG_LINE 16   // granules per item record in the exchange buffer (one 128-byte line)
#define PD_F_STRIDE 12    // floats per slot of the per-item F in LDS (9 used; 16-byte aligned rows)

// --------------------------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------------------------
// 64-lane sum on the DPP cross-lane network (no LDS round trips): xor-1, xor-2 quad permutes,
// half-row and row mirrors give every lane its 16-lane row sum; row_bcast15/31 chain the four rows;
// lane 63 holds the total, read back into an SGPR.  Fixed tree -> bitwise reproducible, and the
// result is wave-uniform by construction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_allsum(float v) {
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);   // row_mirror
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// 1-ulp hardware reciprocal / sqrt for scale factors on the serial per-iteration chain and for the gradient scales
// of the match pass.  The hard `sampson < sampson_max` test (geometry_guided_sampling.py:170) is decided on the IEEE
// quotient top / bottom like torch's: see sampson_step2 (fast pass + exact re-run of an item that has a match inside
// the band where the 1-ulp quotient could decide differently).
__device__ __forceinline__ float pd_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float pd_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Transposing butterfly over the per-item sums (slots 0..9 carry values; 10..15 are padding): at every step a lane KEEPS half
// of its values and SENDS the other half to the partner that differs in exactly ONE lane bit (who keeps exactly those), so the
// live values go 16 -> 8 -> 4 -> 2 -> 1 per lane.  Lane bits 2 and 3 go first: they select a DPP bank (4 lanes), so a row shift
// with a bank mask adds the partner's value AND picks which of the two values a lane keeps in one v_add_f32_dpp -- no selects
// (hand-written: hipcc only emits the masked form as v_mov_b32_dpp pairs + selects).  Then bits 0 / 1 as quad permutes with
// selects, bits 4 / 5 as permlane swaps.  ~45 instructions per item instead of 10 full 64-lane reductions.  Value `slot`
// ends up in every lane whose low four bits encode that slot.  Fixed tree -> bitwise reproducible; every lane must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// Partner exchanges of the butterfly, all on the VALU cross-lane paths (no LDS crossbar round trips: ds_swizzle / ds_bpermute
// cost ~100+ cycles each on a chain that runs once per work item):
//   lane ^ 4, lane ^ 8   two DPP row shifts each (up for the lanes whose bit is clear, down for the others, picked by bank_mask)
//   lane ^ 16, lane ^ 32 gfx950's v_permlane16_swap / v_permlane32_swap: swapping the odd rows (upper half) of one copy with
//                        the even rows (lower half) of another leaves {x[lane & ~b], x[lane | b]} in the two copies
__device__ __forceinline__ float add_xor16(float v) {    // v[lane] + v[lane ^ 16]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float add_xor32(float v) {    // v[lane] + v[lane ^ 32]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float wave_reduce12_transpose(const float (&a)[PD_ITEM_VALS], int lane, int &slot) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float w0, w1, w2, w3, w4, w5, w6, w7;
    // lane ^ 4: banks 0, 2 (bit 2 clear) keep slot j = a[j] + a[j] of lane + 4; banks 1, 3 keep slot j + 8 (only 8 and 9 exist; the
    // other lanes of w2..w7 stay undefined -- they would hold the padding slots, which nobody reads).  The leading s_nop covers
    // the VALU-write -> DPP-read hazard the assembler cannot see for us.
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %1, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %2, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %3, %11, %11 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %4, %12, %12 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %5, %13, %13 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %6, %14, %14 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %7, %15, %15 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %16, %16 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_add_f32_dpp %1, %17, %17 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "=&v"(w7)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]));
    // lane ^ 8: banks 0, 1 (bit 3 clear) keep w[j], banks 2, 3 keep w[j + 4]
    float q0, q1, q2, q3;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %2, %6, %6 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %3, %7, %7 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %1, %9, %9 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %2, %10, %10 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %3, %11, %11 row_shr:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                 : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));
    const float p0 = (b0 ? q2 : q0) + dpp_mov<0xB1>(b0 ? q0 : q2);     // lane ^ 1
    const float p1 = (b0 ? q3 : q1) + dpp_mov<0xB1>(b0 ? q1 : q3);
    float v = (b1 ? p1 : p0) + dpp_mov<0x4E>(b1 ? p0 : p1);            // lane ^ 2
    v = add_xor16(v);
    v = add_xor32(v);
    slot = ((lane & 4) ? 8 : 0) + ((lane & 8) ? 4 : 0) + (b0 ? 2 : 0) + (b1 ? 1 : 0);
    return v;
}

// two 64-lane sums for little more than the price of one: v_permlane32_swap folds a's upper half onto its lower half and b's lower
// half onto its upper half (one swap + one add), then ONE five-step DPP chain sums both 32-lane halves; a in lane 31, b in lane 63.
__device__ __forceinline__ void wave_allsum2(float a, float b, float &sa, float &sb) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
    float v = __int_as_float(r[0]) + __int_as_float(r[1]);   // lanes < 32: a[l] + a[l + 32]; lanes >= 32: b[l - 32] + b[l]
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);   // row_mirror
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
    sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

struct Cam {   // shared intrinsics of the step: A = K^-1 = [[a0,0,c0],[0,a1,c1],[0,0,1]]
    float a0, a1, c0, c1;
};

// forward of get_essential_matrix for one ordered pair (camera 1 = i, camera 2 = j)
// (get_fundamental_matrix.py:45-51), keeping the intermediates the backward needs.
struct PairFwd {
    float R12[9], t12[3], Et[3], E[9];
};

__device__ __forceinline__ void pair_forward(const float *Ri, const float *ti, const float *Rj, const float *tj,
                                             PairFwd &o) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            o.R12[a * 3 + c] = Rj[a * 3 + 0] * Ri[c * 3 + 0] + Rj[a * 3 + 1] * Ri[c * 3 + 1] + Rj[a * 3 + 2] * Ri[c * 3 + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        o.t12[a] = tj[a] - (o.R12[a * 3 + 0] * ti[0] + o.R12[a * 3 + 1] * ti[1] + o.R12[a * 3 + 2] * ti[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a)
        o.Et[a] = -(o.R12[0 * 3 + a] * o.t12[0] + o.R12[1 * 3 + a] * o.t12[1] + o.R12[2 * 3 + a] * o.t12[2]);
    const float ex = o.Et[0], ey = o.Et[1], ez = o.Et[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // E = R12 * hat(Et), hat = [[0,-z,y],[z,0,-x],[-y,x,0]]
        o.E[a * 3 + 0] = o.R12[a * 3 + 1] * ez - o.R12[a * 3 + 2] * ey;
        o.E[a * 3 + 1] = o.R12[a * 3 + 2] * ex - o.R12[a * 3 + 0] * ez;
        o.E[a * 3 + 2] = o.R12[a * 3 + 0] * ey - o.R12[a * 3 + 1] * ex;
    }
}

// F as used by _sampson_distance after the permute of geometry_guided_sampling.py:155:
// F = (K2^-T E K1^-1)^T = (A^T E A)^T   (get_fundamental_matrix.py:41; K1 = K2, focal is the mean)
__device__ __forceinline__ void fundamental_from_E(const float *E, const Cam &c, float *F) {
    float Mx[9];   // A^T E
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        Mx[0 * 3 + q] = c.a0 * E[0 * 3 + q];
        Mx[1 * 3 + q] = c.a1 * E[1 * 3 + q];
        Mx[2 * 3 + q] = c.c0 * E[0 * 3 + q] + c.c1 * E[1 * 3 + q] + E[2 * 3 + q];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {   // Fo = Mx A ; F[c][r] = Fo[r][c]
        F[0 * 3 + r] = Mx[r * 3 + 0] * c.a0;
        F[1 * 3 + r] = Mx[r * 3 + 1] * c.a1;
        F[2 * 3 + r] = Mx[r * 3 + 0] * c.c0 + Mx[r * 3 + 1] * c.c1 + Mx[r * 3 + 2];
    }
}

// Two matches per lane at once on packed-fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): every per-match
// quantity is a float2 (x = match A, y = match B).  P2 is VALU-issue bound (2 waves per SIMD x ~85 instructions per
// match when the compiler packs within one match), so packing ACROSS matches nearly halves its instruction count.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pd_fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pd_splat(float a) { return (v2f){a, a}; }
// F[k] as a two-match operand.  From nine scalars (the wave-per-item kernels: F is wave-uniform) or from five register PAIRS
// {F0,F1} {F2,F3} {F4,F5} {F6,F7} {F8,-} (the lane-per-item kernel: F is per lane; a half of a 64-bit pair feeds both halves of a
// packed instruction through op_sel, so the nine values cost 10 registers instead of 18 splatted ones)
struct PdFPairs {
    v2f p[5];
};
template <int K>
__device__ __forceinline__ v2f pd_fsplat(const float *F) { return pd_splat(F[K]); }
template <int K>
__device__ __forceinline__ v2f pd_fsplat(const PdFPairs &F) {
    return (K & 1) ? __builtin_shufflevector(F.p[K / 2], F.p[K / 2], 1, 1) : __builtin_shufflevector(F.p[K / 2], F.p[K / 2], 0, 0);
}

// Sampson residual + dL/dF of two matches (geometry_guided_sampling.py:157-170); acc[0..8] dL/dF sums,
// acc[9] sum(s valid), each as {match A, match B} partial sums.  Kept out of the per-match work (item_totals() finishes them per item):
//   * acc[0..8] accumulate HALF of dL/dF (ca, cb below without their factor 2 -- an exact scaling, doubled after the reduction);
//   * n_valid (slot 10) is counted on the scalar unit: popcounts of the two compare masks, wave-uniform and exact;
//   * sum(min(s, max)) (slot 11, the printed statistic :169) = sum(s valid) + max * (in-range matches - n_valid).
//
// Threshold rule (:170 `sampson < sampson_max` on torch's IEEE quotient top / bottom): EXACT = false computes the
// quotient as top * v_rcp_f32(bottom) (within 2 ulp of the IEEE quotient) and records in `mind` how close any in-range
// match came to the threshold; the caller re-runs the whole item with EXACT = true (IEEE divide for the quotient that
// is compared, clamped and summed) when some match of the wave lies within PD_SAMPSON_BAND_ULPS of sampson_max --
// outside that band both quotients decide alike, so the valid set is always the one the IEEE quotient gives.  The
// gradient scales 1/bottom keep the 1-ulp reciprocal in both variants (no threshold hangs on them).
#define PD_SAMPSON_BAND_ULPS 16.0f
// Every fused multiply-add below is written out and contraction is off inside the two step functions, so the packed and the
// single-match form perform the same roundings: an item's sums do not depend on which form ran its tail.
template <bool EXACT, typename FT>
__device__ __forceinline__ void sampson_step2(const v2f u1, const v2f v1, const v2f u2, const v2f v2, bool ina, bool inb, const FT &F, float smax,
                                              v2f (&acc)[PD_ITEM_VALS], float &mind, int &nv, unsigned long long lanes = ~0ull) {
#pragma clang fp contract(off)
    // left = x1^T F, right = F x2   (:158-159)
    const v2f l0 = pd_fma2(u1, pd_fsplat<0>(F), pd_fma2(v1, pd_fsplat<3>(F), pd_fsplat<6>(F)));
    const v2f l1 = pd_fma2(u1, pd_fsplat<1>(F), pd_fma2(v1, pd_fsplat<4>(F), pd_fsplat<7>(F)));
    const v2f l2 = pd_fma2(u1, pd_fsplat<2>(F), pd_fma2(v1, pd_fsplat<5>(F), pd_fsplat<8>(F)));
    const v2f r0 = pd_fma2(pd_fsplat<0>(F), u2, pd_fma2(pd_fsplat<1>(F), v2, pd_fsplat<2>(F)));
    const v2f r1 = pd_fma2(pd_fsplat<3>(F), u2, pd_fma2(pd_fsplat<4>(F), v2, pd_fsplat<5>(F)));
    const v2f ee = pd_fma2(l0, u2, pd_fma2(l1, v2, l2));
    const v2f bottom = pd_fma2(r1, r1, pd_fma2(r0, r0, pd_fma2(l1, l1, l0 * l0)));   // :161
    const v2f inv = {pd_rcp(bottom.x), pd_rcp(bottom.y)};
    const v2f top = ee * ee;
    v2f sam;                                                            // :162-164
    if (EXACT) {
        sam = (v2f){top.x / bottom.x, top.y / bottom.y};                // IEEE, as torch divides
    } else {
        sam = top * inv;
        const v2f d = sam - pd_splat(smax);
        // lanes past the item's end carry a clamped copy of its last match: harmless (same decision as that match)
        mind = fminf(mind, fminf(fabsf(d.x), fabsf(d.y)));              // one v_min3_f32 with |.| modifiers
    }
    const bool va = ina && (sam.x < smax), vb = inb && (sam.y < smax);   // :170 (false for NaN)
    // everything below is scaled by inv_v = valid ? 1/bottom : 0 (a select, not a product: 1/bottom may be inf),
    // so invalid / out-of-range matches contribute exact zeros without further masking
    const v2f inv_v = {va ? inv.x : 0.0f, vb ? inv.y : 0.0f};
    const v2f ca = ee * inv_v;                        // ee / bottom      (half of d sam / d ee)
    const v2f sam_v = EXACT ? (v2f){va ? sam.x : 0.0f, vb ? sam.y : 0.0f} : top * inv_v;   // = sam where valid, else 0
    const v2f cb = sam_v * inv_v;                     // sam / bottom     (half of -d sam / d bottom)
    acc[9] += sam_v;
    // (`lanes`: the lanes that count -- all of them in the wave-per-item kernels; the lane-per-item kernel's last wave has lanes without an item)
    nv += __builtin_popcountll(__builtin_amdgcn_ballot_w64(va) & lanes) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(vb) & lanes);
    // (d sam / dF[r][c]) / 2 = x1[r] g_c - cb r_r x2[c] [r<2],  g_c = ca x2[c] - cb l_c [c<2]   (x1[2] = x2[2] = 1)
    const v2f g0 = pd_fma2(ca, u2, -(cb * l0)), g1 = pd_fma2(ca, v2, -(cb * l1));
    const v2f nbr0 = -(cb * r0), nbr1 = -(cb * r1);
    acc[0] = pd_fma2(nbr0, u2, pd_fma2(u1, g0, acc[0]));
    acc[1] = pd_fma2(nbr0, v2, pd_fma2(u1, g1, acc[1]));
    acc[2] = pd_fma2(u1, ca, acc[2]) + nbr0;
    acc[3] = pd_fma2(nbr1, u2, pd_fma2(v1, g0, acc[3]));
    acc[4] = pd_fma2(nbr1, v2, pd_fma2(v1, g1, acc[4]));
    acc[5] = pd_fma2(v1, ca, acc[5]) + nbr1;
    acc[6] += g0;
    acc[7] += g1;
    acc[8] += ca;
}

// W two-match steps at once, operation by operation (the lane-per-item kernel: one or two waves per SIMD cannot hide the VALU dependency
// latency of ONE step's serial chain l -> bottom -> 1/bottom -> sam -> valid -> ca, cb -> g -> sums; W independent chains issued in
// lockstep can).  Same operations as sampson_step2 on every match, and the sums take step 0's contribution first, then step 1's, ...:
// exactly what W successive sampson_step2 calls compute.
template <bool EXACT, int W, typename FT>
__device__ __forceinline__ void sampson_stepW(const v2f (&u1)[W], const v2f (&v1)[W], const v2f (&u2)[W], const v2f (&v2)[W], const bool (&ina)[W],
                                              const bool (&inb)[W], const FT &F, float smax, v2f (&acc)[PD_ITEM_VALS], float &mind, int &nv,
                                              unsigned long long lanes) {
#pragma clang fp contract(off)
    v2f l0[W], l1[W], l2[W], r0[W], r1[W], ee[W], bottom[W], inv[W], top[W], sam[W], inv_v[W], ca[W], sam_v[W], cb[W], g0[W], g1[W], nbr0[W], nbr1[W];
    bool va[W], vb[W];
#define PD_W for (int w = 0; w < W; ++w)
#pragma unroll
    PD_W l0[w] = pd_fma2(v1[w], pd_fsplat<3>(F), pd_fsplat<6>(F));
#pragma unroll
    PD_W l1[w] = pd_fma2(v1[w], pd_fsplat<4>(F), pd_fsplat<7>(F));
#pragma unroll
    PD_W l2[w] = pd_fma2(v1[w], pd_fsplat<5>(F), pd_fsplat<8>(F));
#pragma unroll
    PD_W r0[w] = pd_fma2(pd_fsplat<1>(F), v2[w], pd_fsplat<2>(F));
#pragma unroll
    PD_W r1[w] = pd_fma2(pd_fsplat<4>(F), v2[w], pd_fsplat<5>(F));
#pragma unroll
    PD_W l0[w] = pd_fma2(u1[w], pd_fsplat<0>(F), l0[w]);
#pragma unroll
    PD_W l1[w] = pd_fma2(u1[w], pd_fsplat<1>(F), l1[w]);
#pragma unroll
    PD_W l2[w] = pd_fma2(u1[w], pd_fsplat<2>(F), l2[w]);
#pragma unroll
    PD_W r0[w] = pd_fma2(pd_fsplat<0>(F), u2[w], r0[w]);
#pragma unroll
    PD_W r1[w] = pd_fma2(pd_fsplat<3>(F), u2[w], r1[w]);
#pragma unroll
    PD_W ee[w] = pd_fma2(l1[w], v2[w], l2[w]);
#pragma unroll
    PD_W bottom[w] = l0[w] * l0[w];
#pragma unroll
    PD_W ee[w] = pd_fma2(l0[w], u2[w], ee[w]);
#pragma unroll
    PD_W bottom[w] = pd_fma2(l1[w], l1[w], bottom[w]);
#pragma unroll
    PD_W bottom[w] = pd_fma2(r0[w], r0[w], bottom[w]);
#pragma unroll
    PD_W bottom[w] = pd_fma2(r1[w], r1[w], bottom[w]);                      // :161
#pragma unroll
    PD_W top[w] = ee[w] * ee[w];
#pragma unroll
    PD_W inv[w] = (v2f){pd_rcp(bottom[w].x), pd_rcp(bottom[w].y)};
    if (EXACT) {
#pragma unroll
        PD_W sam[w] = (v2f){top[w].x / bottom[w].x, top[w].y / bottom[w].y};   // IEEE, as torch divides   (:162-164)
    } else {
#pragma unroll
        PD_W sam[w] = top[w] * inv[w];
#pragma unroll
        PD_W {
            const v2f d = sam[w] - pd_splat(smax);
            mind = fminf(mind, fminf(fabsf(d.x), fabsf(d.y)));               // one v_min3_f32 with |.| modifiers
        }
    }
#pragma unroll
    PD_W {
        va[w] = ina[w] && (sam[w].x < smax);                                 // :170 (false for NaN)
        vb[w] = inb[w] && (sam[w].y < smax);
    }
#pragma unroll
    PD_W inv_v[w] = (v2f){va[w] ? inv[w].x : 0.0f, vb[w] ? inv[w].y : 0.0f};
#pragma unroll
    PD_W ca[w] = ee[w] * inv_v[w];
#pragma unroll
    PD_W sam_v[w] = EXACT ? (v2f){va[w] ? sam[w].x : 0.0f, vb[w] ? sam[w].y : 0.0f} : top[w] * inv_v[w];
#pragma unroll
    PD_W cb[w] = sam_v[w] * inv_v[w];
#pragma unroll
    PD_W nv += __builtin_popcountll(__builtin_amdgcn_ballot_w64(va[w]) & lanes) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(vb[w]) & lanes);
#pragma unroll
    PD_W g0[w] = -(cb[w] * l0[w]);
#pragma unroll
    PD_W g1[w] = -(cb[w] * l1[w]);
#pragma unroll
    PD_W nbr0[w] = -(cb[w] * r0[w]);
#pragma unroll
    PD_W nbr1[w] = -(cb[w] * r1[w]);
#pragma unroll
    PD_W g0[w] = pd_fma2(ca[w], u2[w], g0[w]);
#pragma unroll
    PD_W g1[w] = pd_fma2(ca[w], v2[w], g1[w]);
#pragma unroll
    PD_W {                                                                   // the sums, step by step
        acc[9] += sam_v[w];
        acc[0] = pd_fma2(nbr0[w], u2[w], pd_fma2(u1[w], g0[w], acc[0]));
        acc[1] = pd_fma2(nbr0[w], v2[w], pd_fma2(u1[w], g1[w], acc[1]));
        acc[2] = pd_fma2(u1[w], ca[w], acc[2]) + nbr0[w];
        acc[3] = pd_fma2(nbr1[w], u2[w], pd_fma2(v1[w], g0[w], acc[3]));
        acc[4] = pd_fma2(nbr1[w], v2[w], pd_fma2(v1[w], g1[w], acc[4]));
        acc[5] = pd_fma2(v1[w], ca[w], acc[5]) + nbr1[w];
        acc[6] += g0[w];
        acc[7] += g1[w];
        acc[8] += ca[w];
    }
#undef PD_W
}

// LDS carve (floats).  Everything lives in the one dynamic region (guide G17).
#define PD_GGS_PSUM_FLOATS (PD_GGS_FAST_FRAMES * 48 > 64 * 16 ? PD_GGS_FAST_FRAMES * 48 : 64 * 16)
#define PD_FR_STRIDE 12   // floats per frame in L.Rc: R_cv (9, row-major) | t_cv (3) -- three 16-byte LDS accesses per frame
#define PD_XS_STRIDE 12   // floats per frame in L.xst / L.mst: the 9 parameters / momenta (+ 3 unused), 16-byte accesses too
#define PD_GGS_LDS_FIXED (64 * PD_FR_STRIDE + 64 * 4 + 8 + 64 * 3 + 64 * 9 + 64 * 4 + 8 + 32 + PD_GGS_PSUM_FLOATS + 2 * 64 * PD_XS_STRIDE)   // + pinc_rows * 16
struct Lds {
    float *Rc;     // [64*12] per frame R_cv (9) | t_cv (3)  (opencv_from_cameras_projection)
    float *fl;     // [64*4]  per frame clamped focal (x, y) | clamp pass-through mask (1/0: x, y)
    float *cam;    // [8]     a0,a1,c0,c1,fbar_x,fbar_y
    float *gT;     // [64*3]  per-frame dL/dT  (un-normalised: not yet divided by n_valid)
    float *gR;     // [64*9]  per-frame dL/dR  (PyTorch3D R, un-normalised)
    float *gA;     // [64*4]  per-frame partial dL/dA {00,02,11,12}
    float *ctl;    // [8]     ctl[0] = stage done flag, ctl[1] = abort
    long long *prof;   // [16] phase cycle counters of the one wave that records them (pd_debug_ggs_prof): in LDS, not in 20 registers of every wave
    float *xst;    // [64*12] pose parameters per frame (lane = frame in P4) -- in LDS, not in registers: wave 0 touches them once per
    float *mst;    // [64*12] iteration, and 18 VGPRs held by every wave for the whole launch is what the match pass cannot spare
    float *pinc;   // [pinc_rows*16] backward results of the current chunk of pairs, one row per (pair, side), frame-sorted
                   //   (pinc_rows = 2 x pairs per chunk, at most PD_GGS_PINC_ROWS; the two-hop kernel always carves the maximum)
    float *psum;   // [PD_GGS_PSUM_FLOATS] general serial path (more than PD_GGS_FAST_FRAMES frames or several chunks of pairs): per-frame
                   //   partial sums across chunks [64*16]
    float *W;      // = psum, fast serial path: per frame the 4 x 9 Jacobian d(R entries)/d(quaternion) of the CURRENT parameters, rows
                   //   padded to 12 floats [PD_GGS_FAST_FRAMES*48] (jac_all)
    float *gq;     // = gR: [64*8] per frame {dL/dq (4), dL/dT (3), -} as P4 reads them (un-normalised: not yet divided by n_valid)
    int4 *itab;    // [n_slots] (first match, count, i, j) of the local items
    int *incoff;   // [PD_GGS_MAX_PCHUNKS][68] per chunk of pairs: CSR offsets of its incidences per frame
    float *F;      // [n_slots*PD_F_STRIDE]
    float *item;   // [n_items*12]
    float *stage;  // [8 waves][2 buffers][STAGE_P KiB] (or [12 waves][1 buffer]) LDS-DMA staging of the match pass (pd_ggs_kernel<STAGE_P > 0>), 1 KiB aligned
};

__device__ __forceinline__ Lds carve(float *base, int n_slots, int pinc_rows, int n_items_cap) {
    Lds L;
    L.Rc = base;
    L.fl = L.Rc + 64 * PD_FR_STRIDE;
    L.cam = L.fl + 64 * 4;
    L.gT = L.cam + 8;
    L.gR = L.gT + 64 * 3;
    L.gA = L.gR + 64 * 9;
    L.ctl = L.gA + 64 * 4;
    L.prof = (long long *)(L.ctl + 8);
    L.xst = L.ctl + 8 + 32;
    L.mst = L.xst + 64 * PD_XS_STRIDE;
    L.pinc = L.mst + 64 * PD_XS_STRIDE;
    L.psum = L.pinc + pinc_rows * 16;
    L.W = L.psum;
    L.gq = L.gR;
    L.itab = (int4 *)(L.psum + PD_GGS_PSUM_FLOATS);
    L.incoff = (int *)(L.itab + n_slots);
    L.F = (float *)(L.incoff + PD_GGS_MAX_PCHUNKS * 68);
    L.item = L.F + n_slots * PD_F_STRIDE;
    L.stage = base + ((((L.item + n_items_cap * PD_ITEM_VALS) - base) + 255) & ~255);   // 1 KiB aligned (base is the LDS origin)
    return L;
}
static size_t ggs_lds_bytes(int n_slots, int n_items, int pinc_rows, int stage_p, int stage_bufs = PD_GGS_WAVES * 2) {
    const size_t f9 = (size_t)n_slots * PD_F_STRIDE;
    size_t b = ((size_t)PD_GGS_LDS_FIXED + (size_t)pinc_rows * 16 + PD_GGS_MAX_PCHUNKS * 68 + f9 + (size_t)n_items * PD_ITEM_VALS) * 4 +
               (size_t)n_slots * 16;
    if (stage_p > 0) b = ((b + 1023) & ~(size_t)1023) + (size_t)stage_bufs * stage_p * 1024;   // 8 waves x 2 buffers, or 12 x 1
    return b;
}

// per-frame tables in LDS, 16 bytes at a time
__device__ __forceinline__ void frame_load(const Lds &L, int n, float (&R)[9], float (&t)[3]) {
    const float4 *p = (const float4 *)(L.Rc + n * PD_FR_STRIDE);
    const float4 a = p[0], b = p[1], c = p[2];
    R[0] = a.x; R[1] = a.y; R[2] = a.z; R[3] = a.w;
    R[4] = b.x; R[5] = b.y; R[6] = b.z; R[7] = b.w;
    R[8] = c.x; t[0] = c.y; t[1] = c.z; t[2] = c.w;
}
__device__ __forceinline__ void params_load(const float *tab, int n, float (&x)[9]) {     // tab = L.xst or L.mst
    const float4 *p = (const float4 *)(tab + n * PD_XS_STRIDE);
    const float4 a = p[0], b = p[1];
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    x[8] = tab[n * PD_XS_STRIDE + 8];
}
__device__ __forceinline__ void params_store(float *tab, int n, const float (&x)[9]) {
    float4 *p = (float4 *)(tab + n * PD_XS_STRIDE);
    p[0] = make_float4(x[0], x[1], x[2], x[3]);
    p[1] = make_float4(x[4], x[5], x[6], x[7]);
    tab[n * PD_XS_STRIDE + 8] = x[8];
}

// decode one frame's 9-vector into R_cv, t_cv, focal (camera_transform.py:80-97 + pytorch3d
// quaternion_to_matrix + opencv_from_cameras_projection); executed by lane n of wave 0.  In three parts, so that a stage
// that leaves R / T / the focal lengths alone (geometry_guided_sampling.py:144-151) does not recompute them.
__device__ __forceinline__ void decode_frame_r(const float *x, float *Rc) {
    const float r = x[3], i = x[4], j = x[5], k = x[6];
    const float two_s = 2.0f * pd_rcp(r * r + i * i + j * j + k * k);
    float R[9];
    R[0] = 1.0f - two_s * (j * j + k * k);
    R[1] = two_s * (i * j - k * r);
    R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r);
    R[4] = 1.0f - two_s * (i * i + k * k);
    R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r);
    R[7] = two_s * (j * k + i * r);
    R[8] = 1.0f - two_s * (i * i + j * j);
    // Rc[a][b] = D[a] * R[b][a], D = diag(-1,-1,1); tc = D * T
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Rc[a * 3 + c] = (a < 2 ? -1.0f : 1.0f) * R[c * 3 + a];
}
__device__ __forceinline__ void decode_frame_t(const float *x, float *tc) {
    tc[0] = -x[0];
    tc[1] = -x[1];
    tc[2] = x[2];
}
__device__ __forceinline__ void decode_frame_fl(const float *x, float &flx, float &fly, float &px, float &py) {
    const float fx = __expf(x[7] + 1.8f), fy = __expf(x[8] + 1.8f);
    px = (fx >= 0.1f && fx <= 20.0f) ? 1.0f : 0.0f;   // torch.clamp backward passes min <= v <= max
    py = (fy >= 0.1f && fy <= 20.0f) ? 1.0f : 0.0f;
    flx = fminf(fmaxf(fx, 0.1f), 20.0f);
    fly = fminf(fmaxf(fy, 0.1f), 20.0f);
}
__device__ __forceinline__ void decode_frame(const float *x, float *Rc, float *tc, float &flx, float &fly,
                                             float &px, float &py) {
    decode_frame_r(x, Rc);
    decode_frame_t(x, tc);
    decode_frame_fl(x, flx, fly, px, py);
}

// wave 0: publish the decoded cameras of the current parameters to LDS (do_*: the parts whose parameters changed)
__device__ __forceinline__ void decode_all(const Lds &L, const float *xr, int lane, int N, const PdSeqDesc &D, bool do_r = true,
                                           bool do_t = true, bool do_fl = true) {
    if (lane < N) {
        float *dst = L.Rc + lane * PD_FR_STRIDE;
        if (do_r) {
            float Rc[9];
            decode_frame_r(xr, Rc);
            ((float4 *)dst)[0] = make_float4(Rc[0], Rc[1], Rc[2], Rc[3]);
            ((float4 *)dst)[1] = make_float4(Rc[4], Rc[5], Rc[6], Rc[7]);
            dst[8] = Rc[8];
        }
        if (do_t) decode_frame_t(xr, dst + 9);
    }
    if (!do_fl) return;                               // (wave-uniform)
    float flx = 0.f, fly = 0.f, px = 0.f, py = 0.f;
    if (lane < N) {
        decode_frame_fl(xr, flx, fly, px, py);
        *(float4 *)&L.fl[lane * 4] = make_float4(flx, fly, px, py);
    }
    // focal_length.mean(dim=0) over all cameras (geometry_guided_sampling.py:142)
    int n_op = N;
    asm volatile("" : "+s"(n_op));                   // formed here every time: hoisted out of the iteration loop the reciprocal becomes a register held
    const float rN = pd_rcp((float)n_op);            // for the whole launch -- in the 168-register variants a spill, reloaded from scratch on the critical path
    float fbx, fby;
    wave_allsum2(flx, fly, fbx, fby);
    fbx *= rN;
    fby *= rN;
    if (lane == 0) {
        const float a0 = pd_rcp(fbx * D.sc), a1 = pd_rcp(fby * D.sc);
        L.cam[0] = a0;
        L.cam[1] = a1;
        L.cam[2] = -D.cx * a0;
        L.cam[3] = -D.cy * a1;
        L.cam[4] = fbx;
        L.cam[5] = fby;
    }
}

// lane `K` of this lane's 16-lane row (DPP row_newbcast); the whole row must be active
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, false));
}

// The chain rule from dL/dR (PyTorch3D R = I + two_s Pm(q), two_s = 2 / |q|^2) to the un-normalised quaternion is LINEAR in dL/dR:
// dL/dq_x = sum_c J[x][c] dL/dR[c], J[x][c] = two_s dPm_c/dq_x - two_s^2 q_x Pm_c.  J depends on the parameters only, so an idle wave
// (lane = frame) forms it while the others compute the next F's -- off the critical path -- and the per-frame sums of the backward
// phase turn into dL/dq with nine multiply-adds per quaternion component.  Stored for the ORDER those sums come in:
// S[m], m = a * 3 + b, = D[a] dL/dRc[a][b] = dL/dR[b][a]  ->  W[frame][x][m] = J[x][b * 3 + a]   (rows padded to 12 floats).
__device__ __forceinline__ void jac_row(const float (&q)[4], const float (&dPx)[9], float qx, float (&w)[9]) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float rn2 = pd_rcp(r * r + i * i + j * j + k * k);
    const float ts = 2.0f * rn2, qs = ts * ts * qx;
    const float Pm[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r, i * k - j * r, j * k + i * r, -(i * i + j * j)};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) w[a * 3 + b2] = ts * dPx[b2 * 3 + a] - qs * Pm[b2 * 3 + a];
}
// row x of the Jacobian of the quaternion q (x is a compile-time constant in jac_all's unrolled loop, a lane value in the general path)
__device__ __forceinline__ void jac_row_x(const float (&q)[4], int x, float (&w)[9]) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    if (x == 0) {
        const float d[9] = {0.0f, -k, j, k, 0.0f, -i, -j, i, 0.0f};
        jac_row(q, d, r, w);
    } else if (x == 1) {
        const float d[9] = {0.0f, j, k, j, -2.0f * i, -r, k, r, -2.0f * i};
        jac_row(q, d, i, w);
    } else if (x == 2) {
        const float d[9] = {-2.0f * j, i, r, i, 0.0f, k, -r, k, -2.0f * j};
        jac_row(q, d, j, w);
    } else {
        const float d[9] = {-2.0f * k, -r, i, r, -2.0f * k, j, i, j, 0.0f};
        jac_row(q, d, k, w);
    }
}
__device__ __forceinline__ void jac_all(const Lds &L, int lane, int N) {
    if (lane >= N || N > PD_GGS_FAST_FRAMES) return;
    const float q[4] = {L.xst[lane * PD_XS_STRIDE + 3], L.xst[lane * PD_XS_STRIDE + 4], L.xst[lane * PD_XS_STRIDE + 5], L.xst[lane * PD_XS_STRIDE + 6]};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        float w[9];
        jac_row_x(q, x, w);
        float4 *dst = (float4 *)(L.W + (lane * 4 + x) * 12);
        dst[0] = make_float4(w[0], w[1], w[2], w[3]);
        dst[1] = make_float4(w[4], w[5], w[6], w[7]);
        dst[2] = make_float4(w[8], 0.0f, 0.0f, 0.0f);
    }
}

// The same for ONE match per lane on plain fp32 VALU (half the issue cycles of a packed step): the tail of an item whose
// last 128-match step would be at most half full (300 matches = 2 packed steps + 44: the third packed step ran 34 % full).
// It accumulates into the .x halves exactly what the packed step accumulates there when its .y match is masked off
// (+0 contributions), so an item's sums do not depend on which of the two forms ran its tail.
template <bool EXACT>
__device__ __forceinline__ void sampson_step1(const float4 pa, bool ina, const float *F, float smax, v2f (&acc)[PD_ITEM_VALS],
                                              float &mind, int &nv) {
#pragma clang fp contract(off)
    const float u1 = pa.x, v1 = pa.y, u2 = pa.z, v2 = pa.w;
    const float l0 = __builtin_fmaf(u1, F[0], __builtin_fmaf(v1, F[3], F[6]));
    const float l1 = __builtin_fmaf(u1, F[1], __builtin_fmaf(v1, F[4], F[7]));
    const float l2 = __builtin_fmaf(u1, F[2], __builtin_fmaf(v1, F[5], F[8]));
    const float r0 = __builtin_fmaf(F[0], u2, __builtin_fmaf(F[1], v2, F[2]));
    const float r1 = __builtin_fmaf(F[3], u2, __builtin_fmaf(F[4], v2, F[5]));
    const float ee = __builtin_fmaf(l0, u2, __builtin_fmaf(l1, v2, l2));
    const float bottom = __builtin_fmaf(r1, r1, __builtin_fmaf(r0, r0, __builtin_fmaf(l1, l1, l0 * l0)));
    const float inv = pd_rcp(bottom);
    const float top = ee * ee;
    float sam;
    if (EXACT) {
        sam = top / bottom;
    } else {
        sam = top * inv;
        mind = fminf(mind, fabsf(sam - smax));
    }
    const bool va = ina && (sam < smax);
    const float inv_v = va ? inv : 0.0f;
    const float ca = ee * inv_v;
    const float sam_v = EXACT ? (va ? sam : 0.0f) : top * inv_v;
    const float cb = sam_v * inv_v;
    acc[9].x += sam_v;
    nv += __builtin_popcountll(__builtin_amdgcn_ballot_w64(va));
    const float g0 = __builtin_fmaf(ca, u2, -(cb * l0)), g1 = __builtin_fmaf(ca, v2, -(cb * l1));
    const float nbr0 = -(cb * r0), nbr1 = -(cb * r1);
    acc[0].x = __builtin_fmaf(nbr0, u2, __builtin_fmaf(u1, g0, acc[0].x));
    acc[1].x = __builtin_fmaf(nbr0, v2, __builtin_fmaf(u1, g1, acc[1].x));
    acc[2].x = __builtin_fmaf(u1, ca, acc[2].x) + nbr0;
    acc[3].x = __builtin_fmaf(nbr1, u2, __builtin_fmaf(v1, g0, acc[3].x));
    acc[4].x = __builtin_fmaf(nbr1, v2, __builtin_fmaf(v1, g1, acc[4].x));
    acc[5].x = __builtin_fmaf(v1, ca, acc[5].x) + nbr1;
    acc[6].x += g0;
    acc[7].x += g1;
    acc[8].x += ca;
}

// where an item's matches come from: registers (resident / streamed through registers) or this wave's LDS staging buffer
// (lane-linear image written by LDS-DMA: match m at byte 16 m)
// Table layout (built at upload, host and device builders alike): inside an item every FULL group of 128 matches is stored
// pair-interleaved -- element lane of the group = (u1_A, u1_B, v1_A, v1_B), element 64 + lane = (u2_A, u2_B, v2_A, v2_B) with A = match
// lane, B = match 64 + lane of the group -- so a full packed step finds its four operand pairs in adjacent registers (8 register
// moves per step less; the pass is bound by VALU cycles).  The remainder of an item (< 128 matches) stays one float4 per match.
struct MatchRegs {
    const float4 (&M)[8];
    __device__ __forceinline__ float4 get(int j, int) const { return M[j]; }
    __device__ __forceinline__ void full_pairs(int j, v2f &u1, v2f &v1, v2f &u2, v2f &v2) const {
        const float4 q0 = M[2 * j], q1 = M[2 * j + 1];
        u1 = (v2f){q0.x, q0.y}; v1 = (v2f){q0.z, q0.w}; u2 = (v2f){q1.x, q1.y}; v2 = (v2f){q1.z, q1.w};
    }
};
struct MatchLds {
    const float4 *B;
    __device__ __forceinline__ float4 get(int j, int lane) const { return B[lane + 64 * j]; }
};


// the (<= 4) two-match steps of an item as straight-line code per step count: without the per-step branch the
// scheduler interleaves the independent steps, which hides the VALU dependency latency two waves per SIMD cannot
// (a FULL step lies wholly inside the item: its range masks are compile-time true and the selects they feed fold away)
#define PD_P2_FULL(j) do { v2f a_, b_, c_, d_; src.full_pairs(j, a_, b_, c_, d_); sampson_step2<EXACT>(a_, b_, c_, d_, true, true, Fm, smax, acc2, mind, nv); } while (0)
#define PD_P2_STEP(j) do { const float4 pa_ = src.get(2 * (j), lane), pb_ = src.get(2 * (j) + 1, lane);                                       \
        sampson_step2<EXACT>((v2f){pa_.x, pb_.x}, (v2f){pa_.y, pb_.y}, (v2f){pa_.z, pb_.z}, (v2f){pa_.w, pb_.w}, (lane + 128 * (j)) < cnt,   \
                             (lane + 128 * (j) + 64) < cnt, Fm, smax, acc2, mind, nv); } while (0)
#define PD_P2_TAIL(j)                                                                                           \
    do {                                                                                                        \
        if (rem > 64 || (!TAIL1 && rem > 0)) PD_P2_STEP(j);                                                     \
        else if (TAIL1 && rem > 0) sampson_step1<EXACT>(src.get(2 * (j), lane), (lane + 128 * (j)) < cnt, Fm, smax, acc2, mind, nv); \
    } while (0)
// TAIL1: run a tail of <= 64 matches as a single-match step (same sums; the variants that keep matches in registers leave it
// off -- they sit at the register limit and are latency-, not issue-bound)
template <bool EXACT, bool TAIL1, typename Src>
__device__ __forceinline__ void item_steps(const Src &src, int cnt, int lane, const float *Fm, float smax,
                                           v2f (&acc2)[PD_ITEM_VALS], float &mind, int &nv) {
    const int full = cnt >> 7, rem = cnt & 127;       // full packed steps; the rest: a packed step, a single-match step or nothing
    if (full >= 4) {
        PD_P2_FULL(0); PD_P2_FULL(1); PD_P2_FULL(2); PD_P2_FULL(3);
    } else if (full == 3) {
        PD_P2_FULL(0); PD_P2_FULL(1); PD_P2_FULL(2); PD_P2_TAIL(3);
    } else if (full == 2) {
        PD_P2_FULL(0); PD_P2_FULL(1); PD_P2_TAIL(2);
    } else if (full == 1) {
        PD_P2_FULL(0); PD_P2_TAIL(1);
    } else {
        PD_P2_TAIL(0);
    }
}
// one work item: the fast pass, and -- when some match of the wave came within the band of the threshold where the
// 1-ulp quotient could decide differently from the IEEE quotient -- the exact pass over the same data instead
template <bool TAIL1, typename Src>
__device__ __forceinline__ void item_pass(const Src &src, int cnt, int lane, const float *Fm, float smax,
                                          v2f (&acc2)[PD_ITEM_VALS], int &nv) {
    float mind = __int_as_float(0x7f800000);
    nv = 0;
#pragma unroll
    for (int c = 0; c < PD_ITEM_VALS; ++c) acc2[c] = (v2f){0.0f, 0.0f};
    item_steps<false, TAIL1>(src, cnt, lane, Fm, smax, acc2, mind, nv);
    const float band = smax * (PD_SAMPSON_BAND_ULPS * 1.1920929e-7f);
    if (__builtin_amdgcn_ballot_w64(mind <= band) != 0ull) {   // wave-uniform, rare (P ~ 1e-7 per match)
#pragma unroll
        for (int c = 0; c < PD_ITEM_VALS; ++c) acc2[c] = (v2f){0.0f, 0.0f};
        nv = 0;
        item_steps<true, false>(src, cnt, lane, Fm, smax, acc2, mind, nv);
    }
}
// fold the two-match partial sums, reduce across the wave: this lane then holds the item total of `slot` (n_valid: the scalar count)
__device__ __forceinline__ float item_totals(const v2f (&acc2)[PD_ITEM_VALS], int nv, int cnt, float smax, int lane, int &slot) {
    float acc[PD_ITEM_VALS];
#pragma unroll
    for (int c = 0; c < 10; ++c) acc[c] = acc2[c].x + acc2[c].y;
    acc[10] = acc[11] = 0.0f;                           // not reduced: finished from the scalar count below
    const float tot = wave_reduce12_transpose(acc, lane, slot);
    const float tot9 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 6));   // lane 6 holds slot 9 (see the slot map)
    if (slot < 9) return tot + tot;                     // the factor 2 of ca, cb
    if (slot == 10) return (float)nv;
    if (slot == 11) return tot9 + smax * (float)(cnt - nv);   // every in-range match that is not valid contributes min(s, max) = max
    return tot;
}

// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land lane-linear at the wave-uniform LDS byte address in M0) straight from
// global memory, no VGPR round trip.  Hand-issued: hipcc neither counts it (so nothing drains it at the next s_barrier and a
// prefetch can cross the serial phases of an iteration) nor waits for it -- every consumer waits with pd_vmcnt<> itself,
// and the kernel drains before it exits (an LDS-DMA landing after the workgroup's LDS was handed on would corrupt it).
// A whole staged item (P pieces of 1 KiB) in ONE statement: wave-uniform 64-bit base in SGPRs, a 32-bit byte offset per lane
// and piece, M0 stepped by 1 KiB between the pieces -- ~3 instructions per piece instead of ~12 (64-bit address arithmetic,
// M0 save / restore and readfirstlane per piece): the match pass is bound by how fast a wave ISSUES instructions.
template <int P>
__device__ __forceinline__ void pd_glds_item(const float4 *base, const unsigned (&off)[6], unsigned lds_dst) {
    static_assert(P == 3 || P == 5 || P == 6, "staging pieces");
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    if constexpr (P == 3)
        asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o0], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o1], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o2], %[b]\n\ts_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep) : [d] "s"(lds_dst), [b] "s"(base), [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]) : "memory", "scc");
    else if constexpr (P == 5)
        asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o0], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o1], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o2], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o3], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o4], %[b]\n\ts_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep) : [d] "s"(lds_dst), [b] "s"(base), [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]),
                       [o3] "v"(off[3]), [o4] "v"(off[4]) : "memory", "scc");
    else
        asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o0], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o1], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o2], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o3], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o4], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o5], %[b]\n\ts_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep) : [d] "s"(lds_dst), [b] "s"(base), [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]),
                       [o3] "v"(off[3]), [o4] "v"(off[4]), [o5] "v"(off[5]) : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void pd_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// --------------------------------------------------------------------------------------------
// the kernel
// --------------------------------------------------------------------------------------------
// STAGE_P > 0: waves that own several items stream them through a per-wave double buffer in LDS filled by LDS-DMA
// (STAGE_P pieces of 1 KiB = up to 64 STAGE_P matches per item), the next item in flight while the current one is computed,
// the first item of the next iteration in flight across the serial phases.  STAGE_P = 0: through registers (any item size).
// RESIDENT: every wave owns at most one item (n_slots == 8, the k = ceil(items / 8) regime): its matches stay in registers for
// the whole launch -- a compile-time variant, so the other variants do not carry those 32 registers.
// NW: waves per workgroup.  8 (two per SIMD, up to 256 VGPRs) everywhere; 12 (three per SIMD, 168 VGPRs: the compiler spills launch
// constants of the serial phases, the match pass itself stays in registers) for the staged k = 1 shape, where the match pass is bound
// by VALU cycles two waves per SIMD leave unused.  Slots, chunks of pairs and the P3 thread roles keep their 8-wave / 512-thread
// geometry (the extra waves only take part in the match pass and the strided loops), so every sum is the one the 8-wave kernel forms;
// with 12 waves a wave has ONE staging buffer: the item is read out of LDS whole, after which the buffer takes the next item.
template <int STAGE_P, bool RESIDENT, int NW = PD_GGS_WAVES>
__global__ __launch_bounds__(NW * 64, NW > PD_GGS_WAVES ? 3 : PD_GGS_MIN_WAVES_PER_SIMD) void pd_ggs_kernel(PdGgsParams P, int B, int n_slots, int pinc_rows, int items_cap) {
    constexpr int NT = NW * 64;                       // threads of this instantiation
    constexpr bool SINGLE = NW > PD_GGS_WAVES;        // one staging buffer per wave
    static_assert(!(RESIDENT && SINGLE) && (NW == PD_GGS_WAVES || STAGE_P > 0), "12 waves: the staged variants only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;       // (kept a vector value: readfirstlane would move what derives from it to SGPRs, of which the kernel has
                                     //  none to spare -- measured 0.1 us per iteration slower at several workgroups per sequence, equal at one)
    const int b = blockIdx.x % B, wg = blockIdx.x / B;   // XCD-aware: see header comment (B: the launch's sequences, padded to 8 if P.xchg_local)
    if (b >= P.n_seqs) return;                           // (padding blocks of the XCD-local placement)
    const PdSeqDesc D = P.seqs[b];
    const int N = P.N, k = P.k;
    const int nW = k * PD_GGS_WAVES;
    const int n_items = D.n_items;
    const Lds L = carve(smem, n_slots, pinc_rows, items_cap);
    const bool p3t = tid < PD_GGS_THREADS;            // takes part in the 512-thread roles of P3
    float *xg = P.x + (size_t)b * N * PD_POSE_DIM;
    u64 *xchg = P.xchg ? P.xchg + (size_t)b * 2 * P.xchg_stride : nullptr;

    // wave 0, lane n owns frame n: parameters + momentum live in LDS (L.xst / L.mst) and visit registers only inside P4
    const bool own = (wave == 0 && lane < N);
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            L.xst[lane * PD_XS_STRIDE + c] = own ? xg[lane * 9 + c] : 0.0f;
            L.mst[lane * PD_XS_STRIDE + c] = 0.0f;
        }
    }
    // local item table -> LDS (slot = wave + 8 * round <-> item = wg*8 + wave + round * nW)
    for (int s = tid; s < n_slots; s += NT) {
        const int item = wg * PD_GGS_WAVES + (s & 7) + (s >> 3) * nW;
        int4 e = make_int4(0, 0, 0, 0);
        if (item < n_items) {
            const int4 it = D.items[item];
            const int2 ij = D.pair_ij[it.x];
            e = make_int4(it.y, it.z, ij.x, ij.y);
        }
        L.itab[s] = e;
    }
    for (int q = tid; q < D.n_pchunks * (N + 1); q += NT) L.incoff[(q / (N + 1)) * 68 + q % (N + 1)] = D.pchunk_off[q];
    if (tid == 0) {
        L.ctl[0] = 0.0f;
        L.ctl[1] = 0.0f;
        L.ctl[3] = 0.0f;
    }
    if (wave == 0) {
        float xr0[9];
        params_load(L.xst, lane, xr0);
        decode_all(L, xr0, lane, N, D);
    }
    __syncthreads();
    // k > 1, XCD-local placement (P.xchg_local: the launch maps block -> (sequence, workgroup) so that the dispatcher's round-robin puts
    // all workgroups of a sequence on one XCD): the per-iteration exchange can then stay in that XCD's L2 -- plain stores instead of
    // write-through agent-scope ones, 1.1 us instead of 1.9 us per exchange of 24 workgroups (tools/xchg_probe.hip).  The placement is
    // the dispatcher's habit, not a guarantee, so it is VERIFIED once per launch: every workgroup publishes its XCC_ID the safe way
    // (agent scope) and all of them read all of them; only if they agree do the stores stay local.  Same answer in every workgroup.
    bool xl = false;
    if (k > 1 && P.xchg_local) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        u64 *ids = xchg + (size_t)P.xchg_stride - 256;          // the last 256 granules of the sequence's first slot (pd_ggs_plan keeps them free)
        if (tid == 0) __hip_atomic_store(ids + wg, (0x7fffffffull << 32) | (u64)(xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true, fail = false;
        if (tid < k) {
            unsigned spins = 0;
            u64 v;
            for (;;) {
                v = __hip_atomic_load(ids + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 32) == 0x7fffffffull) break;
                if (++spins > (1u << 20)) {
                    fail = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            same = !fail && (unsigned)(v & 0xffffffffull) == xcc + 1;
        }
        if (fail) atomicOr(P.err_flag, 1u);
        if (!same) L.ctl[3] = 1.0f;                              // (zeroed before the barrier above; no static LDS: the dynamic region is the whole 160 KiB)
        __syncthreads();
        xl = L.ctl[3] == 0.0f;
    }
    // matches of this wave's first item stay in registers for the whole launch when every wave owns
    // at most one item (the k = ceil(items/8) regime): no per-iteration match traffic at all
    constexpr bool resident = RESIDENT;
    float4 mres[RESIDENT ? 8 : 1];
    if (RESIDENT) {
        const int4 e = L.itab[wave];
        const int last = e.y > 0 ? e.y - 1 : 0;
        const float4 *pts = D.pts + e.x;
#pragma unroll
        for (int st = 0; st < (RESIDENT ? 8 : 1); ++st) {
            const int m = lane + 64 * st;
            mres[st] = (e.y > 0) ? pts[m < e.y ? m : last] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // LDS staging of the match pass (STAGE_P > 0, several items per wave): this wave's items, its double buffer, and the
    // first item already on its way
    // Items of this workgroup are slots 0 .. n_local-1 (slot q <-> item wg*8 + (q & 7) + (q >> 3) * nW, increasing in q).
    // A wave starts every iteration with its own slot `wave` and then PULLS further slots from a workgroup-wide counter:
    // the two waves that share a SIMD do not issue at the same rate (the older one gets the VALU first), and a static
    // round-robin leaves the faster half idle for the last quarter of the match pass.  Which wave computes an item does
    // not enter its sums, so results stay bitwise reproducible.
    int n_local = 0;
    if (n_items > wg * PD_GGS_WAVES) {
        const int d = n_items - wg * PD_GGS_WAVES, r0 = d / nW;
        n_local = r0 * PD_GGS_WAVES + min(d - r0 * nW, PD_GGS_WAVES);
    }
    int *q_ctr = (int *)&L.ctl[4];
#ifdef PD_GGS_PROF2
    if (P.prof_wave & 0x100) {                   // experiment: only one wave per SIMD works in the match pass
        if (wave >= 4) n_local = 0;
    }
#endif
    const bool staged = STAGE_P > 0 && !resident && wave < n_local;
    const unsigned stage_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(L.stage + wave * ((SINGLE ? 1 : 2) * STAGE_P * 256)));
    const float4 *stage_ptr = (const float4 *)(L.stage + wave * ((SINGLE ? 1 : 2) * STAGE_P * 256));
    int pb = 0;                                   // buffer that holds (or is receiving) the item computed next
    // byte offsets of this lane's match in each piece (lane + 64 q), clamped per item to its last match
    // (the byte offsets lane * 16 + 1024 q are formed where they are used: six registers held across the whole launch are six spills in the
    // 168-register variants, whose reloads from scratch land in the serial phases)
    auto stage_item = [&](int slot, int buf) {
        int4 e = L.itab[slot];
        const int first = __builtin_amdgcn_readfirstlane(e.x);
        const unsigned last16 = (unsigned)(__builtin_amdgcn_readfirstlane(e.y) - 1) * 16u;
        if constexpr (STAGE_P > 0) {
            unsigned off[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) off[q] = min((unsigned)lane * 16u + 1024u * q, last16);     // no predicated loads: a clamped copy of the last match
            pd_glds_item<STAGE_P>(D.pts + first, off, stage_lds + (unsigned)(buf * STAGE_P) * 1024u);
        }
    };
    if (staged) stage_item(wave, 0);

    // ---- geometry of the serial phases ------------------------------------------------------------------------------------------
    // FAST (<= PD_GGS_FAST_FRAMES frames, one chunk of pairs -- every BASELINE config of this kernel): the rows of the pair backward are
    // laid out per frame at a FIXED stride `cap` (the largest number of pairs incident to a frame, rounded up to 4; unused rows stay
    // zero for the whole launch), thread (fb_n, fb_c) = (frame, component) of the first 16 N threads sums its column without index
    // clamps or selects, turns the nine dL/dR sums into dL/dq inside its 16-lane row (row_bcast + the Jacobian of jac_all) and hands P4
    // seven numbers per frame; waves 6 and 7 form the dL/dA and loss totals beside them.
    // GENERAL (more frames or several chunks of pairs): CSR rows, partial sums carried across chunks in LDS, as before.
    const int fb_n = tid >> 4, fb_c = tid & 15;
    const float fb_sign = (fb_c < 9) ? (fb_c < 6 ? -1.0f : 1.0f) : ((fb_c < 11) ? -1.0f : 1.0f);   // D = diag(-1,-1,1): rows a < 2 of dL/dRc, entries < 2 of dL/dtc
    int cap = 0;
    if (D.n_pchunks == 1 && N <= PD_GGS_FAST_FRAMES) {
        int dmax = 0;
        for (int n = 0; n < N; ++n) dmax = max(dmax, L.incoff[n + 1] - L.incoff[n]);
        cap = (dmax + 3) & ~3;
    }
    // a frame's rows start `rstride` = cap + 1 rows apart: with cap % 4 == 0 the four frames a wave sums then sit 16 banks apart -- 64 lanes
    // on 64 distinct LDS banks (at a stride of cap rows all four would share 16 banks: every column load a four-way conflict)
    const int rstride = cap + 1;
    const bool fast34 = cap > 0 && N * rstride <= pinc_rows;    // (block-uniform)
    const int n_row_waves = (N * 16 + 63) >> 6;                 // waves that hold (frame, component) threads in the fast path (<= 6)
    const int ga_parts = fast34 ? n_row_waves : 1;              // dL/dA partials P4 adds up (fast: one per row wave; general: the totals)
    constexpr int W_LOSS = PD_GGS_WAVES - 1;                    // idle in the fast backward phase: forms the loss totals meanwhile
    // pair-level backward in chunks of PD_GGS_THREADS pairs (one chunk up to N = 32); chunk 0's table entry is hoisted
    // (held for the whole launch in THREE registers: frames i | j << 8 and the item count share one -- pd_ggs_set_matches bounds both to 16 bits)
    int4 my_pair = (p3t && tid < D.n_pairs) ? D.ptab[tid] : make_int4(0, 0, 0, 0);
    if (fast34) {
        if (p3t && tid < D.n_pairs) {       // CSR positions -> fixed-stride rows
            const int pi = my_pair.x & 0xff, pj = my_pair.x >> 8;
            const int r0 = pi * rstride + ((my_pair.w & 0xffff) - L.incoff[pi]), r1 = pj * rstride + ((my_pair.w >> 16) - L.incoff[pj]);
            my_pair.w = r0 | (r1 << 16);
        }
        for (int q = tid; q < N * rstride * 4; q += NT) ((float4 *)L.pinc)[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const int my_pair_xz = (my_pair.x & 0xffff) | (my_pair.z << 16), my_pair_y = my_pair.y, my_pair_w = my_pair.w;
    unsigned epoch = 0;
    int trace_row = 0;
    // the in-kernel cycle counters cost 24 VGPRs for the whole launch: compiled out of the 12-wave variants, which run at the
    // 168-register limit (build with -DPD_GGS_PROF12 to study those; pd_ggs_plan keeps 8 waves while profiling is on otherwise)
    constexpr bool HAS_PROF = !SINGLE || PD_GGS_PROF12;
    const bool prof = HAS_PROF && P.prof != nullptr && blockIdx.x == 0 && wave == (P.prof_wave & 7);   // one wave of WG 0
    // (build with -DPD_GGS_PROF2 for the timers INSIDE the match pass: claim / DMA issue / wait / pass / reduce -> counters 10..14)
    long long pc = 0, pq = 0;
    if (prof && lane == 0) {
        for (int i = 0; i < 16; ++i) L.prof[i] = 0;
    }
#define PD_PROF(i) do { if (prof) { const long long _n = __builtin_readcyclecounter(); if (lane == 0) L.prof[i] += _n - pc; pc = _n; } } while (0)
    const float inv_M = 1.0f / (float)D.M;
    for (int st = 0; st < P.n_stages; ++st) {
        const PdGgsStage S = P.stages[st];
        int stepped = 0;
        // {printed statistic, valid count, loss} of the stage's last iteration: LDS (L.ctl[5..7], written by P4's lane 0), not three registers
        if (tid == 0) {
            L.ctl[5] = __int_as_float(0x7fc00000);
            L.ctl[6] = 0.0f;
            L.ctl[7] = __int_as_float(0x7fc00000);
        }
        for (int it = 0; it < S.iters; ++it) {
            if (prof) pc = __builtin_readcyclecounter();
            // ---- P1: F for the pairs of this workgroup's items -------------------------------
            const Cam cam = {L.cam[0], L.cam[1], L.cam[2], L.cam[3]};
            if (tid == 0) *q_ctr = NW;                    // first slot the match pass hands out dynamically
            for (int s = tid; s < n_slots && !(PD_GGS_ABLATE & 1); s += NT) {
                const int4 e = L.itab[s];
                if (e.y > 0) {
                    float Ri[9], Rj[9], ti[3], tj[3];
                    frame_load(L, e.z, Ri, ti);
                    frame_load(L, e.w, Rj, tj);
                    PairFwd f;
                    pair_forward(Ri, ti, Rj, tj, f);
                    float F[9];
                    fundamental_from_E(f.E, cam, F);
#pragma unroll
                    for (int c = 0; c < 9; ++c) L.F[s * PD_F_STRIDE + c] = F[c];
                }
            }
            __syncthreads();
            PD_PROF(0);

            // ---- P2: per-match Sampson residual + dL/dF, one (pair, chunk) item per wave ------
            ++epoch;
            // slot claims run one item ahead: the LDS atomic for the item after the next one is issued before the pass and
            // consumed after it, so its latency is never exposed (a wave over-claims one slot per iteration: harmless)
            int s_next = n_local;
            if constexpr (!RESIDENT) {
                int t0 = 0;
                if (lane == 0 && wave < n_local) t0 = atomicAdd(q_ctr, 1);
                s_next = __builtin_amdgcn_readfirstlane(t0);
            }
            for (int s = wave; s < n_local && !(PD_GGS_ABLATE & 16);) {
                const int item = wg * PD_GGS_WAVES + (s & 7) + (s >> 3) * nW;
#ifdef PD_GGS_PROF2
#define PD_PROF2(i) do { if (prof) { const long long _n = __builtin_readcyclecounter(); if (lane == 0) L.prof[i] += _n - pq; pq = _n; } } while (0)
                if (prof) pq = __builtin_readcyclecounter();
#else
#define PD_PROF2(i) do { } while (0)
#endif
                int t_claim = 0;
                if constexpr (!RESIDENT) {
                    if (lane == 0) t_claim = atomicAdd(q_ctr, 1);
                }
                PD_PROF2(10);
                int4 e = L.itab[s];
                e.x = __builtin_amdgcn_readfirstlane(e.x);      // wave-uniform by construction: lets the step-count branches be scalar
                e.y = __builtin_amdgcn_readfirstlane(e.y);
                float Fm[9];
                {   // three LDS reads (slot stride 12 floats, 16-byte aligned) instead of nine 4-byte ones
                    const float4 f0 = *(const float4 *)(L.F + s * PD_F_STRIDE), f1 = *(const float4 *)(L.F + s * PD_F_STRIDE + 4);
                    Fm[0] = f0.x; Fm[1] = f0.y; Fm[2] = f0.z; Fm[3] = f0.w;
                    Fm[4] = f1.x; Fm[5] = f1.y; Fm[6] = f1.z; Fm[7] = f1.w;
                    Fm[8] = L.F[s * PD_F_STRIDE + 8];
                }
                // two 64-match steps per pass: lane handles matches lane + 64*(2j) and lane + 64*(2j+1) together
                v2f acc2[PD_ITEM_VALS];
                int nv;
                if constexpr (RESIDENT) {   // straight from the resident registers (no copies)
                    item_pass<false>(MatchRegs{mres}, e.y, lane, Fm, P.sampson_max, acc2, nv);
                } else if constexpr (STAGE_P > 0) {
                    // the slot this wave computes next (or its own first one, for the next iteration: the matches never
                    // change) goes into the other buffer while this one is computed; STAGE_P pieces stay in flight
                    if constexpr (!SINGLE) stage_item(s_next < n_local ? s_next : wave, pb ^ 1);
                    PD_PROF2(11);
                    pd_vmcnt<SINGLE ? 0 : STAGE_P>();
                    PD_PROF2(12);
                    // the whole item out of LDS at once (one exposed LDS latency instead of one per step)
                    float4 mb[8];
                    {
                        const float4 *Bp = stage_ptr + (SINGLE ? 0 : pb) * (STAGE_P * 64) + lane;
#pragma unroll
                        for (int q = 0; q < 8; ++q) mb[q] = q < STAGE_P ? Bp[64 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if constexpr (SINGLE) {     // the item is in registers: its buffer takes the next one while this one is computed
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mb[0].x), "+v"(mb[1].x), "+v"(mb[2].x), "+v"(mb[3].x), "+v"(mb[4].x), "+v"(mb[5].x) :: "memory");
                        stage_item(s_next < n_local ? s_next : wave, 0);
                    }
                    item_pass<true>(MatchRegs{mb}, e.y, lane, Fm, P.sampson_max, acc2, nv);
                    pb ^= 1;
                } else {
                    // stream this item: all (<= 8) lines in flight at once, indices clamped (no
                    // predicated loads), out-of-range lanes are masked in the arithmetic instead
                    float4 mb[8];
                    const float4 *pts = D.pts + e.x;
                    const int last = e.y - 1;
#pragma unroll
                    for (int st = 0; st < 8; ++st) {
                        const int m = lane + 64 * st;
                        mb[st] = pts[m < e.y ? m : last];
                    }
                    item_pass<false>(MatchRegs{mb}, e.y, lane, Fm, P.sampson_max, acc2, nv);
                }
#ifdef PD_GGS_PROF2
                if (prof) { acc2[0].x += 0.0f * (float)__builtin_amdgcn_readfirstlane(__float_as_int(acc2[9].y)); }   // (keeps the pass before the timer)
#endif
                PD_PROF2(13);
                int slot;
                const float tot = item_totals(acc2, nv, e.y, P.sampson_max, lane, slot);   // this lane holds the item total of `slot`
                if (lane < 16 && slot < PD_ITEM_VALS) {
                    if (k == 1) {
                        L.item[item * PD_ITEM_VALS + slot] = tot;
                    } else {
                        u64 *g = xchg + (size_t)(epoch & 1) * P.xchg_stride + (size_t)item * PD_XCHG_LINE + slot;
                        const u64 gv = ((u64)epoch << 32) | (u64)__float_as_uint(tot);
                        if (xl) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g), "v"(gv) : "memory");   // stays in this XCD's L2 (the readers' sc1 loads find it there)
                        else __hip_atomic_store(g, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                PD_PROF2(14);
                s = s_next;
                s_next = RESIDENT ? n_local : __builtin_amdgcn_readfirstlane(t_claim);
            }
            PD_PROF(1);
            if (k > 1) {
                // all-gather of every item's 12 sums: the data IS the flag (tag == epoch)
                const u64 *slot = xchg + (size_t)(epoch & 1) * P.xchg_stride;
                bool fail = false;
                // each item is one 128-byte line of 16 granules (12 used); a thread fetches 16-byte pieces
                // (2 granules) with write-through-coherent (sc1) loads, up to 3 pieces in flight per pass
                const int n_piece = (PD_GGS_ABLATE & 32) ? 0 : n_items * 6;
                for (int p0 = tid; p0 < n_piece; p0 += 3 * NT) {
                    const u64 *a[3];
                    int pi_[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int pc = p0 + u * NT;
                        pi_[u] = pc < n_piece ? pc : p0;
                        a[u] = slot + (size_t)(pi_[u] / 6) * PD_XCHG_LINE + (pi_[u] % 6) * 2;
                    }
                    u32x4 v0, v1, v2;
                    unsigned spins = 0;
                    for (;;) {
                        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                                     "global_load_dwordx4 %1, %4, off sc1\n\t"
                                     "global_load_dwordx4 %2, %5, off sc1\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : "=&v"(v0), "=&v"(v1), "=&v"(v2)
                                     : "v"(a[0]), "v"(a[1]), "v"(a[2])
                                     : "memory");
                        const bool ok = v0[1] == epoch && v0[3] == epoch && v1[1] == epoch && v1[3] == epoch &&
                                        v2[1] == epoch && v2[3] == epoch;
                        if (ok) break;
                        if (++spins > (1u << 20) ||
                            ((spins & 255u) == 0 &&
                             __hip_atomic_load(P.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            fail = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    const u32x4 vv[3] = {v0, v1, v2};
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        if (p0 + u * NT < n_piece) {
                            const int g = (pi_[u] / 6) * PD_ITEM_VALS + (pi_[u] % 6) * 2;
                            L.item[g] = __uint_as_float(vv[u][0]);
                            L.item[g + 1] = __uint_as_float(vv[u][2]);
                        }
                    }
                }
                if (fail) {
                    atomicOr(P.err_flag, 1u);
                    L.ctl[1] = 1.0f;
                }
            }
            __syncthreads();
            if (L.ctl[1] != 0.0f) {   // a bounded spin gave up: abort the whole workgroup
                if (STAGE_P > 0) pd_vmcnt<0>();
                return;
            }
            PD_PROF(2);

            // ---- P3a: pair backward, one thread per frame pair, in chunks of PD_GGS_THREADS pairs (one chunk up to 32 frames) ----
            // results go to LDS rows of 16 floats per (pair, side): 9 dL/dRc_side + 3 dL/dtc_side + the pair's 4 dL/dA partials (side 0;
            // zeros on side 1).  P3b sums the first twelve per frame in fixed order; of the last four only the sum over ALL rows is needed
            // (the focal length is the mean over frames, geometry_guided_sampling.py:142): an idle wave forms it
            const bool need_rt = S.update_R || S.update_T;
            // totals over all items by one wave: one 16-byte read per item gets {dF22, sum(s valid), n_valid, sum(min(s, max))}
            auto loss_totals = [&]() {
                float s_sum = 0.0f, s_cnt = 0.0f, s_cl = 0.0f;
                for (int q2 = lane; q2 < n_items; q2 += 64) {
                    const float4 v4 = *(const float4 *)&L.item[q2 * PD_ITEM_VALS + 8];
                    s_sum += v4.y;
                    s_cnt += v4.z;
                    s_cl += v4.w;
                }
                s_sum = wave_allsum(s_sum);
                s_cnt = wave_allsum(s_cnt);
                s_cl = wave_allsum(s_cl);
                if (lane == 0) {
                    L.cam[6] = s_sum;
                    L.cam[7] = s_cnt;
                    L.ctl[2] = s_cl;
                }
            };
            for (int ck = 0; ck < D.n_pchunks; ++ck) {
                const int *coff = L.incoff + ck * 68;   // this chunk's incidences by frame (CSR positions within L.pinc; general path)
                {
                    // ONE thread per frame pair runs the shared backward chain once and writes both sides' results straight into
                    // their rows.  190 threads = 3 waves, one per SIMD: the cost is per wave-instruction.
                    const int pair = ck * PD_GGS_THREADS + tid;
                    if (p3t && pair < D.n_pairs && !(PD_GGS_ABLATE & 2)) {
                        const int4 mp = (ck == 0) ? make_int4(my_pair_xz & 0xffff, my_pair_y, (int)((unsigned)my_pair_xz >> 16), my_pair_w) : D.ptab[pair];
                        const int pi = mp.x & 0xff, pj = mp.x >> 8, nit = mp.z;
                        float G[9];
#pragma unroll
                        for (int c = 0; c < 9; ++c) G[c] = 0.0f;
                        for (int u = 0; u < nit; ++u)
#pragma unroll
                            for (int c = 0; c < 9; ++c) G[c] += L.item[(mp.y + u) * PD_ITEM_VALS + c];
#include "pd_ggs_pairbwd.inc"
                    }
                    // the quaternion Jacobian of this iteration's parameters, on a wave the pair backward leaves idle (<= 276 pairs: waves 0 .. 4), read by P3b behind the
                    // barrier below.  (Round 6: until here it ran at the top of the iteration, where a one-item-per-wave sequence -- B = 1, k = 24 -- has nothing to hide it
                    // behind: every wave waited at P1's barrier for this one.)
                    if (ck == 0 && wave == PD_GGS_WAVES - 1 && S.update_R && fast34) jac_all(L, lane, N);
                }
                if (prof) { pq = __builtin_readcyclecounter(); if (lane == 0) L.prof[6] += pq - pc; }
                __syncthreads();
                if (prof) { const long long n_ = __builtin_readcyclecounter(); if (lane == 0) L.prof[7] += n_ - pq; pq = n_; }
                // ---- P3b: per-frame sums over the rows of this chunk, fixed (ascending) order ----
                if (PD_GGS_ABLATE & 4) {
                } else if (fast34) {
#include "pd_ggs_p3b.inc"
                    if (wave >= n_row_waves && wave == W_LOSS) {
                        loss_totals();
                    }
                } else {         // several passes over the frames, partial sums carried across chunks in LDS
                    for (int n0 = 0; n0 < N; n0 += PD_GGS_THREADS / 16) {
                        const int n = n0 + fb_n;
                        if (p3t && n < N) {
                            const int lo = coff[n], hi = coff[n + 1];
                            float acc2 = (ck == 0) ? 0.0f : L.psum[n * 16 + fb_c];
                            for (int e = lo; e < hi; e += 16) {   // 16 LDS loads in flight, summed in order
                                float t16[16];
#pragma unroll
                                for (int u = 0; u < 16; ++u) t16[u] = L.pinc[min(e + u, hi - 1) * 16 + fb_c];
#pragma unroll
                                for (int u = 0; u < 16; ++u) acc2 += (e + u < hi) ? t16[u] : 0.0f;
                            }
                            L.psum[n * 16 + fb_c] = acc2;
                        }
                    }
                }
                if (prof) { const long long n_ = __builtin_readcyclecounter(); if (lane == 0) L.prof[8] += n_ - pq; pq = n_; }
                if (ck + 1 < D.n_pchunks) __syncthreads();   // the rows (and the dL/dA slots) are rewritten by the next chunk
            }
            if (!fast34) {
                if (wave == W_LOSS) loss_totals();
                __syncthreads();                         // every frame's sums are complete
                // the same seven numbers per frame as the fast path hands to P4: dL/dq through the Jacobian, dL/dT (signs: D = diag(-1,-1,1))
                for (int q = tid; q < N * 8; q += NT) {
                    const int n = q >> 3, x = q & 7;
                    float v = 0.0f;
                    if (x < 4) {
                        if (S.update_R) {
                            const float qn[4] = {L.xst[n * PD_XS_STRIDE + 3], L.xst[n * PD_XS_STRIDE + 4], L.xst[n * PD_XS_STRIDE + 5], L.xst[n * PD_XS_STRIDE + 6]};
                            float Wr[9];
                            jac_row_x(qn, x, Wr);
                            const float *ps = L.psum + n * 16;
#pragma unroll
                            for (int m = 0; m < 9; ++m) v = __builtin_fmaf(m < 6 ? -ps[m] : ps[m], Wr[m], v);
                        }
                    } else if (x < 7) {
                        if (S.update_T) {
                            const float t = L.psum[n * 16 + 9 + (x - 4)];
                            v = (x - 4 < 2) ? -t : t;
                        }
                    }
                    L.gq[q] = v;
                }
                // dL/dA totals over the frames, in frame order
                if (tid < 4) {
                    float v = 0.0f;
                    for (int n = 0; n < N; ++n) v += L.psum[n * 16 + 12 + tid];
                    L.gA[tid] = v;
                }
            }
            __syncthreads();
            PD_PROF(3);

#include "pd_ggs_p4q.inc"
            __syncthreads();
            PD_PROF(4);
            if (prof && lane == 0) L.prof[5] += 1;
            if (L.ctl[0] != 0.0f) break;
        }
        if (wave == 0 && lane == 0 && wg == 0 && P.stats) {
            float *so = P.stats + ((size_t)b * P.n_stages + st) * 4;
            so[0] = L.ctl[5];
            so[1] = (float)stepped;
            so[2] = L.ctl[6];
            so[3] = L.ctl[7];
        }
        if (P.eval_only) break;
    }
    if (prof && lane == 0) {
        for (int i = 0; i < 16; ++i) P.prof[i] = L.prof[i];
    }
    if (own && wg == 0 && !P.eval_only) {
#pragma unroll
        for (int c = 0; c < 9; ++c) xg[lane * 9 + c] = L.xst[lane * PD_XS_STRIDE + c];
    }
    if (STAGE_P > 0) pd_vmcnt<0>();   // the look-ahead LDS-DMA of the last item must land before the LDS is handed on
}

// --------------------------------------------------------------------------------------------
// the two-hop kernel for many frames (N > 32: several chunks of pairs)
//
// pd_ggs_kernel lets EVERY workgroup of a sequence gather all item sums and back-propagate all pairs;
// that replication is cheap at N = 20 (190 pairs) and dominates at N = 50 (1 225 pairs: 18 MB of
// exchange reads and 3 chunks of pair backward per iteration and workgroup).  Here the backward is
// distributed instead -- same arithmetic per pair and per frame, two small exchanges per iteration:
//   P1/P2  as before, but a workgroup keeps its items' sums to itself (needs one item per pair);
//   P3a    it back-propagates only ITS pairs and publishes the two 16-float results of each pair as
//          one exchange line per (pair, side), at the row the frame-sorted order gives it   (hop 1)
//   P3b    the owner of frame n (workgroup n % k) gathers that frame's rows, sums them in row order and
//          publishes the frame's 16 gradient sums; every workgroup also publishes its loss totals   (hop 2)
//   P4     every workgroup gathers the N frame lines + k total lines (a few KB) and runs the update.
// Exchange lines live in the sequence's slot of the same tagged-granule buffer:
//   [0, n_inc) (pair, side) rows | [n_inc, n_inc + k) per-workgroup totals | [n_inc + k, + N) per-frame sums.
// --------------------------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ bool ggs2_gather(const u64 *src_lines, int piece0, int n_piece, int pieces_per_line, unsigned epoch,
                                            float *dst, int dst_stride, unsigned *err_flag) {
    // piece p = (line p / pieces_per_line, 16-byte part p % pieces_per_line) -> dst[line * dst_stride + 2 * part .. + 1]
    bool fail = false;
    for (int p0 = piece0; p0 < n_piece; p0 += U * PD_GGS_THREADS) {
        unsigned spins = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * PD_GGS_THREADS;
            if (p < n_piece) {
                const int line = p / pieces_per_line, part = p - line * pieces_per_line;
                const u64 *a = src_lines + (size_t)line * PD_XCHG_LINE + part * 2;
                u32x4 v;
                for (;;) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
                    if (v[1] == epoch && v[3] == epoch) break;
                    if (++spins > (1u << 20) ||
                        ((spins & 255u) == 0 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        fail = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                dst[line * dst_stride + part * 2] = __uint_as_float(v[0]);
                dst[line * dst_stride + part * 2 + 1] = __uint_as_float(v[2]);
            }
        }
    }
    return !fail;
}

__global__ __launch_bounds__(PD_GGS_THREADS) void pd_ggs2_kernel(PdGgsParams P, int B, int n_slots) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x % B, wg = blockIdx.x / B;
    if (b >= P.n_seqs) return;
    const PdSeqDesc D = P.seqs[b];
    const int N = P.N, k = P.k;
    const int nW = k * PD_GGS_WAVES;
    const int n_items = D.n_items;          // == D.n_pairs (one item per pair)
    const int n_inc = 2 * D.n_pairs;
    const Lds L = carve(smem, n_slots, PD_GGS_PINC_ROWS, n_slots);
    float *xg = P.x + (size_t)b * N * PD_POSE_DIM;
    u64 *xbase = P.xchg + (size_t)b * 2 * P.xchg_stride;
    // LDS reuse: L.item holds this workgroup's item sums [n_slots][12]; L.pinc rows [0, 2 n_slots <= 512) the results of
    // its pairs, rows [512, 576) the gathered rows of an owned frame, rows [640, 704) the gathered totals [k <= 256][4],
    // rows [768, 800) the exchange row of each local (pair, side); L.psum the gathered frame sums [N][16]
    float *own_rows = L.pinc;
    float *frame_rows = L.pinc + 512 * 16;
    float *tot_rows = L.pinc + 640 * 16;
    int *grow = (int *)(L.pinc + 768 * 16);

    const bool own = (wave == 0 && lane < N);
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            L.xst[lane * PD_XS_STRIDE + c] = own ? xg[lane * 9 + c] : 0.0f;
            L.mst[lane * PD_XS_STRIDE + c] = 0.0f;
        }
    }
    for (int s = tid; s < n_slots; s += PD_GGS_THREADS) {
        const int item = wg * PD_GGS_WAVES + (s & 7) + (s >> 3) * nW;
        int4 e = make_int4(0, 0, 0, 0);
        int2 gp = make_int2(0, 0);
        if (item < n_items) {
            const int4 it = D.items[item];
            const int2 ij = D.pair_ij[it.x];
            e = make_int4(it.y, it.z, ij.x, ij.y);
            gp = D.gpos[it.x];
        }
        L.itab[s] = e;
        grow[2 * s] = gp.x;
        grow[2 * s + 1] = gp.y;
    }
    for (int q = tid; q <= N; q += PD_GGS_THREADS) L.incoff[q] = D.ginc_off[q];
    if (tid == 0) {
        L.ctl[0] = 0.0f;
        L.ctl[1] = 0.0f;
    }
    if (wave == 0) {
        float xr0[9];
        params_load(L.xst, lane, xr0);
        decode_all(L, xr0, lane, N, D);
    }
    __syncthreads();
    // one item per wave (the usual case here: k = ceil(pairs / 8)): its matches stay in registers for the whole launch
    const bool resident = (n_slots == PD_GGS_WAVES);
    float4 mres[8];
    {
        const int4 e = L.itab[wave];
        const int last = e.y > 0 ? e.y - 1 : 0;
        const float4 *pts = D.pts + e.x;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = lane + 64 * q;
            mres[q] = (resident && e.y > 0) ? pts[m < e.y ? m : last] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    unsigned epoch = 0;
    int trace_row = 0;
    const float inv_M = 1.0f / (float)D.M;
    // phase clocks (pd_debug_ggs_prof; round 5): wave 0 of workgroup 0 (owner of frame 0) -> prof[0..7], of the last workgroup (owns no frame when
    // k > N) -> prof[8..15]: {P1, P2, P3a, hop-1 publish + totals, P3b owner loop, hop-2 gathers, frame gradients + totals, P4}, shader cycles
    const bool prof2 = P.prof != nullptr && b == 0 && wave == 0 && (wg == 0 || wg == k - 1);
    long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0, qc = 0;
#define PD_PROF2H(acc) do { if (prof2) { const long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - qc; qc = n_; } } while (0)
    for (int st = 0; st < P.n_stages; ++st) {
        const PdGgsStage S = P.stages[st];
        int stepped = 0;
        float last_print = __int_as_float(0x7fc00000), last_cnt = 0.0f, last_loss = __int_as_float(0x7fc00000);
        const bool need_rt = S.update_R || S.update_T;
        for (int it = 0; it < S.iters; ++it) {
            if (prof2) qc = __builtin_amdgcn_s_memtime();
            // ---- P1: F for the pairs of this workgroup's items
            const Cam cam = {L.cam[0], L.cam[1], L.cam[2], L.cam[3]};
            for (int s = tid; s < n_slots; s += PD_GGS_THREADS) {
                const int4 e = L.itab[s];
                if (e.y > 0) {
                    float Ri[9], Rj[9], ti[3], tj[3];
                    frame_load(L, e.z, Ri, ti);
                    frame_load(L, e.w, Rj, tj);
                    PairFwd f;
                    pair_forward(Ri, ti, Rj, tj, f);
                    float F[9];
                    fundamental_from_E(f.E, cam, F);
#pragma unroll
                    for (int c = 0; c < 9; ++c) L.F[s * PD_F_STRIDE + c] = F[c];
                }
            }
            __syncthreads();
            PD_PROF2H(q0);
            // ---- P2: per-match Sampson residual + dL/dF, a wave per item; the 12 sums stay in this workgroup's LDS
            ++epoch;
            u64 *xs = xbase + (size_t)(epoch & 1) * P.xchg_stride;
            for (int r = 0; r * PD_GGS_WAVES < n_slots; ++r) {
                const int s = wave + PD_GGS_WAVES * r;
                const int4 e = L.itab[s];
                if (e.y > 0) {
                    float Fm[9];
                    {   // three LDS reads (slot stride 12 floats, 16-byte aligned) instead of nine 4-byte ones
                    const float4 f0 = *(const float4 *)(L.F + s * PD_F_STRIDE), f1 = *(const float4 *)(L.F + s * PD_F_STRIDE + 4);
                    Fm[0] = f0.x; Fm[1] = f0.y; Fm[2] = f0.z; Fm[3] = f0.w;
                    Fm[4] = f1.x; Fm[5] = f1.y; Fm[6] = f1.z; Fm[7] = f1.w;
                    Fm[8] = L.F[s * PD_F_STRIDE + 8];
                }
                    v2f acc2[PD_ITEM_VALS];
                    int nv;
                    if (resident) {
                        item_pass<false>(MatchRegs{mres}, e.y, lane, Fm, P.sampson_max, acc2, nv);
                    } else {
                        float4 mb[8];
                        const float4 *pts = D.pts + e.x;
                        const int last = e.y - 1;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int m = lane + 64 * q;
                            mb[q] = pts[m < e.y ? m : last];
                        }
                        item_pass<false>(MatchRegs{mb}, e.y, lane, Fm, P.sampson_max, acc2, nv);
                    }
                    int slot;
                    const float tot = item_totals(acc2, nv, e.y, P.sampson_max, lane, slot);
                    if (lane < 16 && slot < PD_ITEM_VALS) L.item[s * PD_ITEM_VALS + slot] = tot;
                }
            }
            __syncthreads();
            PD_PROF2H(q1);
            // ---- P3a: backward of this workgroup's pairs (thread per local item), rows 2s (side 0), 2s + 1 (side 1)
            if (tid < n_slots && L.itab[tid].y > 0) {
                const int4 e = L.itab[tid];
                const int pi = e.z, pj = e.w;
                const int4 mp = make_int4(0, 0, 1, (2 * tid) | ((2 * tid + 1) << 16));
                float G[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) G[c] = L.item[tid * PD_ITEM_VALS + c];
#include "pd_ggs_pairbwd.inc"
            }
            __syncthreads();
            PD_PROF2H(q2);
            // ---- hop 1: publish the (pair, side) rows; thread (row = tid / 16, component = tid % 16), 32 rows per pass
            for (int r0 = 0; r0 < 2 * n_slots; r0 += PD_GGS_THREADS / 16) {
                const int row = r0 + (tid >> 4);
                if (row < 2 * n_slots && L.itab[row >> 1].y > 0) {
                    u64 *g = xs + (size_t)grow[row] * PD_XCHG_LINE + (tid & 15);
                    __hip_atomic_store(g, ((u64)epoch << 32) | (u64)__float_as_uint(own_rows[row * 16 + (tid & 15)]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // this workgroup's loss totals {sum s valid, n valid, sum min(s, max)} -> its totals line
            if (wave == PD_GGS_WAVES - 1) {
                float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
                for (int s = lane; s < n_slots; s += 64) {
                    if (L.itab[s].y > 0) {
                        t0 += L.item[s * PD_ITEM_VALS + 9];
                        t1 += L.item[s * PD_ITEM_VALS + 10];
                        t2 += L.item[s * PD_ITEM_VALS + 11];
                    }
                }
                t0 = wave_allsum(t0);
                t1 = wave_allsum(t1);
                t2 = wave_allsum(t2);
                if (lane < 4) {
                    const float v = lane == 0 ? t0 : (lane == 1 ? t1 : (lane == 2 ? t2 : 0.0f));
                    __hip_atomic_store(xs + (size_t)(n_inc + wg) * PD_XCHG_LINE + lane, ((u64)epoch << 32) | (u64)__float_as_uint(v),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            PD_PROF2H(q3);
            // ---- P3b: the owner of frame n sums that frame's rows in row order and publishes the frame line
            bool ok = true;
            for (int n = wg; n < N; n += k) {
                const int lo = L.incoff[n], cn = L.incoff[n + 1] - lo;   // <= 2 (N - 1) <= 126 rows
                ok = ggs2_gather<1>(xs + (size_t)lo * PD_XCHG_LINE, tid, cn * 8, 8, epoch, frame_rows, 16, P.err_flag) && ok;
                __syncthreads();
                if (tid < 64) {
                    // the frame's rows summed in a FIXED order that does not depend on the workgroup count: the four 16-lane rows of wave 0 each sum
                    // every fourth row (rows p, p + 4, ...: eight LDS reads in flight at a time), then (p0 + p1) + (p2 + p3) on the permlane swaps.
                    // History (tools/ggs_prof_n50.py, round 5): a plain loop over the rows was a chain of <= 63 dependent LDS round trips -- 5 700 of the
                    // 22 100 cycles of an iteration at 50 frames; eight reads in flight on 16 lanes: 3 300; this form: see profiles/round5_ggs_n50_phase_clocks.txt
                    const int c16 = tid & 15, part = tid >> 4;
                    float a = 0.0f;
                    for (int e0 = part; e0 < cn; e0 += 32) {              // (cn <= 2 (N - 1) rows; the loop bound differs between the four parts: no cross-lane operation inside)
                        float r[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) r[u] = e0 + 4 * u < cn ? frame_rows[(e0 + 4 * u) * 16 + c16] : 0.0f;
#pragma unroll
                        for (int u = 0; u < 8; ++u) a += r[u];
                    }
                    a = add_xor16(a);
                    a = add_xor32(a);
                    if (tid < 16)
                        __hip_atomic_store(xs + (size_t)(n_inc + k + n) * PD_XCHG_LINE + tid, ((u64)epoch << 32) | (u64)__float_as_uint(a),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
            PD_PROF2H(q4);
            // ---- hop 2: everybody gathers the N frame lines and the k totals lines
            ok = ggs2_gather<2>(xs + (size_t)(n_inc + k) * PD_XCHG_LINE, tid, N * 8, 8, epoch, L.psum, 16, P.err_flag) && ok;
            ok = ggs2_gather<1>(xs + (size_t)n_inc * PD_XCHG_LINE, tid, k * 2, 2, epoch, tot_rows, 4, P.err_flag) && ok;
            if (!ok) {
                atomicOr(P.err_flag, 1u);
                L.ctl[1] = 1.0f;
            }
            __syncthreads();
            if (L.ctl[1] != 0.0f) return;
            PD_PROF2H(q5);
            // per-frame gradients back through tc = D T and Rc[a][b] = D[a] R[b][a]; totals in workgroup order
            for (int q = tid; q < N * 16; q += PD_GGS_THREADS) {
                const int n = q >> 4, c = q & 15;
                const float v = L.psum[n * 16 + c];
                if (c < 9) {
                    const int aa = c / 3, bb = c % 3;
                    L.gR[n * 9 + bb * 3 + aa] = (aa < 2 ? -v : v);
                } else if (c < 12) {
                    L.gT[n * 3 + (c - 9)] = (c - 9 < 2 ? -v : v);
                } else {
                    L.gA[n * 4 + (c - 12)] = v;
                }
            }
            if (wave >= PD_GGS_WAVES - 3) {                 // one wave per total (round 5: one wave ran the 3 x ceil(k / 64) reductions back to back)
                const int c = wave - (PD_GGS_WAVES - 3);
                float t = 0.0f;
                for (int w0 = 0; w0 < k; w0 += 64) {      // fixed order: 64 workgroups at a time, tree inside
                    const int w = w0 + lane;
                    t += wave_allsum(w < k ? tot_rows[w * 4 + c] : 0.0f);
                }
                if (lane == 0) *(c == 0 ? &L.cam[6] : (c == 1 ? &L.cam[7] : &L.ctl[2])) = t;
            }
            __syncthreads();
            PD_PROF2H(q6);
#include "pd_ggs_p4.inc"
            __syncthreads();
            PD_PROF2H(q7);
            if (L.ctl[0] != 0.0f) break;
        }
        if (wave == 0 && lane == 0 && wg == 0 && P.stats) {
            float *so = P.stats + ((size_t)b * P.n_stages + st) * 4;
            so[0] = last_print;
            so[1] = (float)stepped;
            so[2] = last_cnt;
            so[3] = last_loss;
        }
        if (P.eval_only) break;
    }
    if (prof2 && lane == 0) {
        long long *o = P.prof + (wg == 0 ? 0 : 8);
        o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3; o[4] = q4; o[5] = q5; o[6] = q6; o[7] = q7;
    }
#undef PD_PROF2H
    if (own && wg == 0 && !P.eval_only) {
#pragma unroll
        for (int c = 0; c < 9; ++c) xg[lane * 9 + c] = L.xst[lane * PD_XS_STRIDE + c];
    }
}

#include "pd_ggs_lane.inc"
// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
void pd_ggs_free_seq(PdSeqHost &h) {
    if (h.blob) (void)hipFree(h.blob);
    h.blob = nullptr;
    h.blob_bytes = 0;
    memset(&h.desc, 0, sizeof(h.desc));
}

// (re)record the event a list keeps for stream `s`
int pd_record_stream_event(std::vector<pd_engine::StreamEvent> &list, hipStream_t s) {
    for (auto &e : list)
        if (e.stream == s) {
            PD_HIP_CHECK(hipEventRecord(e.event, s));
            return PD_OK;
        }
    hipEvent_t ev = nullptr;
    PD_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    list.push_back({s, ev});
    PD_HIP_CHECK(hipEventRecord(ev, s));
    return PD_OK;
}

// remember the point on `s` after which this engine's match tables are no longer read
int pd_mark_use(pd_engine *eng, hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return PD_OK;   // inside a graph capture: pd_sample_phase marks the replay instead
    return pd_record_stream_event(eng->uses, s);
}

// every enqueue that reads the match tables, on whatever stream it ran: waited for on the device (stream s) or on the host
int pd_wait_uses(pd_engine *eng, hipStream_t s, bool host) {
    for (auto &e : eng->uses) {
        if (host) PD_HIP_CHECK(hipEventSynchronize(e.event));
        else if (e.stream != s) PD_HIP_CHECK(hipStreamWaitEvent(s, e.event, 0));
    }
    return PD_OK;
}

// one slot's descriptor -> device (other slots may hold descriptors the ingestion kernels wrote on the device)
static int upload_seq_desc(pd_engine *eng, int seq) {
    PD_HIP_CHECK(hipMemcpy(eng->d_seqs + seq, &eng->seqs[seq].desc, sizeof(PdSeqDesc), hipMemcpyHostToDevice));
    return PD_OK;
}

int pd_wait_uploads(pd_engine *eng, hipStream_t s) {
    if (eng->uploads.empty()) return PD_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return PD_OK;   // pd_sample_phase waits before the replay
    for (auto &e : eng->uploads)
        if (e.stream != s) PD_HIP_CHECK(hipStreamWaitEvent(s, e.event, 0));      // (same stream: already ordered)
    return PD_OK;
}

extern "C" int pd_ggs_set_matches(pd_engine *eng, int seq, const double *kp1, const double *kp2, const int64_t *i12,
                                  int64_t M, int n_frames, int height, int width) {
    if (!eng || seq < 0 || seq >= eng->max_B) {
        pd_set_error("pd_ggs_set_matches: bad engine or sequence slot %d", seq);
        return PD_ERR_INVALID_ARG;
    }
    PD_HIP_CHECK(hipSetDevice(eng->device));
    // nothing of THIS engine in flight may still read the old tables; other engines (other batches of a pipeline) keep
    // running: no device-wide synchronisation here, and the blob is re-used when the new tables fit
    {
        int rc = pd_wait_uses(eng, nullptr, true);
        if (rc) return rc;
    }
    for (auto &e : eng->uploads) PD_HIP_CHECK(hipEventSynchronize(e.event));   // a pending device-side build of this slot
    eng->seqs[seq].device_built = false;
    if (M == 0) {
        pd_ggs_free_seq(eng->seqs[seq]);
        return upload_seq_desc(eng, seq);
    }
    if (!kp1 || !kp2 || !i12 || M < 0 || n_frames <= 0 || n_frames > PD_MAX_FRAMES || n_frames > eng->max_N ||
        height <= 0 || width <= 0) {
        pd_set_error("pd_ggs_set_matches: invalid arguments (M=%lld n_frames=%d h=%d w=%d; n_frames <= %d)",
                     (long long)M, n_frames, height, width, std::min(PD_MAX_FRAMES, eng->max_N));
        return PD_ERR_INVALID_ARG;
    }
    const int N = n_frames;
    // stable counting sort by pair key = i * N + j   (geometry_guided_sampling.py:26-27)
    std::vector<int> cnt((size_t)N * N + 1, 0);
    for (int64_t m = 0; m < M; ++m) {
        const int64_t a = i12[2 * m], c = i12[2 * m + 1];
        if (a < 0 || a >= N || c < 0 || c >= N) {
            pd_set_error("pd_ggs_set_matches: frame index (%lld,%lld) out of range [0,%d) at match %lld",
                         (long long)a, (long long)c, N, (long long)m);
            return PD_ERR_INVALID_ARG;
        }
        cnt[a * N + c + 1]++;
    }
    std::vector<int> key_off((size_t)N * N + 1, 0);
    for (int q = 0; q < N * N; ++q) key_off[q + 1] = key_off[q] + cnt[q + 1];
    std::vector<float4> pts((size_t)M);
    {
        std::vector<int> cur(key_off.begin(), key_off.end() - 1);
        for (int64_t m = 0; m < M; ++m) {
            const int key = (int)(i12[2 * m] * N + i12[2 * m + 1]);
            // .float() cast of geometry_guided_sampling.py:167 (round-to-nearest fp64 -> fp32)
            pts[cur[key]++] = make_float4((float)kp1[2 * m], (float)kp1[2 * m + 1], (float)kp2[2 * m], (float)kp2[2 * m + 1]);
        }
    }
    std::vector<int2> pair_ij;
    std::vector<int> pair_item_off;
    std::vector<int4> items;
    for (int q = 0; q < N * N; ++q) {
        const int m = key_off[q + 1] - key_off[q];
        if (m == 0) continue;
        const int p = (int)pair_ij.size();
        pair_ij.push_back(make_int2(q / N, q % N));
        pair_item_off.push_back((int)items.size());
        const int nch = (m + PD_ITEM_MAX_MATCHES - 1) / PD_ITEM_MAX_MATCHES;
        int start = key_off[q];
        for (int c = 0; c < nch; ++c) {
            const int len = m / nch + (c < m % nch ? 1 : 0);
            items.push_back(make_int4(p, start, len, 0));
            start += len;
        }
    }
    pair_item_off.push_back((int)items.size());
    const int n_pairs = (int)pair_ij.size(), n_items = (int)items.size();
    int max_item_len = 0;
    for (const int4 &it : items) max_item_len = std::max(max_item_len, it.z);
    for (int p = 0; p < n_pairs; ++p)
        if (pair_item_off[p + 1] - pair_item_off[p] > 0xffff) {
            pd_set_error("pd_ggs_set_matches: a frame pair holds too many matches");
            return PD_ERR_UNSUPPORTED;
        }
    // per-pair table: positions of the pair's two incidences (side 0 under frame i, side 1 under frame j) among the
    // incidences of its CHUNK of PD_GGS_THREADS pairs, sorted by frame; pchunk_off[chunk][n] = first position of frame n
    const int n_pchunks = (n_pairs + PD_GGS_THREADS - 1) / PD_GGS_THREADS;
    if (n_pchunks > PD_GGS_MAX_PCHUNKS) {
        pd_set_error("pd_ggs_set_matches: %d frame pairs with matches (max %d)", n_pairs, PD_GGS_MAX_PCHUNKS * PD_GGS_THREADS);
        return PD_ERR_UNSUPPORTED;
    }
    std::vector<int4> ptab(n_pairs);
    std::vector<int> pchunk_off((size_t)n_pchunks * (N + 1), 0);
    {
        std::vector<int> pos0(n_pairs, 0), pos1(n_pairs, 0);
        for (int ck = 0; ck < n_pchunks; ++ck) {
            const int p_lo = ck * PD_GGS_THREADS, p_hi = std::min(n_pairs, p_lo + PD_GGS_THREADS);
            int q = 0;
            for (int n = 0; n < N; ++n) {
                pchunk_off[(size_t)ck * (N + 1) + n] = q;
                for (int p = p_lo; p < p_hi; ++p) {
                    if (pair_ij[p].x == n) pos0[p] = q++;
                    if (pair_ij[p].y == n) pos1[p] = q++;
                }
            }
            pchunk_off[(size_t)ck * (N + 1) + N] = q;
        }
        for (int p = 0; p < n_pairs; ++p)
            ptab[p] = make_int4(pair_ij[p].x | (pair_ij[p].y << 8), pair_item_off[p], pair_item_off[p + 1] - pair_item_off[p],
                                pos0[p] | (pos1[p] << 16));
    }
    int max_deg = 0;                       // most pairs incident to one frame: the row stride of the fast per-frame sums (pd_ggs_kernel)
    {
        std::vector<int> deg(N, 0);
        for (int p = 0; p < n_pairs; ++p) {
            deg[pair_ij[p].x]++;
            deg[pair_ij[p].y]++;
        }
        for (int n = 0; n < N; ++n) max_deg = std::max(max_deg, deg[n]);
    }
    // the same positions among ALL incidences (two-hop kernel: one exchange line per (pair, side), grouped by frame)
    std::vector<int2> gpos(n_pairs);
    std::vector<int> ginc_off(N + 1, 0);
    int single_item_pairs = 1;
    {
        int q = 0;
        for (int n = 0; n < N; ++n) {
            ginc_off[n] = q;
            for (int p = 0; p < n_pairs; ++p) {
                if (pair_ij[p].x == n) gpos[p].x = q++;
                if (pair_ij[p].y == n) gpos[p].y = q++;
            }
        }
        ginc_off[N] = q;
        for (int p = 0; p < n_pairs; ++p)
            if (pair_item_off[p + 1] - pair_item_off[p] != 1) single_item_pairs = 0;
    }

    // lane-per-item tables (pd_ggs_lane_kernel): every pair is cut into ceil(m / len) lane items of balanced size, len = the smallest
    // length that leaves a sequence at most PD_LANE_MAX_ITEMS items (+ one more cut for the pairs with the longest items while lanes are
    // left, round 4); lane item q belongs to thread q, a wave's stream holds
    // max-over-its-lanes steps of two matches per lane (a lane past its item's end re-reads its last match, masked in the kernel)
    std::vector<int4> litems;
    std::vector<int2> lwave, lptab(n_pairs);
    std::vector<float4> lstream;
    int l_item_len = 0, l_max_steps = 0;
    if (n_pairs <= PD_LANE_MAX_ITEMS && n_pchunks == 1 && N <= PD_LANE_MAX_FRAMES) {
        int lo = 1, hi = 1;
        for (int q = 0; q < N * N; ++q) hi = std::max(hi, key_off[q + 1] - key_off[q]);
        auto count_items = [&](int len) {
            long long n = 0;
            for (int q = 0; q < N * N; ++q) n += pd_lane_items_of(key_off[q + 1] - key_off[q], len);
            return n;
        };
        while (lo < hi) {                                   // smallest len with <= PD_LANE_MAX_ITEMS items (n_pairs items at len = hi)
            const int mid = (lo + hi) / 2;
            if (count_items(mid) <= PD_LANE_MAX_ITEMS) hi = mid;
            else lo = mid + 1;
        }
        l_item_len = lo;
        // Cuts per pair at that length; the lanes this leaves over go, one more cut each, to the pairs whose items are longest (ties: the
        // earlier pair).  Then the items are ORDERED by length (steps of the pair's longest item, descending; pair; cut), 64 per wave: a
        // wave runs as many steps as its longest item, and waves w and w + 4 share a SIMD (tools/simd_probe.hip), so long and short waves
        // pair up.  A pair's items stay adjacent and in cut order (the pair backward sums them in that order).
        std::vector<int> l_m(n_pairs), l_nch(n_pairs), l_steps(n_pairs);
        int l_total = 0;
        for (int p = 0; p < n_pairs; ++p) {
            const int q = pair_ij[p].x * N + pair_ij[p].y;
            l_m[p] = key_off[q + 1] - key_off[q];
            l_nch[p] = pd_lane_items_of(l_m[p], l_item_len);
            l_total += l_nch[p];
        }
        const int spare = PD_LANE_MAX_ITEMS - l_total;
        {
            // round 6: k more cuts for the spare / k pairs with the longest items, k by the modelled match pass (pd_lane_pass_cost)
            std::vector<int> rank(n_pairs, 0), nch_k(n_pairs), steps_k(n_pairs), best_nch = l_nch;
            for (int p = 0; p < n_pairs; ++p)
                if (l_nch[p] != 0 && l_m[p] > l_nch[p]) rank[p] = pd_lane_rank(l_m.data(), l_nch.data(), n_pairs, p, false);
            int best_cost = 0x7fffffff;
            for (int k = 1; k <= PD_LANE_MORE_MAX; ++k)
                for (int d = 0; d < PD_LANE_MORE_SLACK && (d == 0 || spare / k - d > 0); ++d) {
                    int n_items = 0;
                    for (int p = 0; p < n_pairs; ++p) {
                        const bool elig = l_nch[p] != 0 && l_m[p] > l_nch[p] && rank[p] < spare / k - d;
                        nch_k[p] = l_nch[p] + (elig ? std::min(k, l_m[p] - l_nch[p]) : 0);
                        steps_k[p] = nch_k[p] ? (pd_lane_items_of(l_m[p], nch_k[p]) + 1) / 2 : 0;
                        n_items += nch_k[p];
                    }
                    int T[PD_LANE_WAVES] = {}, Tmin[PD_LANE_WAVES] = {};
                    for (int p = 0; p < n_pairs; ++p) {
                        if (!nch_k[p]) continue;
                        const int first = pd_lane_rank(steps_k.data(), nch_k.data(), n_pairs, p, true), end = first + nch_k[p];
                        for (int w = (first + 63) / 64; w < PD_LANE_WAVES && 64 * w < end; ++w) T[w] = steps_k[p];                       // item 64 w: the wave's longest
                        for (int w = first / 64; w < PD_LANE_WAVES && 64 * w < end; ++w)
                            if (std::min(64 * w + 63, n_items - 1) < end && std::min(64 * w + 63, n_items - 1) >= first) Tmin[w] = steps_k[p];   // its last item
                    }
                    const int cost = pd_lane_pass_cost(T, Tmin);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_nch = nch_k;
                    }
                }
            l_nch = best_nch;
        }
        for (int p = 0; p < n_pairs; ++p) l_steps[p] = l_nch[p] ? (pd_lane_items_of(l_m[p], l_nch[p]) + 1) / 2 : 0;
        litems.assign((size_t)PD_LANE_MAX_ITEMS, make_int4(0, 0, 0, 0));
        int n_lit = 0;
        for (int p = 0; p < n_pairs; ++p) {
            const int q = pair_ij[p].x * N + pair_ij[p].y, m = l_m[p], nch = l_nch[p];
            const int first = pd_lane_rank(l_steps.data(), l_nch.data(), n_pairs, p, true);
            lptab[p] = make_int2(first, nch);
            int start = key_off[q];
            for (int c = 0; c < nch; ++c) {
                const int len = m / nch + (c < m % nch ? 1 : 0);
                litems[(size_t)first + c] = make_int4(pair_ij[p].x | (pair_ij[p].y << 8), len, p, start);
                start += len;
            }
            n_lit += nch;
        }
        litems.resize(n_lit);
        const int n_lw = ((int)litems.size() + 63) / 64;
        size_t base = 0;
        for (int w = 0; w < n_lw; ++w) {
            int steps = 0;
            for (int l = 0; l < 64 && w * 64 + l < (int)litems.size(); ++l) steps = std::max(steps, (litems[w * 64 + l].y + 1) / 2);
            lwave.push_back(make_int2((int)base, steps));
            l_max_steps = std::max(l_max_steps, steps);
            base += (size_t)steps * 128;
        }
        lstream.assign(base, make_float4(1.0f, 1.0f, 1.0f, 1.0f));
        for (int w = 0; w < n_lw; ++w)
            for (int t = 0; t < lwave[w].y; ++t)
                for (int l = 0; l < 64 && w * 64 + l < (int)litems.size(); ++l) {
                    const int4 it = litems[w * 64 + l];
                    const float4 a = pts[(size_t)it.w + std::min(2 * t, it.y - 1)], b2 = pts[(size_t)it.w + std::min(2 * t + 1, it.y - 1)];
                    float4 q0, q1;
                    pd_interleave_pair(a, b2, q0, q1);
                    lstream[(size_t)lwave[w].x + (size_t)(2 * t) * 64 + l] = q0;
                    lstream[(size_t)lwave[w].x + (size_t)(2 * t + 1) * 64 + l] = q1;
                }
    }

    // one blob: pts | pair_ij | pair_item_off | items | ptab | pchunk_off   (aligned pieces)
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_pts = 0;
    const size_t o_pij = al(o_pts + sizeof(float4) * pts.size());
    const size_t o_pio = al(o_pij + sizeof(int2) * pair_ij.size());
    const size_t o_itm = al(o_pio + sizeof(int) * pair_item_off.size());
    const size_t o_ptb = al(o_itm + sizeof(int4) * items.size());
    const size_t o_pco = al(o_ptb + sizeof(int4) * ptab.size());
    const size_t o_gps = al(o_pco + sizeof(int) * pchunk_off.size());
    const size_t o_gio = al(o_gps + sizeof(int2) * gpos.size());
    const size_t o_lit = al(o_gio + sizeof(int) * ginc_off.size());
    const size_t o_lwv = al(o_lit + sizeof(int4) * litems.size());
    const size_t o_lpt = al(o_lwv + sizeof(int2) * lwave.size());
    const size_t o_lst = al(o_lpt + sizeof(int2) * lptab.size());
    const size_t total = al(o_lst + sizeof(float4) * lstream.size());
    std::vector<char> host(total, 0);
    for (const int4 &it : items)                                  // full 128-match groups of every item: pair-interleaved (MatchRegs)
        for (int g = 0; g + 128 <= it.z; g += 128)
            for (int l = 0; l < 64; ++l) {
                float4 &a = pts[(size_t)it.y + g + l], &b = pts[(size_t)it.y + g + 64 + l];
                float4 q0, q1;
                pd_interleave_pair(a, b, q0, q1);
                a = q0;
                b = q1;
            }
    memcpy(host.data() + o_pts, pts.data(), sizeof(float4) * pts.size());
    memcpy(host.data() + o_pij, pair_ij.data(), sizeof(int2) * pair_ij.size());
    memcpy(host.data() + o_pio, pair_item_off.data(), sizeof(int) * pair_item_off.size());
    memcpy(host.data() + o_itm, items.data(), sizeof(int4) * items.size());
    memcpy(host.data() + o_ptb, ptab.data(), sizeof(int4) * ptab.size());
    memcpy(host.data() + o_pco, pchunk_off.data(), sizeof(int) * pchunk_off.size());
    memcpy(host.data() + o_gps, gpos.data(), sizeof(int2) * gpos.size());
    memcpy(host.data() + o_gio, ginc_off.data(), sizeof(int) * ginc_off.size());
    if (!litems.empty()) {
        memcpy(host.data() + o_lit, litems.data(), sizeof(int4) * litems.size());
        memcpy(host.data() + o_lwv, lwave.data(), sizeof(int2) * lwave.size());
        memcpy(host.data() + o_lpt, lptab.data(), sizeof(int2) * lptab.size());
        memcpy(host.data() + o_lst, lstream.data(), sizeof(float4) * lstream.size());
    }
    PdSeqHost &h = eng->seqs[seq];
    if (h.blob_bytes < total) {
        pd_ggs_free_seq(h);
        PD_HIP_CHECK(hipMalloc(&h.blob, total));
        h.blob_bytes = total;
    }
    memset(&h.desc, 0, sizeof(h.desc));
    PD_HIP_CHECK(hipMemcpy(h.blob, host.data(), total, hipMemcpyHostToDevice));
    char *base = (char *)h.blob;
    h.desc.pts = (const float4 *)(base + o_pts);
    h.desc.pair_ij = (const int2 *)(base + o_pij);
    h.desc.pair_item_off = (const int *)(base + o_pio);
    h.desc.items = (const int4 *)(base + o_itm);
    h.desc.ptab = (const int4 *)(base + o_ptb);
    h.desc.pchunk_off = (const int *)(base + o_pco);
    h.desc.n_pchunks = n_pchunks;
    h.desc.gpos = (const int2 *)(base + o_gps);
    h.desc.ginc_off = (const int *)(base + o_gio);
    h.desc.single_item_pairs = single_item_pairs;
    h.desc.lstream = (const float4 *)(base + o_lst);
    h.desc.litems = (const int4 *)(base + o_lit);
    h.desc.lwave = (const int2 *)(base + o_lwv);
    h.desc.lptab = (const int2 *)(base + o_lpt);
    h.desc.n_litems = (int)litems.size();
    h.desc.n_lwaves = (int)lwave.size();
    h.desc.l_item_len = l_item_len;
    h.desc.l_max_steps = l_max_steps;
    h.desc.M = (int)M;
    h.desc.n_pairs = n_pairs;
    h.desc.n_items = n_items;
    h.desc.n_frames = N;
    h.desc.sc = (float)std::min(height, width) / 2.0f;   // opencv_from_cameras_projection scale
    h.desc.cx = (float)width / 2.0f;
    h.desc.cy = (float)height / 2.0f;
    h.max_item_len = max_item_len;
    h.max_deg = max_deg;
    return upload_seq_desc(eng, seq);
}

// Zeroes the exchange granules before every launch (tags restart at 1 per launch).  A KERNEL rather than
// hipMemsetAsync: under hipGraph replay with a second graph running concurrently, the memset NODE was observed
// not to be ordered against the neighbouring kernel nodes (stale tags of the previous launch were accepted ->
// silently wrong sums; tests/test_gpu_parity.py::test_two_engines_overlapped...); kernel -> kernel edges are.
__global__ void pd_ggs_zero_kernel(unsigned long long *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0ull;
}

int pd_ggs_init() {
    const void *variants[] = {(const void *)pd_ggs_kernel<0, true>, (const void *)pd_ggs_kernel<0, false>, (const void *)pd_ggs_kernel<3, false>,
                              (const void *)pd_ggs_kernel<5, false>, (const void *)pd_ggs_kernel<6, false>,
                              (const void *)pd_ggs_kernel<3, false, 12>, (const void *)pd_ggs_kernel<5, false, 12>, (const void *)pd_ggs_kernel<6, false, 12>};
    for (const void *f : variants) PD_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PD_HIP_CHECK(hipFuncSetAttribute((const void *)pd_ggs2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PD_HIP_CHECK(hipFuncSetAttribute((const void *)pd_ggs_lane_kernel<PD_LANE_RV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PD_OK;
}

// The launch shape of one GGS launch, derived from the uploaded match tables: workgroups per sequence, local item
// slots, dynamic LDS and which kernel.  Captured hipGraphs bake these in, so pd_sample_phase keys its graph cache on
// the plan (a re-upload with another item count must never replay the old shape: the kernel would index its LDS
// tables past their size).  Also validates what pd_ggs_launch validates, so a graph replay cannot skip the checks.
int pd_ggs_plan(pd_engine *eng, int B, int N, const pd_ggs_cfg *cfg, PdGgsPlan *out) {
    if (!eng || !cfg || !out || B <= 0 || B > eng->max_B || N <= 0 || N > eng->max_N || N > PD_MAX_FRAMES) {
        pd_set_error("pd_ggs: invalid arguments (B=%d N=%d)", B, N);
        return PD_ERR_INVALID_ARG;
    }
    int max_items = 0;
    for (int b = 0; b < B; ++b) {
        const PdSeqDesc &d = eng->seqs[b].desc;
        if (d.M <= 0 || !eng->seqs[b].blob) {
            pd_set_error("pd_ggs: sequence slot %d has no matches (call pd_ggs_set_matches)", b);
            return PD_ERR_STATE;
        }
        if (d.n_frames != N) {
            pd_set_error("pd_ggs: slot %d matches were uploaded for %d frames, called with N=%d", b, d.n_frames, N);
            return PD_ERR_INVALID_ARG;
        }
        max_items = std::max(max_items, d.n_items);
    }
    // workgroups per sequence: one item per wave if the chip has room (<= 256 resident workgroups)
    int device_cus = eng->num_cus > 0 ? eng->num_cus : 256;
    int k = cfg->wgs_per_seq > 0 ? cfg->wgs_per_seq : (max_items + PD_GGS_WAVES - 1) / PD_GGS_WAVES;
    k = std::max(1, std::min(k, device_cus / B));
    // the lane-per-item kernel: on request (PD_GGS_CFG_LANE_ITEMS: what the pipeline sets wherever it gives a sequence ONE workgroup), or when
    // the engine picks the shape and the launch holds more sequences than half the CUs (nothing to gain from several workgroups per sequence).
    // An explicit wgs_per_seq without the flag keeps the wave-per-item kernels, whose results are bitwise independent of the workgroup
    // count (the lane kernel sums in another fixed order: rounding-level differences).  Round 4: its stream goes through an
    // LDS ring fed by LDS-DMA that never stops (pd_ggs_lane.inc) and it is 10 - 14 % faster than the 12-wave wave-per-item kernel at
    // the bench shape (18.6 against 21.6 ms per 256-sequence launch, profiles/round4_lane_ring.txt); fully resident sequences 1.5 - 2 x.
    memset(out, 0, sizeof(*out));
    if (!(cfg->reserved & PD_GGS_CFG_NO_LANE_ITEMS) &&
        ((cfg->reserved & PD_GGS_CFG_LANE_ITEMS) || (cfg->wgs_per_seq == 0 && device_cus / B <= 1))) {
        bool ok = N <= PD_GGS_FAST_FRAMES;
        int pairs = 0, steps = 0, deg = 0;
        for (int b = 0; b < B && ok; ++b) {
            const PdSeqDesc &d = eng->seqs[b].desc;
            ok = d.n_litems > 0 && d.n_pchunks == 1;
            pairs = std::max(pairs, d.n_pairs);
            steps = std::max(steps, d.l_max_steps);
            deg = std::max(deg, eng->seqs[b].max_deg);
        }
        if (ok) {
            // rows of the pair backward at the fixed per-frame stride of the fast serial phases (pd_ggs_p3b.inc)
            const int pinc_rows = std::max(2 * std::min(PD_LANE_MAX_ITEMS, std::max(pairs, 1)), N * (((deg + 3) & ~3) + 1));
            const size_t lds = lane_lds_bytes(pinc_rows);        // tables + the waves' rings (PD_LANE_RING steps of 2 KiB each)
            if (lds <= 160 * 1024) {
                out->lane = 1;
                out->lane_rl = std::max(0, std::min(PD_LANE_RL, steps - PD_LANE_RV));     // (reported: steps of the longest wave that live in LDS for the launch, beside the ring)
                out->k = 1;
                out->waves = PD_LANE_WAVES;
                out->pinc_rows = pinc_rows;
                out->lds = (int)lds;
                out->max_items = PD_LANE_MAX_ITEMS;
                return PD_OK;
            }
        }
    }
    // k > 1: every workgroup publishes one exchange line per work item in the sequence's [epoch][item] region of d_xchg
    // (pd_ggs_kernel, `xchg + epoch * xchg_stride + item * PD_XCHG_LINE`): a sequence with more items than the region holds
    // (2 max_N^2 + 512 lines, pd_engine_create) would write into the other epoch's lines or the next slot's -- one workgroup
    // per sequence then (no exchange at all).  The two-hop kernel checks its own, smaller line count below.
    if (k > 1 && (size_t)max_items * PD_XCHG_LINE > eng->xchg_granules) k = 1;
    // many frames (several chunks of pairs): the two-hop kernel distributes the backward over the workgroups instead of
    // replicating it -- needs one work item per pair and room for its exchange lines; it keeps only a workgroup's own
    // item sums in LDS, so it also covers item counts whose full table would not fit
    bool two_hop = k > 1 && !(cfg->reserved & PD_GGS_CFG_FORCE_ONE_HOP);
    for (int b = 0; b < B && two_hop; ++b) {
        const PdSeqDesc &d = eng->seqs[b].desc;
        two_hop = d.n_pchunks > 1 && d.single_item_pairs;
    }
    // one-hop kernel: backward rows of one chunk of pairs (both sides); LDS staging of the match pass when every item fits a
    // staging buffer (<= 384 matches) -- both only shrink / extend the LDS image, the arithmetic is the same
    int max_pairs = 0, max_len = 0;
    for (int b = 0; b < B; ++b) {
        max_pairs = std::max(max_pairs, eng->seqs[b].desc.n_pairs);
        max_len = std::max(max_len, eng->seqs[b].max_item_len);
    }
    int pinc_one_hop = std::min(PD_GGS_PINC_ROWS, 2 * std::min(PD_GGS_THREADS, std::max(max_pairs, 1)));
    // fast serial phases (<= PD_GGS_FAST_FRAMES frames, one chunk of pairs): rows at a fixed stride per frame -- N x (largest degree,
    // rounded up to 4, + 1) rows; about 2 x pairs for the complete graph of hloc's exhaustive pairs.  The kernel decides from the actual
    // tables; here only the room is made (dropped again below if the LDS image would not fit).
    int pinc_fast = 0;
    if (N <= PD_GGS_FAST_FRAMES) {
        int deg = 0;
        bool one_chunk = true;
        for (int b = 0; b < B; ++b) {
            deg = std::max(deg, eng->seqs[b].max_deg);
            one_chunk = one_chunk && eng->seqs[b].desc.n_pchunks <= 1;
        }
        if (one_chunk && deg > 0) pinc_fast = N * (((deg + 3) & ~3) + 1);     // (+ 1: the bank-conflict-free row stride, pd_ggs_kernel)
    }
    const int pinc_general = pinc_one_hop;
    if (pinc_fast <= PD_GGS_PINC_ROWS) pinc_one_hop = std::max(pinc_one_hop, pinc_fast);
    int stage_want = 0;
    if (!(cfg->reserved & PD_GGS_CFG_NO_LDS_STAGING) && max_len > 0) {
        const int pieces = (max_len + 63) / 64;
        stage_want = pieces <= 3 ? 3 : pieces <= 5 ? 5 : pieces <= 6 ? 6 : 0;
    }
    int n_slots = 0, pinc_rows = PD_GGS_PINC_ROWS, stage_p = 0;
    size_t lds = 0;
    // LDS image of a candidate shape; three waves per SIMD (12 waves, ONE staging buffer per wave) for the staged match pass at one
    // workgroup per sequence with several rounds of items per wave (the bench shape; A/B switch PD_GGS_CFG_WAVES8 keeps 8 waves)
    int waves = PD_GGS_WAVES;
    auto image = [&](int rows, int sp, int kk_, int slots, int &w_out) -> size_t {
        w_out = PD_GGS_WAVES;
        const size_t l8 = ggs_lds_bytes(slots, two_hop ? slots : max_items, rows, sp, PD_GGS_WAVES * 2);
        if (!two_hop && sp > 0 && kk_ == 1 && slots >= 3 * 12 && !(cfg->reserved & PD_GGS_CFG_WAVES8) && (PD_GGS_PROF12 || !eng->ggs_prof_on)) {
            const size_t l12 = ggs_lds_bytes(slots, max_items, rows, sp, 12);
            if (l12 <= 160 * 1024) {
                w_out = 12;
                return l12;
            }
        }
        return l8;
    };
    for (int pass = 0; pass < 2; ++pass) {
        int kk = k;
        for (;;) {
            const int rounds = (max_items + kk * PD_GGS_WAVES - 1) / (kk * PD_GGS_WAVES);
            n_slots = rounds * PD_GGS_WAVES;
            pinc_rows = two_hop ? PD_GGS_PINC_ROWS : pinc_one_hop;
            stage_p = (!two_hop && rounds > 1) ? stage_want : 0;     // one item per wave: matches are register resident
            lds = image(pinc_rows, stage_p, kk, n_slots, waves);
            if (lds > 160 * 1024 && !two_hop && pinc_rows > pinc_general) {   // the wider rows of the fast serial phases are optional
                pinc_one_hop = pinc_rows = pinc_general;
                lds = image(pinc_rows, stage_p, kk, n_slots, waves);
            }
            if (lds > 160 * 1024 && stage_p > 0) {                   // staging is optional: without it first
                stage_p = 0;
                lds = image(pinc_rows, 0, kk, n_slots, waves);
            }
            if (lds <= 160 * 1024 || kk >= device_cus / B) break;
            ++kk;
        }
        if (two_hop) {
            bool fits = lds <= 160 * 1024 && 2 * n_slots <= 512 && kk <= 256;
            for (int b = 0; b < B && fits; ++b)
                fits = (size_t)(2 * eng->seqs[b].desc.n_pairs + kk + N) * PD_XCHG_LINE <= eng->xchg_granules;
            if (!fits) {
                two_hop = false;   // size again for the single-exchange kernel
                continue;
            }
        }
        k = kk;
        break;
    }
    if (lds > 160 * 1024) {
        pd_set_error("pd_ggs: %d work items need %zu B of LDS per workgroup (> 160 KiB) at B=%d", max_items, lds, B);
        return PD_ERR_UNSUPPORTED;
    }
    if (k > 1 && !two_hop && (size_t)max_items * PD_XCHG_LINE > eng->xchg_granules) {   // (k grew again because one workgroup's LDS image did not fit)
        pd_set_error("pd_ggs: %d work items per sequence exceed the exchange region (%zu lines) and do not fit one workgroup's LDS at B=%d",
                     max_items, eng->xchg_granules / PD_XCHG_LINE, B);
        return PD_ERR_UNSUPPORTED;
    }
    // XCD-local placement of the exchange (k > 1, one-hop kernel): block b of a launch runs on XCD b % 8, so with the block -> (sequence,
    // workgroup) mapping padded to a multiple of 8 sequences all workgroups of a sequence share an XCD -- if its 32 CUs can hold them all at
    // once (they spin on each other: co-residency) and the handshake granules fit behind the items' lines.  PD_GGS_CFG_XCHG_SPREAD: A / B.
    const int seq_per_xcd = (B + 7) / 8;
    out->xchg_local = (k > 1 && !two_hop && device_cus == 256 && seq_per_xcd * k <= 32 && k <= 256 &&
                       (size_t)max_items * PD_XCHG_LINE + 256 <= eng->xchg_granules && !(cfg->reserved & PD_GGS_CFG_XCHG_SPREAD)) ? 1 : 0;
    out->waves = waves;
    out->pinc_rows = pinc_rows;
    out->stage_p = stage_p;
    out->k = k;
    out->n_slots = n_slots;
    out->lds = (int)lds;
    out->two_hop = two_hop ? 1 : 0;
    out->max_items = max_items;
    return PD_OK;
}

int pd_ggs_launch(pd_engine *eng, float *x, int B, int N, const PdGgsStage *stages, int n_stages,
                  const pd_ggs_cfg *cfg, int eval_only, float *stats, float *trace, int trace_iters,
                  float *loss_out, float *grad_out, hipStream_t s) {
    if (!eng || !x || !cfg || n_stages <= 0 || n_stages > PD_GGS_MAX_STAGES) {
        pd_set_error("pd_ggs: invalid arguments (B=%d N=%d stages=%d)", B, N, n_stages);
        return PD_ERR_INVALID_ARG;
    }
    PdGgsPlan plan;
    int prc = pd_ggs_plan(eng, B, N, cfg, &plan);
    if (prc) return prc;
    if ((prc = pd_wait_uploads(eng, s))) return prc;
    const int k = plan.k, n_slots = plan.n_slots;
    const size_t lds = (size_t)plan.lds;
    const bool two_hop = plan.two_hop != 0;
    const int pinc_rows = plan.pinc_rows, items_cap = plan.max_items;
    PdGgsParams P;
    memset(&P, 0, sizeof(P));
    P.seqs = eng->d_seqs;
    P.x = x;
    P.N = N;
    P.k = k;
    for (int i = 0; i < n_stages; ++i) P.stages[i] = stages[i];
    P.n_stages = n_stages;
    P.alpha = cfg->alpha;
    P.lr = cfg->learning_rate;
    P.sampson_max = cfg->sampson_max;
    P.momentum = cfg->momentum;
    P.min_matches = cfg->min_matches;
    P.eval_only = eval_only;
    P.stats = stats;
    P.trace = trace;
    P.trace_iters = trace_iters;
    P.loss_out = loss_out;
    P.grad_out = grad_out;
    P.xchg = (k > 1) ? eng->d_xchg : nullptr;
    P.xchg_stride = (int)eng->xchg_granules;
    P.err_flag = eng->d_err;
    P.prof = eng->ggs_prof_on ? (long long *)(eng->d_err + 2) : nullptr;
    P.prof_wave = eng->ggs_prof_on > 0 ? (eng->ggs_prof_on - 1) & 0x107 : 1;
    P.n_seqs = B;
    P.stamp = eng->d_stamps ? eng->d_stamps + 2 * (size_t)eng->stamp_slot : nullptr;
    P.xchg_local = plan.xchg_local;
    const int B_map = plan.xchg_local ? ((B + 7) & ~7) : B;     // the kernels' block -> (sequence, workgroup) mapping
    if (k > 1) {
        // tags restart at 1 every launch: zero every polled word first (guide G16 "re-initialise every call")
        const size_t n_zero = 2 * eng->xchg_granules * B;
        hipLaunchKernelGGL(pd_ggs_zero_kernel, dim3(256), dim3(256), 0, s, eng->d_xchg, n_zero);
    }
    if (plan.lane)
        hipLaunchKernelGGL(pd_ggs_lane_kernel<PD_LANE_RV>, dim3(B), dim3(PD_LANE_THREADS), lds, s, P, pinc_rows);
    else if (two_hop)
        hipLaunchKernelGGL(pd_ggs2_kernel, dim3(B * k), dim3(PD_GGS_THREADS), lds, s, P, B, n_slots);
    else {
        void (*kern)(PdGgsParams, int, int, int, int) =
            n_slots == PD_GGS_WAVES ? pd_ggs_kernel<0, true>
            : plan.waves == 12 && plan.stage_p == 6 ? pd_ggs_kernel<6, false, 12>
            : plan.waves == 12 && plan.stage_p == 5 ? pd_ggs_kernel<5, false, 12>
            : plan.waves == 12 && plan.stage_p == 3 ? pd_ggs_kernel<3, false, 12>
            : plan.stage_p == 6 ? pd_ggs_kernel<6, false>
            : plan.stage_p == 5 ? pd_ggs_kernel<5, false>
            : plan.stage_p == 3 ? pd_ggs_kernel<3, false>
                                : pd_ggs_kernel<0, false>;
        hipLaunchKernelGGL(kern, dim3(B_map * k), dim3(plan.waves * 64), lds, s, P, B_map, n_slots, pinc_rows, items_cap);
    }
    PD_HIP_CHECK(hipGetLastError());
    return pd_mark_use(eng, s);
}
