// pd_internal.h -- engine-private declarations shared by the HIP translation units.
// gfx950 / CDNA4 only: 64-wide wavefronts, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/pd_engine.h"

#define PD_WAVE 64
#define PD_MAX_FRAMES 64          // one wavefront lane per frame in the GGS update phase
#define PD_GGS_THREADS 512        // 8 waves per GGS workgroup
#define PD_GGS_WAVES (PD_GGS_THREADS / PD_WAVE)
#define PD_GGS_PINC_ROWS (2 * PD_GGS_THREADS)   // pair backward results in LDS: both sides of one chunk of PD_GGS_THREADS pairs
#define PD_GGS_MAX_PCHUNKS 8                   // 64 frames, both orders of every pair: 4032 pairs -> 8 chunks
#define PD_GGS_MAX_STAGES 5
#define PD_ITEM_MAX_MATCHES 512   // one work item = <= 512 matches of one frame pair (8 per lane)
#define PD_ITEM_VALS 12           // 9 dL/dF sums + sum(s valid) + n_valid + sum(min(s, max))
// lane-per-item kernel (pd_ggs_lane_kernel, the throughput shape: one workgroup of 8 waves per sequence with up to 256 VGPRs each -- most
// of them hold matches for the whole launch --, every LANE owns a work item)
#ifndef PD_LANE_WAVES
#define PD_LANE_WAVES 8
#endif
#define PD_LANE_THREADS (PD_LANE_WAVES * PD_WAVE)
#define PD_LANE_MAX_ITEMS PD_LANE_THREADS   // one lane item per thread
#define PD_LANE_MAX_FRAMES 24               // 16 threads per frame in the per-frame sums (= PD_GGS_FAST_FRAMES: the fast serial phases)
#define PD_LANE_ITEM_VALS 10                // 9 dL/dF sums + sum(s valid) per lane item
#define PD_GGS_FAST_FRAMES 24               // one-hop kernel: up to this many frames the per-frame sums take 16 lanes per frame (6 waves) and two
                                            //   idle waves form the totals beside them (pd_ggs_kernel, "fast serial phases")

void pd_set_error(const char *fmt, ...);

// Development A / B switches (environment variables read once per process: PD_DEN_STRIP, PD_DEN_ATTN_MMA, PD_SMALL_DBG,
// PD_LANE_VARIANT, PD_LANE_LDS_SPARE_KB).  They change which kernels a captured hipGraph bakes in and are not part of the
// graph key, so the product build compiles them OUT (every knob is its default, a constant); `make EXTRA=-DPD_DEV_KNOBS`
// brings them back for the probes under tools/.
#ifdef PD_DEV_KNOBS
#include <stdlib.h>
static inline int pd_dev_knob(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
#define pd_dev_knob(name, dflt) (dflt)
#endif

// ReLU as torch computes it (clamp_min semantics: NaN propagates).  fmaxf / v_max_f32 return the OTHER operand for a NaN, which turned
// a network full of NaN (non-finite checkpoint) into finite garbage -- the `_last` MLP's ReLU zeroed them and the output was its bias.
__device__ __forceinline__ float pd_relu(float v) { return v < 0.0f ? 0.0f : v; }

#define PD_HIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            pd_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PD_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

// ---- GGS match container (device view) ---------------------------------------------------------
// Matches of one sequence, sorted by frame pair so that a wavefront owns one pair at a time
// (geometry_guided_sampling.py:26-27 builds pair_idx = i*N + j; hloc already groups by pair).
struct PdSeqDesc {
    const float4 *pts;         // [M] (u1, v1, u2, v2) fp32, pair-sorted
    const int2 *pair_ij;       // [n_pairs] (i, j) frame indices, p2^T F p1 = 0 with 1 = i, 2 = j
    const int *pair_item_off;  // [n_pairs + 1] items of pair p are items[off[p] .. off[p+1])
    const int4 *items;         // [n_items] (pair, first match, match count, 0)
    const int4 *ptab;          // [n_pairs] (i | j << 8, first item, n_items, pos_side0 | pos_side1 << 16): rows of the pair's two
                               //   results (side 0 -> frame i, side 1 -> frame j) among its chunk's frame-sorted incidences
    const int *pchunk_off;     // [n_pchunks][n_frames + 1] per chunk of PD_GGS_THREADS pairs: CSR of the chunk's incidences by frame
    int n_pchunks;             // (ptab positions are chunk-local)
    const int2 *gpos;          // [n_pairs] rows of the pair's (side 0, side 1) results among ALL incidences, frame-sorted
    const int *ginc_off;       // [n_frames + 1] CSR of those rows by frame (two-hop kernel for many frames)
    int single_item_pairs;     // 1: every pair is one work item (<= PD_ITEM_MAX_MATCHES matches)
    // lane-per-item tables (pd_ggs_lane_kernel; n_litems == 0: not built -- more than PD_LANE_MAX_ITEMS pairs or several chunks of pairs)
    const float4 *lstream;     // lane-major stream: wave w, step t, half h, lane l at [lwave[w].x + (2 t + h) * 64 + l]; half 0 =
                               //   (u1_A, u1_B, v1_A, v1_B), half 1 = (u2_A, u2_B, v2_A, v2_B) for A, B = matches 2 t, 2 t + 1 of lane l's item
    const int4 *litems;        // [n_litems] (i | j << 8, match count, pair, first match): lane item q belongs to thread q
    const int2 *lwave;         // [n_lwaves] (first float4 of the wave's stream, steps = max over its lanes of ceil(count / 2))
    const int2 *lptab;         // [n_pairs] (first lane item, lane items) of the pair
    int n_litems, n_lwaves, l_item_len, l_max_steps;
    int M, n_pairs, n_items, n_frames;
    float sc, cx, cy;          // min(h, w) / 2, w / 2, h / 2 (opencv_from_cameras_projection)
    int pad;
};

struct PdGgsStage {
    int update_R, update_T, update_FL, iters;
};

struct PdGgsParams {
    const PdSeqDesc *seqs;     // [B]
    float *x;                  // [B, N, 9] in/out
    int N, k;                  // frames, workgroups per sequence
    PdGgsStage stages[PD_GGS_MAX_STAGES];
    int n_stages;
    float alpha, lr, sampson_max, momentum;
    int min_matches;
    int eval_only;             // 1: single forward/backward, no update (pd_ggs_loss_grad)
    float *stats;              // [B, n_stages, 4] or null
    float *trace;              // [B, trace_iters, N*9 + 3] or null
    int trace_iters;
    float *loss_out;           // [B, 4]   (eval_only)
    float *grad_out;           // [B, N, 9] (eval_only)
    unsigned long long *xchg;  // [B, 2, max_items * 12] tagged granules (k > 1)
    int xchg_stride;           // granules per (sequence, slot)
    unsigned int *err_flag;    // device word: nonzero = a bounded spin gave up
    long long *prof;           // optional phase cycle counters (debug), else null
    int prof_wave;             // which wave of workgroup 0 records them
    int n_seqs;                // sequences of the launch (the kernels' `B` argument is this, or this rounded up to 8: xchg_local)
    unsigned long long *stamp; // [2] {start, end} of this launch in wall_clock64() ticks (constant rate: hipDeviceAttributeWallClockRate), written by the
                               //   lane-per-item kernel itself: workgroup 0 stores its start, every workgroup atomicMax'es its end (the clock only
                               //   grows, so a slot re-used by the next replay needs no reset); null: not recorded.  bench.py reads the launches of
                               //   its timed region from these (pd_ggs_launch_stamps): the in-pipe duration of the dominant kernel
    int xchg_local;            // one-hop kernel, k > 1: the launch places all workgroups of a sequence on ONE XCD (block -> sequence mapping
                               //   padded to a multiple of 8); the kernel verifies it and then keeps its exchange stores in that XCD's L2
};

// launch shape of one GGS launch (pd_ggs_plan): everything a captured graph node bakes in besides its arguments
// Match table layout inside a work item: every FULL group of 128 matches is stored pair-interleaved (pd_ggs.hip MatchRegs) -- this is the
// transform of one lane's two matches A = group[lane], B = group[64 + lane], applied by both table builders (host: pd_ggs_set_matches;
// device: ingest_interleave_kernel)
__host__ __device__ inline void pd_interleave_pair(const float4 a, const float4 b, float4 &q0, float4 &q1) {
    q0 = make_float4(a.x, b.x, a.y, b.y);
    q1 = make_float4(a.z, b.z, a.w, b.w);
}

struct PdGgsPlan {
    int k, n_slots, lds, two_hop, max_items;
    int pinc_rows, stage_p;    // one-hop kernel: LDS rows of the pair backward; LDS-DMA staging pieces per item (0 = through registers)
    int waves;                 // one-hop kernel: waves per workgroup (8, or 12 for the staged k = 1 shape)
    int lane, lane_rl;         // lane-per-item kernel chosen; steps RV .. RV + lane_rl - 1 of its longest wave's stream live in LDS for the whole
                               //   launch (<= PD_LANE_RL; the PD_LANE_RING slots of the ring come on top)
    int xchg_local;            // one-hop kernel: XCD-local placement of a sequence's workgroups (see PdGgsParams)
};

// lane items of one frame pair with m matches at lane-item length len: ceil(m / len) items of balanced size (host and device builders)
__host__ __device__ inline int pd_lane_items_of(int m, int len) { return (m + len - 1) / len; }
// Ordering of the frame pairs of a sequence by item length (host and device builders of the lane tables; n <= 576 keys, quadratic on purpose:
// the same few lines on both sides).  val[q] = matches (by_steps false: compared through the item length ceil(val / nch)) or steps of
// the pair's longest item (by_steps true); nch[q] = its cuts, 0 = no such pair.  Returns, for pair p, by_steps false: the NUMBER OF PAIRS
// ordered before p (longer items first, ties: lower index); by_steps true: the number of ITEMS of the pairs ordered before p.
__host__ __device__ inline int pd_lane_rank(const int *val, const int *nch, int n, int p, bool by_steps) {
    const int vp = by_steps ? val[p] : pd_lane_items_of(val[p], nch[p]);
    int r = 0;
    for (int q = 0; q < n; ++q) {
        if (nch[q] == 0 || q == p) continue;
        const int vq = by_steps ? val[q] : pd_lane_items_of(val[q], nch[q]);
        if (vq > vp || (vq == vp && q < p)) r += by_steps ? nch[q] : 1;
    }
    return r;
}

// Which cut rule the spare lanes get (host and device builders): candidate (k, d), k = 1 .. PD_LANE_MORE_MAX, d = 0 .. PD_LANE_MORE_SLACK - 1,
// gives k MORE cuts to each of the spare / k - d pairs with the longest items; the candidate with the cheapest MODELLED match pass wins
// (ties: the smaller k, then the smaller d).  The model is what the phase clocks of the 8-wave kernel show (profiles/round5_lane_phase_clocks.txt,
// round6_lane_balance.txt): waves w and w + 4 share a SIMD, the older one (w) runs at its own dependent-issue rate (~ 440 cycles per step)
// whatever its partner does, the younger one gets the issue slots that leaves (~ 0.55 steps per step of the older wave) and runs the rest of
// its steps alone afterwards -- so a SIMD with T1 >= T2 steps costs max(T1, 0.45 T1 + T2) step times, cheapest at T2 ~ 0.55 T1 ([75, 38]
// instead of [75, 50] + [50, 50]); and a wave that MIXES item lengths runs the steps past its shortest item masked, ~ 15 % dearer each (the
// wave of 56 x 75 + 8 x 38 steps: 35.6 k cycles per pass against 33.1 k for its unmixed neighbours), which is what d is for: it moves the
// boundary between long and short items onto a wave boundary.
// T[w] = steps of wave w = of its longest item (0: no such wave), Tmin[w] = steps of its shortest item; in units of 1/100 step.
#ifndef PD_LANE_MORE_MAX
#define PD_LANE_MORE_MAX 3
#endif
#ifndef PD_LANE_MORE_SLACK
#define PD_LANE_MORE_SLACK 16
#endif
__host__ __device__ inline int pd_lane_pass_cost(const int *T, const int *Tmin) {
    int c = 0;
    for (int s = 0; s < 4; ++s) {
        int t1 = 100 * T[s] + 15 * (T[s] - Tmin[s]), t2 = 0;
        for (int w = s + 4; w < PD_LANE_WAVES; w += 4) t2 += 100 * T[w] + 15 * (T[w] - Tmin[w]);
        const int a = t1, b = (45 * t1) / 100 + t2;
        c = c > a ? c : a;
        c = c > b ? c : b;
    }
    return c;
}

struct PdSeqHost {
    void *blob = nullptr;      // one hipMalloc holding every array of the PdSeqDesc
    size_t blob_bytes = 0;     // its capacity (re-used by later uploads that fit)
    PdSeqDesc desc{};          // host shadow; for device-built slots (pd_ggs_set_matches_csr_async) the counts are CAPACITIES
    int n_local_max_k1 = 0;
    int max_item_len = 0;      // longest work item (matches); for device-built slots the per-pair hint (or the 512 maximum)
    int max_deg = 0;           // most frame pairs incident to one frame (host-built: exact; device-built: an upper bound from the capacities)
    bool device_built = false; // tables + descriptor were written by the ingestion kernels; the host never saw the counts
};

// ---- denoiser -----------------------------------------------------------------------------------
struct PdDenoiserDev;   // defined in pd_denoiser.hip

struct pd_engine {
    int device = 0;
    int num_cus = 0;           // multiProcessorCount (bounds the resident workgroups of the GGS exchange)
    int max_B = 0, max_N = 0;
    int d_model = 0, nhead = 0, dim_ff = 0, num_layers = 0, z_dim = 0, timesteps = 0;
    PdDenoiserDev *den = nullptr;
    // schedule tables (host copies; kernels take the per-step scalars by value)
    std::vector<float> c_recip, c_recipm1, coef1, coef2, logvar;
    int pred_x0 = 0;                      // pd_weights.reserved & PD_WEIGHTS_PRED_X0
    // GGS
    std::vector<PdSeqHost> seqs;
    PdSeqDesc *d_seqs = nullptr;         // [max_B] device copy of the descriptors
    unsigned long long *d_xchg = nullptr;
    size_t xchg_granules = 0;            // per (sequence, slot)
    unsigned int *d_err = nullptr;       // [0] async error word; [2..] debug phase counters
    int ggs_prof_on = 0;
    int den_fused_attn = 1;          // PD_OPT_DENOISER_FUSED_ATTN: in the fp16-plane mode, in_proj + attention as one kernel with Q / K / V in LDS (N <= 32)
    int den_split = 0;               // PD_OPT_DENOISER_SPLIT: encoder GEMMs of the large-batch path: 0 exact fp32, 1 bf16 planes, 2 fp16 planes (default there)
    int gemm_wide_min_tiles = 200;   // launch_gemm: 32-wide tiles when there are at least this many of them
    float *d_stats_scratch = nullptr;
    unsigned long long *d_stamps = nullptr;   // [PD_STAMP_SLOTS][2] launch stamps of the GGS launches (PdGgsParams::stamp); slot = guided step index of the
    int stamp_slot = 0;                       //   sampling loop (issue_loop sets it before every pd_ggs_guide), 0 for the step-level API
    // sampler buffers (fixed addresses so a captured graph can be replayed)
    float *d_z = nullptr, *d_noise = nullptr, *d_process = nullptr, *d_mean = nullptr, *d_stats = nullptr;
    // graph cache
    struct GraphKey {
        int B, N, cond_start, has_ggs, phase, den_split;
        pd_ggs_cfg cfg;
        PdGgsPlan plan;            // match-derived launch shape baked into the captured GGS nodes
    };
    std::vector<std::pair<GraphKey, hipGraphExec_t>> graphs;   // most recently used last; at most PD_GRAPH_CACHE_MAX entries (LRU eviction)
    hipStream_t own_stream = nullptr;
    // One event PER STREAM (an engine may be driven from several): `uses` are recorded after every enqueue that reads the match tables
    // (uploads wait for all of them: this engine's work only), `uploads` after the ingestion kernels of pd_ggs_set_matches_csr_async
    // (GGS launches on any stream wait for all of them, on the device).  A single re-recorded event would only remember the last stream.
    struct StreamEvent {
        hipStream_t stream;
        hipEvent_t event;
    };
    std::vector<StreamEvent> uses, uploads;
    // outgrown slot blobs that work in flight at retirement time may still read: freed once `done` (recorded on the retiring upload's
    // stream after it waited for every use) has completed -- checked at the next upload, and at the latest with the engine
    struct RetiredBlob {
        void *ptr;
        hipEvent_t done;
    };
    std::vector<RetiredBlob> retired_blobs;
};

// pd_denoiser.hip
int pd_denoiser_create(pd_engine *eng, const pd_weights *w);
void pd_denoiser_destroy(pd_engine *eng);
bool pd_denoiser_has_streamed_path(const pd_engine *eng);   // created with max_B x max_N >= PD_STREAM_MIN_ROWS token rows
bool pd_denoiser_weights_non_finite(const pd_engine *eng);  // the fp16-plane scales were refused because an encoder weight / bias is inf or NaN
int pd_denoiser_build_split(pd_engine *eng, int mode);    // 1: bf16 planes (fast mode), 2: fp16 planes with static scales
// eps_out / mean_out / x_next_out may each be null. noise null => 0.
int pd_denoiser_launch(pd_engine *eng, const float *x, const float *z, int t, int B, int N, float *eps_out,
                       float *mean_out, float *x0_out, const float *noise, float *x_next_out, hipStream_t s, bool z_prepared = false);
// the step-invariant piece of _first for this z (once per sampling call; pd_denoiser_launch(..., z_prepared = true) then skips it)
int pd_denoiser_prepare(pd_engine *eng, const float *z, int B, int N, hipStream_t s);

// pd_ggs.hip
int pd_ggs_init();
int pd_ggs_plan(pd_engine *eng, int B, int N, const pd_ggs_cfg *cfg, PdGgsPlan *out);
int pd_ggs_launch(pd_engine *eng, float *x, int B, int N, const PdGgsStage *stages, int n_stages,
                  const pd_ggs_cfg *cfg, int eval_only, float *stats, float *trace, int trace_iters,
                  float *loss_out, float *grad_out, hipStream_t s);
void pd_ggs_free_seq(PdSeqHost &h);
int pd_ggs_ingest_init();   // pd_ggs_ingest.hip
int pd_wait_uploads(pd_engine *eng, hipStream_t s);   // device-side wait for pending asynchronous match uploads (no-op in a capture)
int pd_mark_use(pd_engine *eng, hipStream_t s);
int pd_record_stream_event(std::vector<pd_engine::StreamEvent> &list, hipStream_t s);   // (re)record this stream's event of the list
int pd_wait_uses(pd_engine *eng, hipStream_t s, bool host);   // wait (device side on s, or on the host) for every recorded use
#define PD_GRAPH_CACHE_MAX 8
#define PD_STAMP_SLOTS 128
