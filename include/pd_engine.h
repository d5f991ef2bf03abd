/*
 * pd_engine.h -- C-ABI of the MI355X-native PoseDiffusion sampling engine (libpd_engine.so).
 *
 * This is the drop-in boundary for ONE hot path of facebookresearch/PoseDiffusion: the DDPM
 * reverse loop over camera-pose tokens and the Geometry-Guided-Sampling (GGS) Sampson step.
 * The reference is pure Python (no FFI of its own), so each entry point below names the
 * reference Python function it replaces (paths relative to /root/reference/pose_diffusion/);
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / C++ types.
 *   - every `float*` / `const float*` documented as DEVICE is a HIP device pointer to
 *     contiguous row-major fp32 owned by the caller; the engine borrows it for the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the HIP default stream).  All calls
 *     are asynchronous on that stream unless stated; the engine has no host threads.
 *   - return value: 0 = PD_OK, negative = error; pd_last_error() gives the message
 *     (thread-local).  The "insufficient valid matches" early exit of GGS is not an error
 *     (geometry_guided_sampling.py:104-108).
 *   - one engine per device; not re-entrant per instance.
 */
#ifndef PD_ENGINE_H
#define PD_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_OK 0
#define PD_ERR_INVALID_ARG (-1)   /* NULL pointer, non-positive size, t out of range ...          */
#define PD_ERR_UNSUPPORTED (-2)   /* shape family the kernels are not built for                  */
#define PD_ERR_HIP (-3)           /* a HIP runtime call failed (message carries hipGetErrorString)*/
#define PD_ERR_STATE (-4)         /* e.g. GGS requested for a sequence with no matches uploaded  */

#define PD_MAX_LAYERS 16
#define PD_POSE_DIM 9             /* absT(3) | quaR wxyz(4) | logFL(2), camera_transform.py:80-97 */

typedef struct pd_engine pd_engine;

/* One nn.TransformerEncoderLayer (models/denoiser.py:88-97, cfgs/default.yaml:27-35).
 * PyTorch layouts: Linear weight [out, in] row-major.  DEVICE pointers. */
typedef struct pd_layer_weights {
    const float *norm1_w, *norm1_b;         /* [d]                                   */
    const float *in_proj_w, *in_proj_b;     /* [3d, d], [3d]   (q | k | v rows)      */
    const float *out_proj_w, *out_proj_b;   /* [d, d], [d]                           */
    const float *norm2_w, *norm2_b;         /* [d]                                   */
    const float *linear1_w, *linear1_b;     /* [ff, d], [ff]                         */
    const float *linear2_w, *linear2_b;     /* [d, ff], [d]                          */
} pd_layer_weights;

/* Everything `Denoiser` + `GaussianDiffusion` hold that the sampler reads
 * (models/denoiser.py:23-51, util/embedding.py:13-37, models/gaussian_diffuser.py:157-182).
 * All pointers DEVICE fp32; the engine repacks/copies them at create time and keeps no
 * reference to the caller's storage afterwards. */
typedef struct pd_weights {
    int32_t d_model;        /* 512  */
    int32_t nhead;          /* 4    */
    int32_t dim_ff;         /* 1024 */
    int32_t num_layers;     /* 8    */
    int32_t z_dim;          /* 384  */
    int32_t n_harmonic;     /* 10   (PoseEmbedding, embedding.py:41)                     */
    int32_t t_emb_dim;      /* 256  (TimeStepEmbedding.dim; output is dim/2 = 128)       */
    int32_t mlp_hidden;     /* 128  (Denoiser.mlp_hidden_dim)                            */
    int32_t timesteps;      /* 100                                                       */
    int32_t reserved;       /* flags; 0 = objective "pred_noise" (cfgs/default.yaml), PD_WEIGHTS_PRED_X0 = "pred_x0" */
    const float *time_w0, *time_b0;     /* time_embed.linear.0  [128,256],[128]          */
    const float *time_w2, *time_b2;     /* time_embed.linear.2  [128,128],[128]          */
    const float *first_w, *first_b;     /* _first  [d, 189+128+z_dim+1 = 702], [d]       */
    pd_layer_weights layers[PD_MAX_LAYERS];
    const float *last0_w, *last0_b;     /* _last.0 [128, d], [128]                       */
    const float *last_ln_w, *last_ln_b; /* _last.1 LayerNorm [128]                       */
    const float *last3_w, *last3_b;     /* _last.3 [9,128], [9]                          */
    /* schedule tables, each [timesteps] (gaussian_diffuser.py:167-182) */
    const float *sqrt_recip_alphas_cumprod;
    const float *sqrt_recipm1_alphas_cumprod;
    const float *posterior_mean_coef1;
    const float *posterior_mean_coef2;
    const float *posterior_log_variance_clipped;
} pd_weights;

#define PD_WEIGHTS_PRED_X0 1   /* pd_weights.reserved: GaussianDiffusion(objective="pred_x0") -- the denoiser's output IS x_start
                                * (models/gaussian_diffuser.py:225-227); pd_p_mean / pd_sample then skip predict_start_from_noise
                                * and pd_denoise_step still returns the raw model output */

/* GGS knobs: cfgs/default.yaml:6-13 as passed through **GGS_cfg to GGS_optimize
 * (geometry_guided_sampling.py:67-81). */
typedef struct pd_ggs_cfg {
    float alpha;            /* 1e-4  */
    float learning_rate;    /* 1e-2  */
    int32_t iter_num;       /* 100 (doubled when R, T and FL are all updated, :86-87) */
    float sampson_max;      /* 10    */
    int32_t min_matches;    /* 10    */
    float momentum;         /* 0.9 (torch.optim.SGD at :89) */
    int32_t wgs_per_seq;    /* 0 = engine picks; >0 = workgroups cooperating on one sequence */
    int32_t reserved;       /* flags; 0 by default.  PD_GGS_CFG_FORCE_ONE_HOP: for sequences with more than 32 frames keep
                             * the single-exchange kernel (every workgroup back-propagates all pairs) instead of the
                             * two-hop kernel that distributes the backward (testing / comparison) */
} pd_ggs_cfg;
#define PD_GGS_CFG_FORCE_ONE_HOP 1
#define PD_GGS_CFG_NO_LDS_STAGING 2   /* pd_ggs_cfg.reserved: stream match items through registers even where the LDS-DMA double
                                       * buffer applies (several items per wavefront, every item <= 384 matches); same arithmetic,
                                       * bitwise the same results -- comparison / testing */
#define PD_GGS_CFG_WAVES8 4           /* pd_ggs_cfg.reserved: keep 8 wavefronts per workgroup where the staged one-workgroup-per-
                                       * sequence shape would run 12 (three per SIMD); bitwise the same results -- comparison */
#define PD_GGS_CFG_LANE_ITEMS 8       /* pd_ggs_cfg.reserved: run the lane-per-item kernel (one workgroup of 8 wavefronts per sequence,
                                       * every LANE owns <= l matches of one frame pair, 14 steps of every lane item resident in registers and 3 in LDS
                                       * for the whole launch, the rest streamed through an LDS ring) whatever the batch size, where its tables
                                       * exist (<= 24 frames, <= 512 frame pairs) and its LDS image fits.  Without the flag it is picked only
                                       * when wgs_per_seq == 0 and the launch holds more sequences than half the CUs.  Same valid sets and
                                       * formulas as the wave-per-item kernels, another (fixed) summation order: results agree to rounding
                                       * (~1e-6), not bit for bit */
#define PD_GGS_CFG_NO_LANE_ITEMS 16   /* pd_ggs_cfg.reserved: never pick the lane-per-item kernel (comparison / testing) */
#define PD_GGS_CFG_XCHG_SPREAD 32     /* pd_ggs_cfg.reserved: with several workgroups per sequence, do NOT place a sequence's workgroups on one XCD
                                       * (where the engine would: at most 32 of them per XCD) -- the exchange then goes through write-through
                                       * agent-scope stores as in rounds 1-3; bitwise the same results -- comparison / testing */

/* ---- lifecycle -------------------------------------------------------------------------- */

/* Build an engine on the current HIP device.  Replaces the module construction +
 * load_state_dict of demo.py:46,56-57 for the sampling path.  Synchronous.
 * max_B sequences x max_N frames bound every later call (workspaces are sized once). */
int pd_engine_create(const pd_weights *w, int max_B, int max_N, pd_engine **out);
void pd_engine_destroy(pd_engine *eng);
const char *pd_last_error(void);
/* ABI/version string, e.g. "pd_engine 0.1 gfx950". */
const char *pd_version(void);

/* ---- denoiser + DDPM (GaussianDiffusion.p_sample pieces) --------------------------------- */

/* eps_out[B,N,9] = Denoiser.forward(x[B,N,9], t (same for all B), z[B,N,z_dim])
 * (models/denoiser.py:53-76).  x, z, eps_out DEVICE.
 * Cost note for the step-level entry points (pd_denoise_step, pd_p_mean): at >= 1 024 token rows `_first` is computed as a
 * step-invariant z piece + a per-step piece (models/denoiser.py:56-70: z does not change over the T steps).  pd_sample /
 * pd_sample_phase compute the z piece ONCE per call; the step-level API cannot know that z is unchanged and recomputes it on
 * every call (one extra launch and a K = 384 GEMM per step) -- a host loop over T steps should use pd_sample. */
int pd_denoise_step(pd_engine *eng, const float *x, const float *z, int t, int B, int N,
                    float *eps_out, void *stream);

/* model_mean of p_mean_variance (gaussian_diffuser.py:231-246): runs the denoiser and
 * x0 = c_recip[t] x - c_recipm1[t] eps  (or x0 = the model output under PD_WEIGHTS_PRED_X0) ; mean = coef1[t] x0 + coef2[t] x.
 * mean_out[B,N,9]; x0_out may be NULL.  DEVICE pointers. */
int pd_p_mean(pd_engine *eng, const float *x, const float *z, int t, int B, int N,
              float *mean_out, float *x0_out, void *stream);

/* pred = mean + exp(0.5 * posterior_log_variance_clipped[t]) * noise  (gaussian_diffuser.py:280);
 * noise == NULL means noise = 0 (guided steps and t == 0, :276-278).  DEVICE pointers. */
int pd_p_finish(pd_engine *eng, const float *mean, const float *noise, int t, int B, int N,
                float *x_out, void *stream);

/* ---- Geometry-Guided Sampling ------------------------------------------------------------ */

/* Upload the matches of sequence slot `seq` (0 <= seq < max_B).  Replaces the per-call host
 * prep of geometry_guided_sampling.py:16-45: HOST pointers exactly as demo.py:82-84 holds them
 * (kp1/kp2 float64 [M,2] pixel coords in the cropped+resized image, i12 int64 [M,2] frame
 * indices); the fp64->fp32 cast of :167, the pair index of :26-27 and the homogeneous pad of
 * :29-33 happen inside.  n_frames/height/width = img_shape[0], [2], [3] (:16).
 * Synchronous (allocates).  M == 0 clears the slot. */
int pd_ggs_set_matches(pd_engine *eng, int seq, const double *kp1, const double *kp2,
                       const int64_t *i12, int64_t M, int n_frames, int height, int width);

/* Asynchronous, device-resident ingestion of the matches of `n_seqs` consecutive sequence slots (seq_first ..) -- the
 * batched match container.  The reference re-uploads kp1 / kp2 / i12 from numpy on every
 * guided step (geometry_guided_sampling.py:19-24); here a batch goes up once, without the host ever waiting:
 *   seq_offsets  HOST   [n_seqs + 1]  CSR offsets: sequence b owns rows [seq_offsets[b], seq_offsets[b+1]) of the arrays below
 *   kp1, kp2     DEVICE-ACCESSIBLE float64 [total, 2]  (device memory, or pinned host memory the GPU reads over PCIe)
 *   i12          DEVICE-ACCESSIBLE int64   [total, 2]  frame indices -- the dtypes demo.py:82-84 holds
 * The fp64 -> fp32 cast (:167), pair key (:26-27), the STABLE sort by pair and every table the GGS kernels read are built
 * by kernels on `stream` (csrc/pd_ggs_ingest.hip); results are bitwise those of pd_ggs_set_matches on the same data.  No
 * host synchronisation and, once a slot's buffer fits, no allocation; GGS launches of this engine issued later on any
 * stream are ordered after the upload on the device.  The arrays must stay valid until the upload has executed.
 * Because the host never learns the pair / item counts, launch shapes are planned from capacities: `hints` (may be NULL =
 * worst case: min(n_frames^2, M) pairs, any number of matches per pair) lets the caller declare tighter ones, e.g.
 * {190, 512} for hloc's exhaustive i < j pairs of 20 frames with <= 512 matches each; with max_matches_per_pair in
 * 1..512 sequences of more than 32 frames take the two-hop kernel.  A violated hint empties the slot and, like a frame
 * index outside [0, n_frames), raises the asynchronous error word (pd_check_async_error). */
#define PD_MATCH_HINT_ONE_ORDER (1 << 30)   /* OR-ed into pd_match_hints.max_pairs: every frame pair occurs in ONE order only (hloc's exhaustive
                                            * i < j pairs, match_extraction.py:64-70), i.e. no frame is in more than n_frames - 1 pairs.  Lets
                                            * the engine plan the launch shape of host-uploaded tables (the lane-per-item kernel needs that bound)
                                            * for device-built ones too; checked on the device like every hint */
typedef struct pd_match_hints {
    int32_t max_pairs;               /* 0 = unknown; else an upper bound on the frame pairs that own matches (| PD_MATCH_HINT_ONE_ORDER) */
    int32_t max_matches_per_pair;    /* 0 = unknown; else an upper bound on the matches of one frame pair               */
} pd_match_hints;
int pd_ggs_set_matches_csr_async(pd_engine *eng, int seq_first, int n_seqs, const int64_t *seq_offsets, const double *kp1,
                                 const double *kp2, const int64_t *i12, int n_frames, int height, int width,
                                 const pd_match_hints *hints, void *stream);

/* model_mean[B,N,9] (DEVICE, in/out) <- geometry_guided_sampling(model_mean, t, ...) for every
 * sequence b using match slot b: the five GGS_optimize calls (all, FL, R, T, all) of
 * geometry_guided_sampling.py:48-63.  stats_out (DEVICE, may be NULL) receives [B,5,4] floats:
 * per optimisation {sampson_to_print (:169), iterations stepped, last n_valid, last loss}. */
int pd_ggs_guide(pd_engine *eng, float *model_mean, int B, int N, int t, const pd_ggs_cfg *cfg,
                 float *stats_out, void *stream);

/* Iterations each of the five GGS_optimize calls of one guided step is GIVEN under `cfg` (geometry_guided_sampling.py:48-63, :86-87:
 * all (2 x iter_num), FL, R, T, all (2 x)) -- the engine's own stage table, for hosts that compare it with the iterations a stage
 * STEPPED (stats[.., 1]) to reproduce the reference's "Drop this pair ..." line (:104-108).  No GPU needed. */
int pd_ggs_stage_iters(const pd_ggs_cfg *cfg, int *iters_out5);

/* One GGS_optimize call (geometry_guided_sampling.py:67-126) with explicit update flags;
 * used by the parity tests.  trace_out (DEVICE, may be NULL): [B, iters, N*9 + 3] per-iteration
 * {x after the step, loss, n_valid, grad-norm} for iterations actually stepped. */
int pd_ggs_optimize(pd_engine *eng, float *model_mean, int B, int N, int update_R, int update_T,
                    int update_FL, const pd_ggs_cfg *cfg, float *stats_out, float *trace_out,
                    void *stream);

/* Sampson loss + analytic gradient at x (no update): loss_out[B,4] = {mean valid sampson,
 * n_valid, mean(clamp(s, max)), 0}; grad_out[B,N,9].  Parity-test entry for
 * compute_sampson_distance + backward (geometry_guided_sampling.py:129-172, :110-112). */
int pd_ggs_loss_grad(pd_engine *eng, const float *x, int B, int N, int update_R, int update_T,
                     int update_FL, const pd_ggs_cfg *cfg, float *loss_out, float *grad_out,
                     void *stream);

/* ---- whole sampler ------------------------------------------------------------------------ */

/* Engine options (no reference counterpart).
 *   PD_OPT_DENOISER_SPLIT  how the four Linear layers of every encoder layer run for batches of >= 1024 token rows (smaller
 *        batches always use the exact-fp32 matrix instruction).  Default: 2 on engines created with max_B x max_N >= 1024, else 0
 *        -- i.e. the results of such an engine (and bench.py's `value`) are those of the fp16-plane arithmetic described under 2
 *        (22 operand bits, fp32 accumulation), not of the exact-fp32 instruction; select 0 for the latter.  Weights that hold
 *        inf / NaN have no static bounds: such an engine stays on 0 and an explicit request for 2 returns PD_ERR_INVALID_ARG.
 *        0: every GEMM of the denoiser on the exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32).
 *        1: FAST MODE: the four Linear layers of every encoder layer run in split precision
 *        on the bf16 matrix pipe (each fp32 operand as bf16 hi + bf16 lo, three products, fp32 accumulation: ~16 mantissa
 *        bits per operand); LayerNorm, softmax, residuals, `_first`, `_last` and the DDPM update stay fp32.  Narrower
 *        arithmetic than the reference's: a separately reported mode, never selected by the engine itself (deviation measured in
 *        tests/test_gpu_parity_r2.py and profiles/round2_denoiser_precision_study.json).  Smaller batches ignore it.
 *        2: the same kernels with FP16 halves (11 + 11 = 22 mantissa bits per operand, the dropped lo*lo term is 2^-22 of a
 *        product, fp32 accumulation) and power-of-two operand scales fixed at the switch from bounds that hold for every input
 *        (LayerNorm output <= sqrt(d); a Linear fed by it <= sqrt(d) ||w|| + |b|): no overflow, no underflow, and results as
 *        close to the fp64 product as the exact-fp32 instruction's (tests/test_gpu_parity_r3.py).  Same batches as 1. */
#define PD_OPT_DENOISER_SPLIT 2
/*   (option 3, PD_OPT_DENOISER_PERSISTENT of round 3 -- a denoiser evaluation of <= 32 token rows as ONE persistent launch -- is no longer
 *    built: correct but 2.4 x slower than the 43-launch path, profiles/round3_small_persistent.txt; source parked under tools/parked/.
 *    Value 0 is accepted, 1 returns PD_ERR_UNSUPPORTED.) */
/*   PD_OPT_DENOISER_FUSED_ATTN  (fp16-plane mode, sequences of <= 32 frames) 1 (default): the in_proj Linear and the attention of a head run as
 *        ONE kernel per group of whole sequences, Q / K / V held in LDS and never written to memory (csrc/pd_qkv_attn.h; models/denoiser.py:88-97);
 *        used where its workgroups (one per 95 / N sequences and head) fill at least three quarters of the chip's rounds -- 256 sequences of 20
 *        frames = 256 workgroups; a batch of 103 keeps the two launches; 2: always; 0: never = the two launches it replaces (QKV GEMM -> fp32
 *        QKV in memory -> attention).  Same arithmetic in the same order: bitwise the same results whatever the choice. */
#define PD_OPT_DENOISER_FUSED_ATTN 5
int pd_engine_set_option(pd_engine *eng, int option, int value);
/* Reads an option back.  PD_OPT_DENOISER_SPLIT: the mode in force (an engine created from weights that hold inf / NaN stays on 0 although
 * it is large enough for 2 -- the only downgrade pd_engine_create performs by itself; PD_OPT_WEIGHTS_NON_FINITE (read-only) then reads 1). */
#define PD_OPT_WEIGHTS_NON_FINITE 4
int pd_engine_get_option(pd_engine *eng, int option, int *value_out);

/* GaussianDiffusion.sample / p_sample_loop (gaussian_diffuser.py:284-306).
 *   z      [B,N,z_dim]                      DEVICE
 *   noise  [T+1,B,N,9]                      DEVICE  noise[0] = the randn(shape) of :289;
 *          noise[1+k] = the randn_like of step t = T-1-k (:278); slots the reference never
 *          draws (t == 0, guided steps) are ignored.
 *   cond_start_step / ggs: if ggs != NULL, steps with t < cond_start_step run pd_ggs_guide on
 *          the model mean with noise = 0 (:270-276); ggs == NULL = unguided.
 *   pose_out    [B,N,9]       DEVICE
 *   process_out [T+1,B,N,9]   DEVICE, may be NULL
 *   stats_out   [cond_start_step,B,5,4] DEVICE, may be NULL (see pd_ggs_guide), ordered by
 *          guided step t = cond_start_step-1 .. 0
 * use_graph != 0 replays a cached hipGraph of the whole loop (captured on first use). */
int pd_sample(pd_engine *eng, const float *z, const float *noise, int B, int N, int cond_start_step,
              const pd_ggs_cfg *ggs, float *pose_out, float *process_out, float *stats_out,
              int use_graph, void *stream);

/* The same loop in two halves, for callers that software-pipeline several batches (one engine
 * and one stream per batch in flight): the unguided steps t = T-1 .. cond_start_step need few
 * CUs, the guided steps t < cond_start_step hold wgs_per_seq CUs per sequence for milliseconds
 * (persistent, co-resident workgroups), so a scheduler gates PD_PHASE_GUIDED on an event while
 * PD_PHASE_UNGUIDED of later batches runs beside it (posediffusion_amd/pipeline.py).
 *   PD_PHASE_ALL      = pd_sample
 *   PD_PHASE_UNGUIDED : uploads z / noise and runs the unguided steps; pose_out is not written
 *   PD_PHASE_GUIDED   : continues from the engine's state of the preceding PD_PHASE_UNGUIDED call
 *                       with the same arguments, then writes pose_out / process_out / stats_out
 * Both halves must be issued on the same stream (or ordered by the caller). */
#define PD_PHASE_ALL 0
#define PD_PHASE_UNGUIDED 1
#define PD_PHASE_GUIDED 2
int pd_sample_phase(pd_engine *eng, const float *z, const float *noise, int B, int N, int cond_start_step,
                    const pd_ggs_cfg *ggs, int phase, float *pose_out, float *process_out, float *stats_out,
                    int use_graph, void *stream);

/* Final decode pose_encoding_to_camera (camera_transform.py:64-105): enc[B*N,9] ->
 * R[B*N,9] row-major 3x3, T[B*N,3], focal[B*N,2] (PyTorch3D NDC).  DEVICE pointers. */
int pd_pose_to_camera(pd_engine *eng, const float *enc, int n_cameras, float *R_out, float *T_out,
                      float *focal_out, void *stream);
/* The same with the reference function's three keyword parameters (camera_transform.py:64-70, :89-97):
 * focal = clamp(exp(logFL + log_focal_length_bias), min_focal_length, max_focal_length); pd_pose_to_camera = (1.8, 0.1, 20).
 * (The guided sampler itself always decodes with the defaults, as geometry_guided_sampling.py:139 does.) */
int pd_pose_to_camera_ex(pd_engine *eng, const float *enc, int n_cameras, float *R_out, float *T_out,
                         float *focal_out, float log_focal_length_bias, float min_focal_length, float max_focal_length,
                         void *stream);

/* ---- rows D2 / D3 as stand-alone operators (stateless, all pointers DEVICE fp32) --------------------
 * The engine fuses both embeddings into the denoiser (a [T,128] table built at creation; the harmonic columns formed while
 * _first's rows are staged).  These two entry points run the same device code for a caller that uses the reference's
 * util/embedding.py modules piecewise (dropin/util/embedding.py binds them). */

/* TimeStepEmbedding.forward (util/embedding.py:28-37, dim = 256): out[n,128] = linear.2(SiLU(linear.0([cos(t f) | sin(t f)]))),
 * f_i = exp(-ln(10000) i / 128); w0 [128,256], b0 [128], w2 [128,128], b2 [128] = the module's linear.0 / linear.2 tensors,
 * timesteps[n] already converted to float (the reference's `timesteps[:, None].float()`). */
int pd_time_embedding(const float *w0, const float *b0, const float *w2, const float *b2, const float *timesteps, int n,
                      float *out, void *stream);

/* PoseEmbedding.forward (util/embedding.py:52-54) = pytorch3d HarmonicEmbedding(n_harmonic_functions = 10, append_input = True):
 * x[rows,dim] -> out[rows, 21 dim] = [sin(x_d 2^k) | sin(x_d 2^k + pi/2) | x], d-major, k = 0..9 (189 columns for dim = 9);
 * 1 <= dim <= 4096, rows >= 0 (an empty batch is a no-op). */
int pd_pose_embedding(const float *x, long long rows, int dim, float *out, void *stream);

/* ---- evaluation metrics (stateless, all pointers DEVICE fp32) ------- */

/* camera_to_rel_deg (util/metric.py:14-47): R_*[B*N,9] row-major 3x3, T_*[B*N,3] in the PyTorch3D convention
 * (X_view = X_world R + T); for every sequence b and every pair i < j in torch.combinations order (:106-111) the angle
 * in degrees between the ground-truth and the predicted relative rotation (so3_relative_angle, :143-151) and between the
 * relative translation directions (:154-172; nan/inf -> 1e6).  Outputs [B * N(N-1)/2] each. */
int pd_metrics_rel_pose_errors(const float *R_pred, const float *T_pred, const float *R_gt, const float *T_gt, int B, int N,
                               float *rel_r_deg, float *rel_t_deg, void *stream);

/* out7 = {Auc_max_threshold (calculate_auc_np, util/metric.py:50-78), Racc_5, Racc_15, Racc_30, Tacc_5, Tacc_15,
 * Tacc_30 (test.py:113-119, percent)} over n error pairs. */
int pd_metrics_summary(const float *rel_r_deg, const float *rel_t_deg, int n, int max_threshold, float *out7, void *stream);

/* compute_ARE (util/metric.py:174-185): absolute rotation error in degrees of n rotation pairs [n,9]. */
int pd_metrics_are(const float *R_a, const float *R_b, int n, float *err_deg, void *stream);

/* pytorch3d.ops.corresponding_cameras_alignment(cameras_src, cameras_tgt, estimate_scale, mode="extrinsics", eps) as
 * demo.py:127-129 calls it: similarity (s, R_A, T_A) that best maps the n source cameras onto the target cameras;
 * R_out[n,9] = R_A R_src, T_out[n,3] = T_A R_src + s T_src; s_R_T_out (may be NULL) receives {s, R_A[9], T_A[3]}. */
int pd_align_cameras(const float *R_src, const float *T_src, const float *R_tgt, const float *T_tgt, int n, int estimate_scale,
                     float eps, float *R_out, float *T_out, float *s_R_T_out, void *stream);

/* ---- image preprocessing (stateless) -------------------------------------- */

/* One frame of load_and_preprocess_images (util/load_img_folder.py:15-48): rgb_hwc [height, width, 3] uint8 DEVICE ->
 * out_chw [3, image_size, image_size] float32 DEVICE in [0, 1]: centre crop to min(h, w) (:68-73), bilinear resize with
 * torch's align_corners=False rule, no antialiasing (:35-40). */
int pd_preprocess_image(const unsigned char *rgb_hwc, int height, int width, int image_size, float *out_chw, void *stream);

/* ---- image features -------------------------------------------------------- */

/* MultiScaleImageFeatureExtractor (models/image_feature_extractor.py:28-87) around a DINO ViT-S/16 (:40-42, third-party
 * facebookresearch/dino vision_transformer.py; restated).  All weight pointers DEVICE fp32 in the layouts of the DINO
 * state_dict: patch_embed.proj.weight [384,3,16,16], cls_token [384], pos_embed [1 + pos_grid^2, 384], per block
 * norm1/norm2 [384], attn.qkv [1152,384], attn.proj [384,384], mlp.fc1 [1536,384], mlp.fc2 [384,1536], norm [384].
 * The engine repacks what it needs at creation; the caller's tensors are not referenced afterwards. */
typedef struct pd_vit_layer_weights {
    const float *norm1_w, *norm1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} pd_vit_layer_weights;
typedef struct pd_vit_weights {
    int32_t dim, depth, num_heads, mlp_hidden, patch_size, pos_grid, reserved0, reserved1;   /* 384, 12, 6, 1536, 16, 14 */
    const float *patch_w, *patch_b, *cls_token, *pos_embed, *norm_w, *norm_b;
    pd_vit_layer_weights layers[16];
} pd_vit_weights;
typedef struct pd_vit pd_vit;
int pd_vit_create(const pd_vit_weights *w, pd_vit **out);
void pd_vit_destroy(pd_vit *v);
/* PD_VIT_OPT_EXACT_FP32, the arithmetic of batches of >= 1024 token rows (smaller batches are exact fp32 whatever it says):
 *   0 (default) = the four Linear layers of every block on fp16 hi + fp16 lo operand planes (22 mantissa bits, power-of-two scales fixed at
 *       creation from bounds that hold for every input, three fp16 matrix products, fp32 accumulation) -- the denoiser's default mode;
 *       LayerNorm, QK^T, softmax, PV, GELU (exact erf form) and residuals in fp32: fp32-grade features.  Non-finite weights have no bound:
 *       such a network runs mode 1;
 *   1 = the exact-fp32 matrix instruction everywhere;
 *   2 = rounds 1-5's bf16 hi + lo planes (16 bits, QK^T on them too, polynomial GELU): features ~1e-5 of max|z| off, kept for comparison. */
#define PD_VIT_OPT_EXACT_FP32 1
int pd_vit_set_option(pd_vit *v, int option, int value);

/* One scale of _compute_multiscale_features (:65-84): images [n_img,3,H,W] DEVICE fp32 in [0,1] -> ImageNet normalisation
 * (:62-63), F.interpolate(scale_factor, bilinear, align_corners=False) (:86-87; skipped for scale_factor == 1), ViT,
 * CLS feature after the final LayerNorm;  z_out[n_img,384] = (accumulate ? z_out : 0) + weight * feature.
 * pos_scaled: DINO's interpolate_pos_encoding for this token grid, [1 + gh*gw, 384] DEVICE (bicubic resampling of the
 * trained grid; the caller -- posediffusion_amd/dropin/models/image_feature_extractor.py -- computes it once per image
 * size); may be NULL when the grid is the trained pos_grid x pos_grid.  At most 1056 tokens per image (<= 512 x 512 pixels
 * after scaling). */
int pd_vit_forward_scale(pd_vit *v, const float *images, int n_img, int H, int W, double scale_factor, const float *pos_scaled,
                         float weight, int accumulate, float *z_out, void *stream);

/* ---- measurement helpers ------------------------------------------------------------------ */

/* Start / end of the GGS launches of the engine's LAST sampling pass, recorded by the kernel itself (lane-per-item kernel: workgroup 0
 * stores wall_clock64() when it starts, every workgroup atomicMax'es it when it ends): dst[k][2] (DEVICE int64) <- {start, end} ticks of
 * the launch of guided step k = 0 .. n-1 (k = cond_start_step - 1 - t; a step-level pd_ggs_guide uses slot 0), copied on `stream`
 * behind the work already enqueued there; *clock_khz_out = tick rate (hipDeviceAttributeWallClockRate).  This is how bench.py times the
 * dominant kernel INSIDE its timed region -- launches replayed from a captured graph, several contexts in flight -- instead of a launch
 * alone on an idle chip.  Slots of launches that ran another GGS kernel read {0, 0}. */
int pd_ggs_launch_stamps(pd_engine *eng, long long *dst, int n, int *clock_khz_out, void *stream);

/* Times `reps` launches of the dominant kernels with hipEvents on `stream` (the stream the
 * kernels run on) and returns average milliseconds per launch.  what: 0 = one full denoiser
 * step (pd_p_mean) at (B,N); 1 = one pd_ggs_guide at (B,N).  Used by bench.py's roofline leg. */
int pd_time_kernel(pd_engine *eng, int what, int B, int N, const pd_ggs_cfg *cfg, int reps,
                   float *ms_out, void *stream);

/* Synchronises the device and reports (PD_ERR_STATE) what the kernels flagged asynchronously since the last check: a
 * bounded spin of the GGS cross-workgroup exchange that gave up (bit 0), an out-of-range frame index (bit 1) or violated
 * pd_match_hints (bit 2) met by pd_ggs_set_matches_csr_async.  Clears the word.  PD_OK otherwise. */
int pd_check_async_error(pd_engine *eng);

/* Debug aid: switch the GGS kernel's in-kernel phase cycle counters on/off and (out6 != NULL)
 * read them: {P1 pair F, P2 matches, exchange, P3 backward, P4 update, iterations} of workgroup 0. */
int pd_debug_ggs_prof(pd_engine *eng, int enable, long long *out6);   /* enable = 1 + wave index to record; out6 holds 16 values */
/* The launch shape pd_ggs_guide / pd_sample would use for (B, N, cfg) with the matches uploaded now:
 * out8 = {workgroups per sequence, item slots per workgroup, LDS bytes, two-hop kernel, waves per workgroup, LDS-DMA staging pieces,
 * lane-per-item kernel, its LDS-resident steps}.  Lets a test pin which kernel the engine picks by itself. */
int pd_debug_ggs_plan(pd_engine *eng, int B, int N, const pd_ggs_cfg *cfg, int *out8);
/* The lane-per-item tables of match slot `seq` as the device holds them (host- or device-built): out[0..3] = {lane items, waves, base item length in
 * matches, steps of the longest wave}, out[4 + w] = steps of wave w (two matches per lane and step).  Synchronises the device.  tests/ compare it with
 * bench_legs.lane_stream_fraction, the Python mirror of the cut rule that bench.py reports streamed bytes from. */
int pd_debug_lane_tables(pd_engine *eng, int seq, int *out, int n_out);
/* Does v_mfma_f32_32x32x16_f16 keep fp16-SUBNORMAL operands (the `lo` halves of small elements in the fp16-plane denoiser mode are
 * subnormal)?  One 32x32x16 product per case, every element of A = a, of B = b: out4 = {C[0][0] for (a, b) = (2^-20, 2^10): 2^-6 if kept;
 * (2^10, 2^-20): 2^-6; (2^-20, 2^-4): 2^-20 (a subnormal times a normal, result far below fp16's range: fp32 accumulation);
 * (1, 1): 16 (control)}; a flushed operand gives 0.  Runs on `stream` and synchronises it. */
int pd_debug_mfma_f16_subnormal(float *out4_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_ENGINE_H */
