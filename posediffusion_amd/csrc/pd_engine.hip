// pd_engine.hip -- C-ABI entry points (include/pd_engine.h), lifecycle, the sampling loop and its
// hipGraph capture.  The kernels live in pd_denoiser.hip and pd_ggs.hip.
//
// Replaces (paths relative to /root/reference/pose_diffusion/):
//   models/gaussian_diffuser.py:248-306   p_sample / p_sample_loop / sample
//   util/camera_transform.py:64-105      pose_encoding_to_camera (final decode)
#include "pd_internal.h"

#include <algorithm>

#include <math.h>
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[1024] = "";

void pd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *pd_last_error(void) { return g_err; }
extern "C" const char *pd_version(void) { return "pd_engine 0.1 gfx950"; }

// ---- small kernels ------------------------------------------------------------------------------
__global__ void pd_finish_kernel(const float *__restrict__ mean, const float *__restrict__ noise, float sigma, int n,
                                 float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = noise ? mean[i] + sigma * noise[i] : mean[i];   // gaussian_diffuser.py:280
}

// torch.clamp: NaN stays NaN (fminf / fmaxf would return the bound)
__device__ __forceinline__ float pd_clamp_torch(float v, float lo, float hi) { return v != v ? v : fminf(fmaxf(v, lo), hi); }

__global__ void pd_camera_kernel(const float *__restrict__ enc, int n, float *__restrict__ R, float *__restrict__ T,
                                 float *__restrict__ F, float fl_bias, float fl_min, float fl_max) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const float *x = enc + (size_t)c * 9;
    const float r = x[3], i = x[4], j = x[5], k = x[6];
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);   // pytorch3d quaternion_to_matrix
    float *o = R + (size_t)c * 9;
    o[0] = 1.0f - two_s * (j * j + k * k);
    o[1] = two_s * (i * j - k * r);
    o[2] = two_s * (i * k + j * r);
    o[3] = two_s * (i * j + k * r);
    o[4] = 1.0f - two_s * (i * i + k * k);
    o[5] = two_s * (j * k - i * r);
    o[6] = two_s * (i * k - j * r);
    o[7] = two_s * (j * k + i * r);
    o[8] = 1.0f - two_s * (i * i + j * j);
    T[c * 3 + 0] = x[0];
    T[c * 3 + 1] = x[1];
    T[c * 3 + 2] = x[2];
    F[c * 2 + 0] = pd_clamp_torch(expf(x[7] + fl_bias), fl_min, fl_max);   // camera_transform.py:89-97
    F[c * 2 + 1] = pd_clamp_torch(expf(x[8] + fl_bias), fl_min, fl_max);
}

// ---- lifecycle ----------------------------------------------------------------------------------
static int fetch_table(std::vector<float> &dst, const float *dev, int n) {
    if (!dev) {
        pd_set_error("pd_engine_create: a schedule table pointer is NULL");
        return PD_ERR_INVALID_ARG;
    }
    dst.resize(n);
    PD_HIP_CHECK(hipMemcpy(dst.data(), dev, sizeof(float) * n, hipMemcpyDeviceToHost));
    return PD_OK;
}

extern "C" void pd_engine_destroy(pd_engine *eng) {
    if (!eng) return;
    (void)hipSetDevice(eng->device);
    (void)hipDeviceSynchronize();
    for (auto &g : eng->graphs) (void)hipGraphExecDestroy(g.second);
    for (auto &e : eng->uses) (void)hipEventDestroy(e.event);
    for (auto &e : eng->uploads) (void)hipEventDestroy(e.event);
    for (auto &s : eng->seqs) pd_ggs_free_seq(s);
    for (auto &r : eng->retired_blobs) {
        (void)hipFree(r.ptr);
        (void)hipEventDestroy(r.done);
    }
    pd_denoiser_destroy(eng);
    void *ptrs[] = {eng->d_seqs, eng->d_xchg, eng->d_err, eng->d_z, eng->d_noise, eng->d_process, eng->d_mean, eng->d_stats, eng->d_stamps};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete eng;
}

extern "C" int pd_engine_create(const pd_weights *w, int max_B, int max_N, pd_engine **out) {
    if (!w || !out || max_B <= 0 || max_N <= 0 || max_N > PD_MAX_FRAMES || w->timesteps <= 0 || w->timesteps > 4096) {
        pd_set_error("pd_engine_create: invalid arguments (max_B=%d, max_N=%d must be in [1,%d])", max_B, max_N, PD_MAX_FRAMES);
        return PD_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (w->reserved & ~PD_WEIGHTS_PRED_X0) {
        pd_set_error("pd_engine_create: unknown pd_weights.reserved flags 0x%x", (unsigned)w->reserved);
        return PD_ERR_INVALID_ARG;
    }
    pd_engine *eng = new pd_engine();
    int rc = PD_OK;
    do {
        if (hipGetDevice(&eng->device) != hipSuccess) {
            pd_set_error("pd_engine_create: no HIP device");
            rc = PD_ERR_HIP;
            break;
        }
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, eng->device) == hipSuccess) eng->num_cus = prop.multiProcessorCount;
        }
        eng->max_B = max_B;
        eng->max_N = max_N;
        eng->d_model = w->d_model;
        eng->nhead = w->nhead;
        eng->dim_ff = w->dim_ff;
        eng->num_layers = w->num_layers;
        eng->z_dim = w->z_dim;
        eng->timesteps = w->timesteps;
        eng->pred_x0 = (w->reserved & PD_WEIGHTS_PRED_X0) ? 1 : 0;
        if ((rc = fetch_table(eng->c_recip, w->sqrt_recip_alphas_cumprod, w->timesteps))) break;
        if ((rc = fetch_table(eng->c_recipm1, w->sqrt_recipm1_alphas_cumprod, w->timesteps))) break;
        if ((rc = fetch_table(eng->coef1, w->posterior_mean_coef1, w->timesteps))) break;
        if ((rc = fetch_table(eng->coef2, w->posterior_mean_coef2, w->timesteps))) break;
        if ((rc = fetch_table(eng->logvar, w->posterior_log_variance_clipped, w->timesteps))) break;
        if ((rc = pd_denoiser_create(eng, w))) break;
        // engines large enough for the streamed path (>= 1024 token rows) run its encoder GEMMs on the fp16 matrix pipe by default:
        // fp16 hi + lo operands, static power-of-two scales, fp32 accumulation -- as close to fp64 as the exact-fp32 instruction
        // (tests/test_gpu_parity_r3.py, profiles/round3_fp16_plane_mode_study.json); PD_OPT_DENOISER_SPLIT = 0 selects the latter
        // (weights with inf / NaN have no static bounds: such an engine stays on the exact-fp32 kernels, which propagate them as
        // the reference does -- only an EXPLICIT request for the mode fails, pd_engine_set_option)
        if (pd_denoiser_has_streamed_path(eng)) {
            rc = pd_denoiser_build_split(eng, 2);
            if (rc == PD_OK) eng->den_split = 2;
            else if (rc == PD_ERR_INVALID_ARG && pd_denoiser_weights_non_finite(eng)) {
                rc = PD_OK;                       // the one intended downgrade; any other failure of the build is an error of the creation
                g_err[0] = 0;                     // ... and not the "last error" of a call that succeeded (pd_engine_get_option reports the mode)
            } else break;
        }
        if ((rc = pd_ggs_init())) break;
        if ((rc = pd_ggs_ingest_init())) break;
        eng->seqs.resize(max_B);
        const int T = w->timesteps;
        const size_t bn9 = (size_t)max_B * max_N * 9;
        // all-pairs upper bound on work items for the exchange buffer: N*(N-1) ordered pairs, 1 item each,
        // plus slack for pairs split into several items
        // one 128-byte line (16 granules) per item; the two-hop kernel lays out (pair, side) lines (both orders of every
        // pair: < 2 N^2) + one line per workgroup + one per frame in the same buffer
        eng->xchg_granules = (size_t)(2 * max_N * max_N + 512) * 16;
#define PD_ALLOC(ptr, bytes)                                             \
    if (hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) {             \
        pd_set_error("pd_engine_create: hipMalloc of %zu B failed", (size_t)(bytes)); \
        rc = PD_ERR_HIP;                                                 \
        break;                                                           \
    }
        PD_ALLOC(eng->d_seqs, sizeof(PdSeqDesc) * max_B);
        PD_ALLOC(eng->d_xchg, sizeof(unsigned long long) * 2 * eng->xchg_granules * max_B);
        PD_ALLOC(eng->d_err, 256);
        PD_ALLOC(eng->d_stamps, sizeof(unsigned long long) * 2 * PD_STAMP_SLOTS);
        PD_ALLOC(eng->d_z, sizeof(float) * max_B * max_N * w->z_dim);
        PD_ALLOC(eng->d_noise, sizeof(float) * (T + 1) * bn9);
        PD_ALLOC(eng->d_process, sizeof(float) * (T + 1) * bn9);
        PD_ALLOC(eng->d_mean, sizeof(float) * bn9);
        PD_ALLOC(eng->d_stats, sizeof(float) * (size_t)T * max_B * 5 * 4);
#undef PD_ALLOC
        if (hipMemset(eng->d_seqs, 0, sizeof(PdSeqDesc) * max_B) != hipSuccess ||
            hipMemset(eng->d_err, 0, 256) != hipSuccess || hipMemset(eng->d_stamps, 0, sizeof(unsigned long long) * 2 * PD_STAMP_SLOTS) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            pd_set_error("pd_engine_create: hipMemset failed");
            rc = PD_ERR_HIP;
            break;
        }
    } while (0);
    if (rc) {
        pd_engine_destroy(eng);
        return rc;
    }
    *out = eng;
    return PD_OK;
}

// ---- step-level API -----------------------------------------------------------------------------
extern "C" int pd_denoise_step(pd_engine *eng, const float *x, const float *z, int t, int B, int N, float *eps_out,
                               void *stream) {
    if (!eng || !eps_out) {
        pd_set_error("pd_denoise_step: NULL argument");
        return PD_ERR_INVALID_ARG;
    }
    return pd_denoiser_launch(eng, x, z, t, B, N, eps_out, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int pd_p_mean(pd_engine *eng, const float *x, const float *z, int t, int B, int N, float *mean_out,
                         float *x0_out, void *stream) {
    if (!eng || !mean_out) {
        pd_set_error("pd_p_mean: NULL argument");
        return PD_ERR_INVALID_ARG;
    }
    return pd_denoiser_launch(eng, x, z, t, B, N, nullptr, mean_out, x0_out, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int pd_p_finish(pd_engine *eng, const float *mean, const float *noise, int t, int B, int N, float *x_out,
                           void *stream) {
    if (!eng || !mean || !x_out || B <= 0 || N <= 0 || t < 0 || t >= eng->timesteps) {
        pd_set_error("pd_p_finish: invalid arguments");
        return PD_ERR_INVALID_ARG;
    }
    const int n = B * N * 9;
    hipLaunchKernelGGL(pd_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, noise,
                       expf(0.5f * eng->logvar[t]), n, x_out);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

extern "C" int pd_pose_to_camera_ex(pd_engine *eng, const float *enc, int n_cameras, float *R_out, float *T_out,
                                    float *focal_out, float log_focal_length_bias, float min_focal_length, float max_focal_length,
                                    void *stream) {
    if (!eng || !enc || !R_out || !T_out || !focal_out || n_cameras <= 0) {
        pd_set_error("pd_pose_to_camera: invalid arguments");
        return PD_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(pd_camera_kernel, dim3((n_cameras + 63) / 64), dim3(64), 0, (hipStream_t)stream, enc, n_cameras,
                       R_out, T_out, focal_out, log_focal_length_bias, min_focal_length, max_focal_length);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
extern "C" int pd_pose_to_camera(pd_engine *eng, const float *enc, int n_cameras, float *R_out, float *T_out,
                                 float *focal_out, void *stream) {
    return pd_pose_to_camera_ex(eng, enc, n_cameras, R_out, T_out, focal_out, 1.8f, 0.1f, 20.0f, stream);
}

// ---- GGS API ------------------------------------------------------------------------------------
static void guide_stages(const pd_ggs_cfg *cfg, PdGgsStage *st) {
    // geometry_guided_sampling.py:48-63: all (2x iters, :86-87), FL, R, T, all
    const int it = cfg->iter_num;
    st[0] = {1, 1, 1, 2 * it};
    st[1] = {0, 0, 1, it};
    st[2] = {1, 0, 0, it};
    st[3] = {0, 1, 0, it};
    st[4] = {1, 1, 1, 2 * it};
}

extern "C" int pd_ggs_stage_iters(const pd_ggs_cfg *cfg, int *iters_out5) {
    if (!cfg || !iters_out5 || cfg->iter_num < 0) {
        pd_set_error("pd_ggs_stage_iters: invalid arguments");
        return PD_ERR_INVALID_ARG;
    }
    PdGgsStage st[PD_GGS_MAX_STAGES];
    guide_stages(cfg, st);
    for (int i = 0; i < 5; ++i) iters_out5[i] = st[i].iters;
    return PD_OK;
}

static int check_cfg(const pd_ggs_cfg *cfg, const char *who) {
    if (!cfg || cfg->iter_num < 0 || !(cfg->learning_rate > 0.0f)) {
        pd_set_error("%s: invalid pd_ggs_cfg", who);
        return PD_ERR_INVALID_ARG;
    }
    return PD_OK;
}

extern "C" int pd_ggs_guide(pd_engine *eng, float *model_mean, int B, int N, int t, const pd_ggs_cfg *cfg,
                            float *stats_out, void *stream) {
    (void)t;   // only printed by the reference (:124)
    int rc = check_cfg(cfg, "pd_ggs_guide");
    if (rc) return rc;
    PdGgsStage st[5];
    guide_stages(cfg, st);
    return pd_ggs_launch(eng, model_mean, B, N, st, 5, cfg, 0, stats_out, nullptr, 0, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int pd_ggs_optimize(pd_engine *eng, float *model_mean, int B, int N, int update_R, int update_T, int update_FL,
                               const pd_ggs_cfg *cfg, float *stats_out, float *trace_out, void *stream) {
    int rc = check_cfg(cfg, "pd_ggs_optimize");
    if (rc) return rc;
    PdGgsStage st = {update_R ? 1 : 0, update_T ? 1 : 0, update_FL ? 1 : 0, cfg->iter_num};
    if (update_R && update_T && update_FL) st.iters *= 2;   // :86-87
    return pd_ggs_launch(eng, model_mean, B, N, &st, 1, cfg, 0, stats_out, trace_out, st.iters, nullptr, nullptr,
                         (hipStream_t)stream);
}

extern "C" int pd_ggs_loss_grad(pd_engine *eng, const float *x, int B, int N, int update_R, int update_T, int update_FL,
                                const pd_ggs_cfg *cfg, float *loss_out, float *grad_out, void *stream) {
    int rc = check_cfg(cfg, "pd_ggs_loss_grad");
    if (rc) return rc;
    if (!loss_out || !grad_out) {
        pd_set_error("pd_ggs_loss_grad: NULL output");
        return PD_ERR_INVALID_ARG;
    }
    PdGgsStage st = {update_R ? 1 : 0, update_T ? 1 : 0, update_FL ? 1 : 0, 1};
    return pd_ggs_launch(eng, const_cast<float *>(x), B, N, &st, 1, cfg, 1, nullptr, nullptr, 0, loss_out, grad_out,
                         (hipStream_t)stream);
}

// ---- whole sampler ------------------------------------------------------------------------------
// Issues every launch of p_sample_loop on `s`, reading/writing only engine-owned buffers (so the
// same code can be captured into a graph and replayed).
static int issue_loop(pd_engine *eng, int B, int N, int cond_start, const pd_ggs_cfg *ggs, bool want_stats, int step_begin,
                      int step_end, hipStream_t s) {
    const int T = eng->timesteps;
    const size_t bn9 = (size_t)B * N * 9;
    float *proc = eng->d_process;
    // the z piece of _first once per call (z is the same in every step, models/denoiser.py:56-70); the guided half of a split call
    // recomputes it (one GEMM) instead of relying on what the engine's buffer held last
    int rc0 = pd_denoiser_prepare(eng, eng->d_z, B, N, s);
    if (rc0) return rc0;
    for (int step = step_begin; step < step_end; ++step) {
        const int t = T - 1 - step;                               // reversed(range(T))  :296
        const float *x = proc + (size_t)step * bn9;
        float *xn = proc + (size_t)(step + 1) * bn9;
        const bool guided = ggs && t < cond_start;                // :270
        int rc;
        if (guided) {
            // mean -> next slot, GGS refines it in place, noise = 0  (:272-276)
            rc = pd_denoiser_launch(eng, x, eng->d_z, t, B, N, nullptr, nullptr, nullptr, nullptr, xn, s, true);
            if (rc) return rc;
            float *st = want_stats ? eng->d_stats + (size_t)(cond_start - 1 - t) * B * 5 * 4 : nullptr;
            eng->stamp_slot = (cond_start - 1 - t) % PD_STAMP_SLOTS;     // this guided step's launch stamps (baked into a captured node like `st`)
            rc = pd_ggs_guide(eng, xn, B, N, t, ggs, st, s);
            eng->stamp_slot = 0;
        } else {
            const float *nz = (t > 0) ? eng->d_noise + (size_t)(step + 1) * bn9 : nullptr;   // :278
            rc = pd_denoiser_launch(eng, x, eng->d_z, t, B, N, nullptr, nullptr, nullptr, nz, xn, s, true);
        }
        if (rc) return rc;
    }
    return PD_OK;
}

static bool same_cfg(const pd_ggs_cfg &a, const pd_ggs_cfg &b) { return memcmp(&a, &b, sizeof(a)) == 0; }

extern "C" int pd_sample_phase(pd_engine *eng, const float *z, const float *noise, int B, int N, int cond_start_step,
                               const pd_ggs_cfg *ggs, int phase, float *pose_out, float *process_out, float *stats_out,
                               int use_graph, void *stream) {
    if (!eng || !z || !noise || !pose_out || B <= 0 || B > eng->max_B || N <= 0 || N > eng->max_N ||
        cond_start_step < 0 || cond_start_step > eng->timesteps || phase < PD_PHASE_ALL || phase > PD_PHASE_GUIDED) {
        pd_set_error("pd_sample: invalid arguments (B=%d N=%d cond_start_step=%d phase=%d)", B, N, cond_start_step, phase);
        return PD_ERR_INVALID_ARG;
    }
    if (ggs) {
        int rc = check_cfg(ggs, "pd_sample");
        if (rc) return rc;
        if (cond_start_step > 0) {
            for (int b = 0; b < B; ++b)
                if (!eng->seqs[b].blob) {
                    pd_set_error("pd_sample: GGS requested but sequence slot %d has no matches", b);
                    return PD_ERR_STATE;
                }
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const int T = eng->timesteps;
    const size_t bn9 = (size_t)B * N * 9;
    const bool has_ggs = ggs && cond_start_step > 0;
    const bool want_stats = has_ggs && stats_out;
    // the unguided steps come first (t = T-1 .. cond_start), the guided ones last (t < cond_start, :270)
    const int split = has_ggs ? T - cond_start_step : T;
    const int step_begin = phase == PD_PHASE_GUIDED ? split : 0;
    const int step_end = phase == PD_PHASE_UNGUIDED ? split : T;
    if (phase != PD_PHASE_GUIDED) {
        PD_HIP_CHECK(hipMemcpyAsync(eng->d_z, z, sizeof(float) * B * N * eng->z_dim, hipMemcpyDeviceToDevice, s));
        PD_HIP_CHECK(hipMemcpyAsync(eng->d_noise, noise, sizeof(float) * (T + 1) * bn9, hipMemcpyDeviceToDevice, s));
        PD_HIP_CHECK(hipMemcpyAsync(eng->d_process, noise, sizeof(float) * bn9, hipMemcpyDeviceToDevice, s));   // :289
    }
    if (has_ggs && phase != PD_PHASE_UNGUIDED) {
        int rc = pd_wait_uploads(eng, s);     // asynchronous match uploads issued on another stream
        if (rc) return rc;
    }
    if (step_begin == step_end) {
        // nothing to run in this phase
    } else if (!use_graph) {
        int rc = issue_loop(eng, B, N, cond_start_step, has_ggs ? ggs : nullptr, true, step_begin, step_end, s);
        if (rc) return rc;
    } else {
        pd_engine::GraphKey key;
        memset(&key, 0, sizeof(key));
        key.B = B;
        key.N = N;
        key.cond_start = has_ggs ? cond_start_step : 0;
        key.has_ggs = has_ggs;
        key.phase = phase;
        key.den_split = eng->den_split | (eng->den_fused_attn << 8);      // the options change the captured launches
        if (has_ggs) {
            key.cfg = *ggs;
            // the GGS nodes bake the match-derived launch shape in: a re-upload with another item count must not
            // replay them (and a replay must not skip pd_ggs_launch's state / frame-count checks)
            if (phase != PD_PHASE_UNGUIDED) {
                int rc = pd_ggs_plan(eng, B, N, ggs, &key.plan);
                if (rc) return rc;
            }
        }
        hipGraphExec_t exec = nullptr;
        for (size_t gi = 0; gi < eng->graphs.size(); ++gi) {
            auto &g = eng->graphs[gi];
            if (g.first.B == B && g.first.N == N && g.first.cond_start == key.cond_start && g.first.has_ggs == key.has_ggs &&
                g.first.phase == phase && g.first.den_split == key.den_split && same_cfg(g.first.cfg, key.cfg) &&
                memcmp(&g.first.plan, &key.plan, sizeof(PdGgsPlan)) == 0) {
                exec = g.second;
                std::rotate(eng->graphs.begin() + gi, eng->graphs.begin() + gi + 1, eng->graphs.end());   // most recently used last
                break;
            }
        }
        if (!exec) {
            // capture on a private stream so the caller's stream state is untouched
            if (!eng->own_stream) PD_HIP_CHECK(hipStreamCreateWithFlags(&eng->own_stream, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            PD_HIP_CHECK(hipStreamBeginCapture(eng->own_stream, hipStreamCaptureModeThreadLocal));
            int rc = issue_loop(eng, B, N, cond_start_step, has_ggs ? ggs : nullptr, true, step_begin, step_end, eng->own_stream);
            hipError_t ce = hipStreamEndCapture(eng->own_stream, &graph);
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            PD_HIP_CHECK(ce);
            PD_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            PD_HIP_CHECK(hipGraphDestroy(graph));
            // bounded cache: a service whose match sets keep changing the launch plan must not pile up instantiated graphs.  The
            // least recently used one goes; it may still be replaying on some stream, so this engine's work is drained first (rare)
            if (eng->graphs.size() >= PD_GRAPH_CACHE_MAX) {
                PD_HIP_CHECK(hipDeviceSynchronize());
                (void)hipGraphExecDestroy(eng->graphs.front().second);
                eng->graphs.erase(eng->graphs.begin());
            }
            eng->graphs.push_back({key, exec});
        }
        PD_HIP_CHECK(hipGraphLaunch(exec, s));
        if (has_ggs) {   // (eager launches mark their use themselves; a replay does not pass through pd_ggs_launch)
            int rc = pd_mark_use(eng, s);
            if (rc) return rc;
        }
    }
    if (phase == PD_PHASE_UNGUIDED) return PD_OK;   // results are copied out by the guided phase
    PD_HIP_CHECK(hipMemcpyAsync(pose_out, eng->d_process + (size_t)T * bn9, sizeof(float) * bn9, hipMemcpyDeviceToDevice, s));
    if (process_out)
        PD_HIP_CHECK(hipMemcpyAsync(process_out, eng->d_process, sizeof(float) * (T + 1) * bn9, hipMemcpyDeviceToDevice, s));
    if (want_stats)
        PD_HIP_CHECK(hipMemcpyAsync(stats_out, eng->d_stats, sizeof(float) * (size_t)cond_start_step * B * 5 * 4,
                                    hipMemcpyDeviceToDevice, s));
    return PD_OK;
}

extern "C" int pd_sample(pd_engine *eng, const float *z, const float *noise, int B, int N, int cond_start_step,
                         const pd_ggs_cfg *ggs, float *pose_out, float *process_out, float *stats_out, int use_graph,
                         void *stream) {
    return pd_sample_phase(eng, z, noise, B, N, cond_start_step, ggs, PD_PHASE_ALL, pose_out, process_out, stats_out, use_graph,
                           stream);
}

// ---- options ------------------------------------------------------------------------------------
extern "C" int pd_engine_set_option(pd_engine *eng, int option, int value) {
    if (!eng) {
        pd_set_error("pd_engine_set_option: NULL engine");
        return PD_ERR_INVALID_ARG;
    }
    switch (option) {
    case PD_OPT_DENOISER_SPLIT:
        if (value < 0 || value > 2) {
            pd_set_error("pd_engine_set_option: PD_OPT_DENOISER_SPLIT takes 0, 1 or 2 (got %d)", value);
            return PD_ERR_INVALID_ARG;
        }
        if (value) {
            PD_HIP_CHECK(hipSetDevice(eng->device));
            int rc = pd_denoiser_build_split(eng, value);
            if (rc) return rc;
        }
        eng->den_split = value;
        break;
    case PD_OPT_DENOISER_FUSED_ATTN:
        if (value < 0 || value > 2) {
            pd_set_error("pd_engine_set_option: PD_OPT_DENOISER_FUSED_ATTN takes 0, 1 or 2 (got %d)", value);
            return PD_ERR_INVALID_ARG;
        }
        eng->den_fused_attn = value;
        break;
    case 3:      // (PD_OPT_DENOISER_PERSISTENT of round 3: the persistent small-batch kernel was measured 2.4 x slower and parked, tools/parked/)
        if (value == 0) break;
        pd_set_error("pd_engine_set_option: option 3 (the persistent small-batch denoiser launch of round 3) is no longer built: it measured "
                     "2.4 x slower than the multi-launch path (profiles/round3_small_persistent.txt; source parked under tools/parked/)");
        return PD_ERR_UNSUPPORTED;
    default:
        pd_set_error("pd_engine_set_option: unknown option %d", option);
        return PD_ERR_INVALID_ARG;
    }
    return PD_OK;      // (captured graphs are keyed on the option: nothing to drop)
}

extern "C" int pd_engine_get_option(pd_engine *eng, int option, int *value_out) {
    if (!eng || !value_out) {
        pd_set_error("pd_engine_get_option: NULL argument");
        return PD_ERR_INVALID_ARG;
    }
    switch (option) {
    case PD_OPT_DENOISER_SPLIT: *value_out = eng->den_split; break;
    case PD_OPT_DENOISER_FUSED_ATTN: *value_out = eng->den_fused_attn; break;
    case PD_OPT_WEIGHTS_NON_FINITE: *value_out = pd_denoiser_weights_non_finite(eng) ? 1 : 0; break;
    default:
        pd_set_error("pd_engine_get_option: unknown option %d", option);
        return PD_ERR_INVALID_ARG;
    }
    return PD_OK;
}

// ---- measurement helper -------------------------------------------------------------------------
extern "C" int pd_time_kernel(pd_engine *eng, int what, int B, int N, const pd_ggs_cfg *cfg, int reps, float *ms_out,
                              void *stream) {
    if (!eng || !ms_out || reps <= 0 || B <= 0 || B > eng->max_B || N <= 0 || N > eng->max_N) {
        pd_set_error("pd_time_kernel: invalid arguments");
        return PD_ERR_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    PD_HIP_CHECK(hipEventCreate(&e0));
    PD_HIP_CHECK(hipEventCreate(&e1));
    const size_t bn9 = (size_t)B * N * 9;
    int rc = PD_OK;
    // inputs: whatever the sampler buffers currently hold (process slot 0 = a pose sample, d_z)
    if (what == 0) {
        // a step as the sampling loop runs it: the z piece of _first prepared once, outside the timed launches
        rc = pd_denoiser_launch(eng, eng->d_process, eng->d_z, 50, B, N, nullptr, eng->d_mean, nullptr, nullptr, nullptr, s, false);
        PD_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps && !rc; ++i)
            rc = pd_denoiser_launch(eng, eng->d_process, eng->d_z, 50, B, N, nullptr, eng->d_mean, nullptr, nullptr, nullptr, s, true);
        PD_HIP_CHECK(hipEventRecord(e1, s));
    } else {
        if ((rc = check_cfg(cfg, "pd_time_kernel"))) return rc;
        // GGS on a scratch copy of the final pose of the last pd_sample (slot T), so timing does not drift
        PD_HIP_CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < reps && !rc; ++i) {
            PD_HIP_CHECK(hipMemcpyAsync(eng->d_mean, eng->d_process + (size_t)eng->timesteps * bn9, sizeof(float) * bn9,
                                        hipMemcpyDeviceToDevice, s));
            rc = pd_ggs_guide(eng, eng->d_mean, B, N, 0, cfg, nullptr, s);
        }
        PD_HIP_CHECK(hipEventRecord(e1, s));
    }
    if (rc) return rc;
    PD_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    PD_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PD_OK;
}

// async error word of the GGS exchange (nonzero = a bounded spin gave up); synchronises the device
extern "C" int pd_check_async_error(pd_engine *eng) {
    if (!eng) return PD_ERR_INVALID_ARG;
    unsigned int v = 0;
    PD_HIP_CHECK(hipDeviceSynchronize());
    PD_HIP_CHECK(hipMemcpy(&v, eng->d_err, sizeof(v), hipMemcpyDeviceToHost));
    if (v) {
        pd_set_error("asynchronous error (flag=%u):%s%s%s", v,
                     (v & 1u) ? " a cross-workgroup exchange spin timed out (co-resident workgroups lost?);" : "",
                     (v & 2u) ? " pd_ggs_set_matches_csr_async met a frame index outside [0, n_frames);" : "",
                     (v & 4u) ? " pd_ggs_set_matches_csr_async: pd_match_hints violated (more pairs / matches per pair than declared): the slot was emptied;" : "");
        (void)hipMemset(eng->d_err, 0, sizeof(v));
        return PD_ERR_STATE;
    }
    return PD_OK;
}

// debug: enable/read the GGS kernel's phase cycle counters (P1, P2, exchange, P3, P4, iterations)
// of workgroup 0.  out6 may be NULL to just switch collection on/off.
extern "C" int pd_debug_ggs_prof(pd_engine *eng, int enable, long long *out6) {
    if (!eng) return PD_ERR_INVALID_ARG;
    eng->ggs_prof_on = enable;
    if (out6) {
        PD_HIP_CHECK(hipDeviceSynchronize());
        PD_HIP_CHECK(hipMemcpy(out6, eng->d_err + 2, sizeof(long long) * 16, hipMemcpyDeviceToHost));
    }
    return PD_OK;
}

__global__ void pd_copy_stamps_kernel(const unsigned long long *__restrict__ src, long long *__restrict__ dst, int n2) {
    const int i = threadIdx.x;
    if (i < n2) dst[i] = (long long)src[i];
}
extern "C" int pd_ggs_launch_stamps(pd_engine *eng, long long *dst, int n, int *clock_khz_out, void *stream) {
    if (!eng || !dst || n <= 0 || n > PD_STAMP_SLOTS) {
        pd_set_error("pd_ggs_launch_stamps: invalid arguments (n=%d, at most %d slots)", n, PD_STAMP_SLOTS);
        return PD_ERR_INVALID_ARG;
    }
    if (clock_khz_out) {
        static int khz = 0;          // one query per process: the attribute call is not free and bench.py asks once per pass
        if (khz == 0) PD_HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, eng->device));
        *clock_khz_out = khz;
    }
    // a one-workgroup kernel into the caller's (preallocated) rows, not hipMemcpyAsync into a fresh tensor: that form of the readout cost bench.py's
    // pipe 40 % (653 instead of 1 080 sequences/s, profiles/round6_launch_stamps_ab.txt); a launch is ordered on its stream and nothing else
    hipLaunchKernelGGL(pd_copy_stamps_kernel, dim3(1), dim3(2 * PD_STAMP_SLOTS), 0, (hipStream_t)stream, eng->d_stamps, dst, 2 * n);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

extern "C" int pd_debug_lane_tables(pd_engine *eng, int seq, int *out, int n_out) {
    if (!eng || !out || seq < 0 || seq >= eng->max_B || n_out < 4) {
        pd_set_error("pd_debug_lane_tables: invalid arguments (seq=%d n_out=%d)", seq, n_out);
        return PD_ERR_INVALID_ARG;
    }
    // the DEVICE descriptor (for slots filled by pd_ggs_set_matches_csr_async the host shadow holds capacities, not counts); synchronous
    PD_HIP_CHECK(hipDeviceSynchronize());
    PdSeqDesc D;
    PD_HIP_CHECK(hipMemcpy(&D, eng->d_seqs + seq, sizeof(D), hipMemcpyDeviceToHost));
    for (int i = 0; i < n_out; ++i) out[i] = 0;
    out[0] = D.n_litems;
    out[1] = D.n_lwaves;
    out[2] = D.l_item_len;
    out[3] = D.l_max_steps;
    if (D.n_lwaves > 0 && D.lwave) {
        int2 lw[PD_LANE_WAVES];
        const int nw = D.n_lwaves < PD_LANE_WAVES ? D.n_lwaves : PD_LANE_WAVES;
        PD_HIP_CHECK(hipMemcpy(lw, D.lwave, sizeof(int2) * nw, hipMemcpyDeviceToHost));
        for (int w = 0; w < nw && 4 + w < n_out; ++w) out[4 + w] = lw[w].y;
    }
    return PD_OK;
}

extern "C" int pd_debug_ggs_plan(pd_engine *eng, int B, int N, const pd_ggs_cfg *cfg, int *out8) {
    if (!eng || !cfg || !out8) return PD_ERR_INVALID_ARG;
    PdGgsPlan plan;
    int rc = pd_ggs_plan(eng, B, N, cfg, &plan);
    if (rc) return rc;
    const int v[8] = {plan.k, plan.n_slots, plan.lds, plan.two_hop, plan.waves, plan.stage_p, plan.lane, plan.lane_rl};
    for (int i = 0; i < 8; ++i) out8[i] = v[i];
    return PD_OK;
}

