// pd_metrics.hip -- evaluation metrics and image preprocessing (row N4) on the device.
//
// Replaces (paths relative to /root/reference/pose_diffusion/):
//   util/metric.py:14-47     camera_to_rel_deg: pairwise relative poses of all i < j per sequence, rotation angle and
//                            translation-direction angle between prediction and ground truth (:106-172)
//   util/metric.py:50-78     calculate_auc_np; test.py:113-121 the Racc / Tacc thresholds
//   util/metric.py:174-185   compute_ARE
//   demo.py:127-129          pytorch3d.ops.corresponding_cameras_alignment(estimate_scale=True, mode="extrinsics")
// pytorch3d pieces (absent from the reference tree, restated from the published 0.7.x algorithms, see
// oracle/pd_oracle.py): get_world_to_view_transform().get_matrix() = [[R, 0], [T, 1]] (row vectors),
// so3_relative_angle with acos_linear_extrapolation, the extrinsics-mode similarity alignment.
// These are tiny, latency-bound kernels (tens of cameras); they exist so that demo.py / test.py style evaluation needs
// neither pytorch3d nor a host round trip.  Stateless: no engine handle.
#include "pd_internal.h"

#include <math.h>

#define PD_RAD2DEG 57.29577951308232f

// angle (radians) from cos with pytorch3d's acos_linear_extrapolation, bounds (-(1 - 1e-4), 1 - 1e-4)
__device__ __forceinline__ float pd_acos_extrap(float x) {
    const float b = 1.0f - 1e-4f;
    if (x >= b) return (x - b) * (-1.0f / sqrtf(1.0f - b * b)) + acosf(b);
    if (x <= -b) return (x + b) * (-1.0f / sqrtf(1.0f - b * b)) + acosf(-b);
    return acosf(x);
}

// C = A^T B (3x3, row-major)
__device__ __forceinline__ void pd_atb(const float *A, const float *B, float *C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[0 * 3 + r] * B[0 * 3 + c] + A[1 * 3 + r] * B[1 * 3 + c] + A[2 * 3 + r] * B[2 * 3 + c];
}

// relative pose of cameras (R1, T1) -> (R2, T2) in the row-vector world-to-view convention:
// inverse(se3_1) @ se3_2 = [[R1^T R2, 0], [T2 - T1 R1^T R2, 1]]   (metric.py:39-40, :114-140)
__device__ __forceinline__ void pd_rel_pose(const float *R1, const float *T1, const float *R2, const float *T2, float *Rr, float *tr) {
    pd_atb(R1, R2, Rr);
#pragma unroll
    for (int c = 0; c < 3; ++c) tr[c] = T2[c] - (T1[0] * Rr[0 * 3 + c] + T1[1] * Rr[1 * 3 + c] + T1[2] * Rr[2 * 3 + c]);
}

__global__ void pd_metrics_pairs_kernel(const float *__restrict__ Rp, const float *__restrict__ Tp, const float *__restrict__ Rg,
                                        const float *__restrict__ Tg, int B, int N, float *__restrict__ r_deg,
                                        float *__restrict__ t_deg) {
    const int P = N * (N - 1) / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * P) return;
    const int b = idx / P;
    int p = idx - b * P, i = 0;
    while (p >= N - 1 - i) {      // torch.combinations order: (0,1), (0,2), ..., (1,2), ...   (metric.py:108)
        p -= N - 1 - i;
        ++i;
    }
    const int j = i + 1 + p;
    const int c1 = b * N + i, c2 = b * N + j;
    float Rgr[9], tgr[3], Rpr[9], tpr[3];
    pd_rel_pose(Rg + c1 * 9, Tg + c1 * 3, Rg + c2 * 9, Tg + c2 * 3, Rgr, tgr);
    pd_rel_pose(Rp + c1 * 9, Tp + c1 * 3, Rp + c2 * 9, Tp + c2 * 3, Rpr, tpr);
    // so3_relative_angle(rot_gt, rot_pred): trace of rot_gt rot_pred^T   (:143-151)
    float tr = 0.0f;
#pragma unroll
    for (int q = 0; q < 9; ++q) tr += Rgr[q] * Rpr[q];
    r_deg[idx] = pd_acos_extrap((tr - 1.0f) * 0.5f) * PD_RAD2DEG;
    // compare_translation_by_angle (:163-172)
    const float eps = 1e-15f;
    const float np_ = sqrtf(tpr[0] * tpr[0] + tpr[1] * tpr[1] + tpr[2] * tpr[2]) + eps;
    const float ng = sqrtf(tgr[0] * tgr[0] + tgr[1] * tgr[1] + tgr[2] * tgr[2]) + eps;
    const float d = (tpr[0] / np_) * (tgr[0] / ng) + (tpr[1] / np_) * (tgr[1] / ng) + (tpr[2] / np_) * (tgr[2] / ng);
    const float loss = fmaxf(1.0f - d * d, eps);
    float e = acosf(sqrtf(1.0f - loss));
    if (isnan(e) || isinf(e)) e = 1e6f;
    t_deg[idx] = e * PD_RAD2DEG;
}

// out[0] = Auc_30 (calculate_auc_np with max_threshold), out[1..3] = Racc_5/15/30, out[4..6] = Tacc_5/15/30 (percent)
__global__ __launch_bounds__(256) void pd_metrics_summary_kernel(const float *__restrict__ r, const float *__restrict__ t, int n,
                                                                 int max_threshold, float *__restrict__ out) {
    __shared__ unsigned hist[64];
    __shared__ unsigned cnt[6];
    const int tid = threadIdx.x;
    if (tid < 64) hist[tid] = 0;
    if (tid < 6) cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const float rv = r[i], tv = t[i];
        const float m = fmaxf(rv, tv);                           // np.max over (r, t)   :64-67
        // np.histogram(bins = arange(max_threshold + 1)): unit bins, the last one closed on the right   :70-73
        if (m >= 0.0f && m <= (float)max_threshold) {
            int bin = (int)floorf(m);
            if (bin >= max_threshold) bin = max_threshold - 1;
            atomicAdd(&hist[bin], 1u);
        }
        if (rv < 5.0f) atomicAdd(&cnt[0], 1u);
        if (rv < 15.0f) atomicAdd(&cnt[1], 1u);
        if (rv < 30.0f) atomicAdd(&cnt[2], 1u);
        if (tv < 5.0f) atomicAdd(&cnt[3], 1u);
        if (tv < 15.0f) atomicAdd(&cnt[4], 1u);
        if (tv < 30.0f) atomicAdd(&cnt[5], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        double cum = 0.0, acc = 0.0;
        for (int k = 0; k < max_threshold; ++k) {
            cum += (double)hist[k] / (double)n;                  // normalised histogram, cumulative sum   :76-78
            acc += cum;
        }
        out[0] = (float)(acc / (double)max_threshold);
        for (int k = 0; k < 6; ++k) out[1 + k] = 100.0f * (float)cnt[k] / (float)n;   // np.mean(err < thr) * 100, test.py:113-119
    }
}

__global__ void pd_metrics_are_kernel(const float *__restrict__ Ra, const float *__restrict__ Rb, int n, float *__restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float tr = 0.0f;                                             // trace(Ra^T Rb) = sum of the elementwise product
#pragma unroll
    for (int q = 0; q < 9; ++q) tr += Ra[i * 9 + q] * Rb[i * 9 + q];
    const float c = fminf(fmaxf((tr - 1.0f) * 0.5f, -1.0f), 1.0f);
    err[i] = acosf(c) * PD_RAD2DEG;
}

// one-sided Jacobi SVD of a 3x3 matrix in double: A = U diag(s) V^T (U, V orthogonal; adequate for the near-rotation
// covariance of the alignment).  Single thread.
__device__ void pd_svd3(const double *A, double *U, double *V) {
    double G[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int q = 0; q < 9; ++q) G[q] = A[q];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;                      // columns p, q of G
                for (int r = 0; r < 3; ++r) {
                    a += G[r * 3 + p] * G[r * 3 + p];
                    b += G[r * 3 + q] * G[r * 3 + q];
                    c += G[r * 3 + p] * G[r * 3 + q];
                }
                off += c * c;
                if (fabs(c) < 1e-300) continue;
                const double zeta = (b - a) / (2.0 * c);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int r = 0; r < 3; ++r) {
                    const double gp = G[r * 3 + p], gq = G[r * 3 + q];
                    G[r * 3 + p] = cs * gp - sn * gq;
                    G[r * 3 + q] = sn * gp + cs * gq;
                    const double wp = W[r * 3 + p], wq = W[r * 3 + q];
                    W[r * 3 + p] = cs * wp - sn * wq;
                    W[r * 3 + q] = sn * wp + cs * wq;
                }
            }
        if (off < 1e-30) break;
    }
    for (int c = 0; c < 3; ++c) {                                // G = U diag(s): normalise the columns
        double s = sqrt(G[0 * 3 + c] * G[0 * 3 + c] + G[1 * 3 + c] * G[1 * 3 + c] + G[2 * 3 + c] * G[2 * 3 + c]);
        for (int r = 0; r < 3; ++r) {
            U[r * 3 + c] = s > 0 ? G[r * 3 + c] / s : (r == c ? 1.0 : 0.0);
            V[r * 3 + c] = W[r * 3 + c];
        }
    }
}

// corresponding_cameras_alignment, mode "extrinsics": one workgroup; thread 0 does the 3x3 algebra
__global__ __launch_bounds__(64) void pd_align_kernel(const float *__restrict__ Rs, const float *__restrict__ Ts,
                                                      const float *__restrict__ Rt, const float *__restrict__ Tt, int n,
                                                      int estimate_scale, float eps, float *__restrict__ Ro,
                                                      float *__restrict__ To, float *__restrict__ srt) {
    __shared__ double RA[9], TA[3], S;
    if (threadIdx.x == 0) {
        double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Amu[3] = {0, 0, 0}, Bmu[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const float *a = Rs + i * 9, *b = Rt + i * 9;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)                      // R_src R_tgt^T
                    cov[r * 3 + c] += (double)a[r * 3 + 0] * b[c * 3 + 0] + (double)a[r * 3 + 1] * b[c * 3 + 1] + (double)a[r * 3 + 2] * b[c * 3 + 2];
            for (int r = 0; r < 3; ++r) {                        // A_i = R_src T_src, B_i = R_src T_tgt (column products)
                Amu[r] += (double)a[r * 3 + 0] * Ts[i * 3 + 0] + (double)a[r * 3 + 1] * Ts[i * 3 + 1] + (double)a[r * 3 + 2] * Ts[i * 3 + 2];
                Bmu[r] += (double)a[r * 3 + 0] * Tt[i * 3 + 0] + (double)a[r * 3 + 1] * Tt[i * 3 + 1] + (double)a[r * 3 + 2] * Tt[i * 3 + 2];
            }
        }
        for (int q = 0; q < 9; ++q) cov[q] /= n;
        for (int r = 0; r < 3; ++r) {
            Amu[r] /= n;
            Bmu[r] /= n;
        }
        double U[9], V[9];
        pd_svd3(cov, U, V);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) RA[r * 3 + c] = V[r * 3 + 0] * U[c * 3 + 0] + V[r * 3 + 1] * U[c * 3 + 1] + V[r * 3 + 2] * U[c * 3 + 2];   // V U^T
        double s = 1.0;
        if (estimate_scale && n > 1) {
            double num = 0.0, den = 0.0;
            for (int i = 0; i < n; ++i) {
                const float *a = Rs + i * 9;
                for (int r = 0; r < 3; ++r) {
                    const double Ai = (double)a[r * 3 + 0] * Ts[i * 3 + 0] + (double)a[r * 3 + 1] * Ts[i * 3 + 1] + (double)a[r * 3 + 2] * Ts[i * 3 + 2] - Amu[r];
                    const double Bi = (double)a[r * 3 + 0] * Tt[i * 3 + 0] + (double)a[r * 3 + 1] * Tt[i * 3 + 1] + (double)a[r * 3 + 2] * Tt[i * 3 + 2] - Bmu[r];
                    num += Ai * Bi;
                    den += Ai * Ai;
                }
            }
            num /= 3.0 * n;                                      // .mean() over all n x 3 entries
            den /= 3.0 * n;
            s = num / (den > (double)eps ? den : (double)eps);  // .clamp(eps)
        }
        for (int r = 0; r < 3; ++r) TA[r] = Bmu[r] - s * Amu[r];
        S = s;
        if (srt) {
            srt[0] = (float)s;
            for (int q = 0; q < 9; ++q) srt[1 + q] = (float)RA[q];
            for (int r = 0; r < 3; ++r) srt[10 + r] = (float)TA[r];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float *a = Rs + i * 9;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)                          // R_A R_src
                Ro[i * 9 + r * 3 + c] = (float)(RA[r * 3 + 0] * a[0 * 3 + c] + RA[r * 3 + 1] * a[1 * 3 + c] + RA[r * 3 + 2] * a[2 * 3 + c]);
        for (int c = 0; c < 3; ++c)                              // T_A R_src + s T_src
            To[i * 3 + c] = (float)(TA[0] * a[0 * 3 + c] + TA[1] * a[1 * 3 + c] + TA[2] * a[2 * 3 + c] + S * Ts[i * 3 + c]);
    }
}

// ---- C-ABI ---------------------------------------------------------------------------------------
extern "C" int pd_metrics_rel_pose_errors(const float *R_pred, const float *T_pred, const float *R_gt, const float *T_gt, int B, int N,
                                          float *rel_r_deg, float *rel_t_deg, void *stream) {
    if (!R_pred || !T_pred || !R_gt || !T_gt || !rel_r_deg || !rel_t_deg || B <= 0 || N < 2) {
        pd_set_error("pd_metrics_rel_pose_errors: invalid arguments (B=%d N=%d)", B, N);
        return PD_ERR_INVALID_ARG;
    }
    const int total = B * (N * (N - 1) / 2);
    hipLaunchKernelGGL(pd_metrics_pairs_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, R_pred, T_pred, R_gt, T_gt,
                       B, N, rel_r_deg, rel_t_deg);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

extern "C" int pd_metrics_summary(const float *rel_r_deg, const float *rel_t_deg, int n, int max_threshold, float *out7, void *stream) {
    if (!rel_r_deg || !rel_t_deg || !out7 || n <= 0 || max_threshold < 1 || max_threshold > 64) {
        pd_set_error("pd_metrics_summary: invalid arguments (n=%d max_threshold=%d, 1..64)", n, max_threshold);
        return PD_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(pd_metrics_summary_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rel_r_deg, rel_t_deg, n, max_threshold, out7);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

extern "C" int pd_metrics_are(const float *R_a, const float *R_b, int n, float *err_deg, void *stream) {
    if (!R_a || !R_b || !err_deg || n <= 0) {
        pd_set_error("pd_metrics_are: invalid arguments (n=%d)", n);
        return PD_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(pd_metrics_are_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, R_a, R_b, n, err_deg);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

extern "C" int pd_align_cameras(const float *R_src, const float *T_src, const float *R_tgt, const float *T_tgt, int n,
                                int estimate_scale, float eps, float *R_out, float *T_out, float *s_R_T_out, void *stream) {
    if (!R_src || !T_src || !R_tgt || !T_tgt || !R_out || !T_out || n <= 0) {
        pd_set_error("pd_align_cameras: invalid arguments (n=%d)", n);
        return PD_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(pd_align_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, R_src, T_src, R_tgt, T_tgt, n, estimate_scale, eps,
                       R_out, T_out, s_R_T_out);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}

// ---- N4: image preprocessing (util/load_img_folder.py:15-48) ----------------------------------------
// One frame: uint8 RGB HWC -> float32 CHW in [0, 1], centre-cropped to a square (:68-73) and resized to
// image_size x image_size with torch's bilinear rule for align_corners=False (:35-40): source index
// (dst + 0.5) * (crop / image_size) - 0.5 clamped at 0, neighbour i1 = min(i0 + 1, crop - 1), no antialiasing.
__global__ void pd_preprocess_image_kernel(const unsigned char *__restrict__ rgb, int H, int W, int S, float *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * S) return;
    const int oy = idx / S, ox = idx - oy * S;
    const int crop = H < W ? H : W, top = (H - crop) / 2, left = (W - crop) / 2;
    const float scale = (float)crop / (float)S;
    const float sy = fmaxf(scale * ((float)oy + 0.5f) - 0.5f, 0.0f), sx = fmaxf(scale * ((float)ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < crop - 1 ? 1 : 0), x1 = x0 + (x0 < crop - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const unsigned char *p00 = rgb + ((size_t)(top + y0) * W + left + x0) * 3, *p01 = rgb + ((size_t)(top + y0) * W + left + x1) * 3;
    const unsigned char *p10 = rgb + ((size_t)(top + y1) * W + left + x0) * 3, *p11 = rgb + ((size_t)(top + y1) * W + left + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = (float)p00[c] / 255.0f, b = (float)p01[c] / 255.0f, d = (float)p10[c] / 255.0f, e = (float)p11[c] / 255.0f;   // :62
        out[((size_t)c * S + oy) * S + ox] = hy * (hx * a + lx * b) + ly * (hx * d + lx * e);
    }
}

extern "C" int pd_preprocess_image(const unsigned char *rgb_hwc, int height, int width, int image_size, float *out_chw, void *stream) {
    if (!rgb_hwc || !out_chw || height < 2 || width < 2 || image_size < 1) {
        pd_set_error("pd_preprocess_image: invalid arguments (h=%d w=%d size=%d)", height, width, image_size);
        return PD_ERR_INVALID_ARG;
    }
    const int total = image_size * image_size;
    hipLaunchKernelGGL(pd_preprocess_image_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, rgb_hwc, height, width,
                       image_size, out_chw);
    PD_HIP_CHECK(hipGetLastError());
    return PD_OK;
}
