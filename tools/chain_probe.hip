// chain_probe.hip -- what does a dependent kernel boundary cost, and can it be hidden?  A chain of NK small dependent kernels
// (each: every workgroup reads the previous kernel's whole output (16 KiB), does a little work, writes its slice) run
//   (a) in ONE stream (the launch boundary orders them),
//   (b) round-robin over S streams with NO stream dependency inside a step: kernel i is ordered behind kernel i-1 by a device-side
//       counter (producer: stores, agent-scope release fence, atomic add; consumer: spin on the counter, acquire fence), so kernel
//       i+1 is already resident -- launch latency paid, its read-only prefetch done -- when kernel i finishes,
// both captured into a hipGraph and replayed.  Prints microseconds per kernel.  hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NK = 40, WG = 96, TPB = 256, N = 4096;   // N floats = 16 KiB per buffer

template <bool CHAIN>
__global__ __launch_bounds__(TPB) void link(const float *__restrict__ w, const float *in, float *out, unsigned *cnt_prev, unsigned *cnt_me,
                                            unsigned target, int *err) {
    __shared__ float red[TPB];
    const int tid = threadIdx.x;
    float wv = w[blockIdx.x * TPB + tid];                      // "weights": independent of the chain, fetched before the wait
    if (CHAIN && cnt_prev) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(cnt_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { *err = 1; break; }   // never hang the box
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    float s = 0.0f;
    for (int i = tid; i < N; i += TPB) s += __builtin_nontemporal_load(in + i) * 0.0f + in[i];
    red[tid] = s * wv;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    for (int i = tid; i < N / WG; i += TPB) out[blockIdx.x * (N / WG) + i] = red[0] * 1e-6f + 1.0f;
    if (CHAIN) {
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(cnt_me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// (c) ONE persistent launch: the NK links as phases of one kernel, a grid-wide barrier (counter + spin + fences) between them
__global__ __launch_bounds__(TPB) void persistent(const float *__restrict__ w, float *b0, float *b1, unsigned *cnt, int *err, int fence_mode) {
    __shared__ float red[TPB];
    const int tid = threadIdx.x;
    const float wv = w[blockIdx.x * TPB + tid];
    for (int ph = 0; ph < NK; ++ph) {
        const float *in = (ph & 1) ? b1 : b0;
        float *out = (ph & 1) ? b0 : b1;
        float s = 0.0f;
        if (fence_mode == 2) { for (int i = tid; i < N; i += TPB) s += __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else for (int i = tid; i < N; i += TPB) s += in[i];
        red[tid] = s * wv;
        __syncthreads();
        for (int o = TPB / 2; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (fence_mode == 2) { for (int i = tid; i < N / WG; i += TPB) __hip_atomic_store(out + blockIdx.x * (N / WG) + i, red[0] * 1e-6f + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else for (int i = tid; i < N / WG; i += TPB) out[blockIdx.x * (N / WG) + i] = red[0] * 1e-6f + 1.0f;
        __syncthreads();
        if (tid == 0) {
            if (fence_mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(cnt + ph, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cnt + ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < WG) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { *err = 1; break; }
            }
            if (fence_mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}
__global__ void zero(unsigned *c, int n) { if (threadIdx.x < n) c[threadIdx.x] = 0; }

int main(int argc, char **argv) {
    const bool nowait = argc > 1 && argv[1][0] == 'n';
    float *w, *buf[2]; unsigned *cnt; int *err;
    CK(hipMalloc(&w, WG * TPB * 4)); CK(hipMemset(w, 0, WG * TPB * 4));
    for (auto &b : buf) { CK(hipMalloc(&b, N * 4 + 4096)); CK(hipMemset(b, 0, N * 4 + 4096)); }
    CK(hipMalloc(&cnt, 64 * 4)); CK(hipMemset(cnt, 0, 64 * 4));
    CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipStream_t st[4]; for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t fork, join[4], e0, e1; CK(hipEventCreate(&fork)); for (auto &j : join) CK(hipEventCreate(&j)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int S = 1; S <= 4; ++S) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
        if (S > 1) {
            hipLaunchKernelGGL(zero, dim3(1), dim3(64), 0, st[0], cnt, NK);
            CK(hipEventRecord(fork, st[0]));
            for (int j = 1; j < S; ++j) CK(hipStreamWaitEvent(st[j], fork, 0));
        }
        for (int i = 0; i < NK; ++i) {
            hipStream_t s = st[i % S];
            if (S == 1) hipLaunchKernelGGL(link<false>, dim3(WG), dim3(TPB), 0, s, w, buf[i & 1], buf[(i + 1) & 1], (unsigned *)nullptr, cnt, 0u, err);
            else hipLaunchKernelGGL(link<true>, dim3(WG), dim3(TPB), 0, s, w, buf[i & 1], buf[(i + 1) & 1], (i && !nowait) ? cnt + i - 1 : nullptr, cnt + i, (unsigned)WG, err);
        }
        for (int j = 1; j < S; ++j) { CK(hipEventRecord(join[j], st[j])); CK(hipStreamWaitEvent(st[0], join[j], 0)); }
        CK(hipStreamEndCapture(st[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st[0]));
        CK(hipStreamSynchronize(st[0]));
        const int R = 20;
        CK(hipEventRecord(e0, st[0]));
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st[0]));
        CK(hipEventRecord(e1, st[0]));
        CK(hipStreamSynchronize(st[0]));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int h_err; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        std::vector<float> h(N); CK(hipMemcpy(h.data(), buf[0], N * 4, hipMemcpyDeviceToHost));
        printf("streams=%d: %.2f us per kernel (%d kernels x %d replays, %.1f us per replay)  spin-timeout=%d  out[0]=%.6f\n", S, ms * 1e3 / (R * NK), NK, R,
               ms * 1e3 / R, h_err, h[0]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        // the same without a graph: launches issued directly
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        for (int r = 0; r < R; ++r) {
            if (S > 1) {
                hipLaunchKernelGGL(zero, dim3(1), dim3(64), 0, st[0], cnt, NK);
                CK(hipEventRecord(fork, st[0]));
                for (int j = 1; j < S; ++j) CK(hipStreamWaitEvent(st[j], fork, 0));
            }
            for (int i = 0; i < NK; ++i) {
                hipStream_t s = st[i % S];
                if (S == 1) hipLaunchKernelGGL(link<false>, dim3(WG), dim3(TPB), 0, s, w, buf[i & 1], buf[(i + 1) & 1], (unsigned *)nullptr, cnt, 0u, err);
                else hipLaunchKernelGGL(link<true>, dim3(WG), dim3(TPB), 0, s, w, buf[i & 1], buf[(i + 1) & 1], (i && !nowait) ? cnt + i - 1 : nullptr, cnt + i, (unsigned)WG, err);
            }
            for (int j = 1; j < S; ++j) { CK(hipEventRecord(join[j], st[j])); CK(hipStreamWaitEvent(st[0], join[j], 0)); }
        }
        CK(hipEventRecord(e1, st[0]));
        CK(hipStreamSynchronize(st[0]));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        printf("   eager streams=%d: %.2f us per kernel  spin-timeout=%d\n", S, ms * 1e3 / (R * NK), h_err);
    }
    for (int fm = 1; fm <= 2; ++fm) {
        float ms;
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL(zero, dim3(1), dim3(64), 0, st[0], cnt, NK);
            hipLaunchKernelGGL(persistent, dim3(WG), dim3(TPB), 0, st[0], w, buf[0], buf[1], cnt, err, fm);
        }
        CK(hipStreamSynchronize(st[0]));
        const int R = 20;
        CK(hipEventRecord(e0, st[0]));
        for (int r = 0; r < R; ++r) {
            hipLaunchKernelGGL(zero, dim3(1), dim3(64), 0, st[0], cnt, NK);
            hipLaunchKernelGGL(persistent, dim3(WG), dim3(TPB), 0, st[0], w, buf[0], buf[1], cnt, err, fm);
        }
        CK(hipEventRecord(e1, st[0]));
        CK(hipStreamSynchronize(st[0]));
        CK(hipEventElapsedTime(&ms, e0, e1));
        int h_err; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        printf("persistent (%s): %.2f us per phase  spin-timeout=%d\n", fm == 1 ? "plain accesses + agent fences" : "agent-scope atomic accesses, no fences", ms * 1e3 / (R * NK), h_err);
    }
    return 0;
}
