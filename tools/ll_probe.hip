// ll_probe.hip -- what a persistent small-batch denoiser kernel would pay per phase (development probe, MI355X): W workgroups, each
// phase every workgroup writes its slice of an activation buffer as 8-byte {value, phase tag} words with agent-scope stores and then
// reads the WHOLE buffer back, spinning per word until the tag is the phase's (the "LL" protocol of collective libraries: the data is
// its own flag, one one-way trip per phase instead of write -> barrier -> read).  Compared with the same exchange through a
// counter barrier.  us per phase for W in {16 .. 256} and buffers of 20 x 512 / 20 x 1024 values.
//   hipcc --offload-arch=gfx950 -O3 tools/ll_probe.hip -o tools/ll_probe && tools/ll_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

template <int PER>   // words each thread reads per phase (E = 256 * PER)
__global__ __launch_bounds__(256) void ll_kernel(u64 *buf0, u64 *buf1, int phases, float *out, int *err) {
    const int tid = threadIdx.x, W = gridDim.x, E = 256 * PER, S = E / W;
    float carry = 1.0f;
    for (int p = 0; p < phases; ++p) {
        u64 *buf = (p & 1) ? buf1 : buf0;
        const unsigned tag = (unsigned)p + 1u;
        for (int i = tid; i < S; i += 256) {
            const u64 w = (u64)__float_as_uint(carry * 1e-3f + (float)i) | ((u64)tag << 32);
            __hip_atomic_store(buf + blockIdx.x * S + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float s = 0.0f;
        constexpr int G = 20;                     // words in flight per thread and polling round
#pragma unroll
        for (int c = 0; c < PER / G; ++c) {
            u64 v[G];
            unsigned pending = (1u << G) - 1u;
            int spins = 0;
            while (pending) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if (pending & (1u << j)) v[j] = __hip_atomic_load(buf + tid + 256 * (c * G + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if ((pending & (1u << j)) && (unsigned)(v[j] >> 32) == tag) {
                        pending &= ~(1u << j);
                        s += __uint_as_float((unsigned)v[j]);
                    }
                if (++spins > (1 << 20)) { *err = 1; break; }
            }
        }
        carry = s * 1e-6f;
    }
    if (tid == 0) out[blockIdx.x] = carry;
}

// the same exchange with plain stores, a counter barrier (agent scope) and agent-scope loads after it
template <int PER>
__global__ __launch_bounds__(256) void barrier_kernel(float *buf0, float *buf1, unsigned *cnt, int phases, float *out, int *err) {
    const int tid = threadIdx.x, W = gridDim.x, E = 256 * PER, S = E / W;
    float carry = 1.0f;
    for (int p = 0; p < phases; ++p) {
        float *buf = (p & 1) ? buf1 : buf0;
        for (int i = tid; i < S; i += 256) __hip_atomic_store(buf + blockIdx.x * S + i, carry * 1e-3f + (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt + p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cnt + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)W)
                if (++spins > (1 << 22)) { *err = 1; break; }
        }
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < PER; ++j) s += __hip_atomic_load(buf + tid + 256 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        carry = s * 1e-6f;
    }
    if (tid == 0) out[blockIdx.x] = carry;
}

int main() {
    u64 *b0, *b1;
    unsigned *cnt;
    float *out;
    int *err;
    const int P = 200;
    CK(hipMalloc(&b0, 32768 * 8)); CK(hipMalloc(&b1, 32768 * 8));
    CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&out, 1024 * 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("us per phase (every workgroup writes its slice, then reads the whole buffer); 256 threads per workgroup, workgroup b on XCD b %% 8\n");
    printf("%6s | %28s | %28s\n", "W", "20 x 512 values: LL   barrier", "20 x 1024 values: LL   barrier");
    for (int W : {8, 16, 32, 64, 128, 256}) {
        float r[4];
        for (int big = 0; big < 2; ++big) {
            for (int mode = 0; mode < 2; ++mode) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipMemset(b0, 0, 32768 * 8)); CK(hipMemset(b1, 0, 32768 * 8)); CK(hipMemset(cnt, 0, 4096 * 4));
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, 0));
                    if (mode == 0) {
                        if (big) hipLaunchKernelGGL(ll_kernel<80>, dim3(W), dim3(256), 0, 0, b0, b1, P, out, err);
                        else hipLaunchKernelGGL(ll_kernel<40>, dim3(W), dim3(256), 0, 0, b0, b1, P, out, err);
                    } else {
                        if (big) hipLaunchKernelGGL(barrier_kernel<80>, dim3(W), dim3(256), 0, 0, (float *)b0, (float *)b1, cnt, P, out, err);
                        else hipLaunchKernelGGL(barrier_kernel<40>, dim3(W), dim3(256), 0, 0, (float *)b0, (float *)b1, cnt, P, out, err);
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                r[big * 2 + mode] = best * 1e3f / P;
            }
        }
        int h_err;
        CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        printf("%6d | %14.2f %12.2f | %14.2f %12.2f   %s\n", W, r[0], r[1], r[2], r[3], h_err ? "SPIN TIMEOUT" : "");
    }
    return 0;
}
