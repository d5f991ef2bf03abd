// xcd_sync_probe -- measures what a per-XCD cooperative kernel relies on (MI355X, gfx950):
//   T1  does workgroup b of a launch run on XCD b % 8 (HW_REG_XCC_ID)?
//   T2  cost of a barrier among the W workgroups of one XCD (counter in L2) vs. one among all 8*W
//   T3  is data written by other CUs of the same XCD visible after that barrier, and with which
//       cache maintenance (none / buffer_inv sc0 / buffer_inv sc1 / sc1 loads)?  cost per exchange
// build: hipcc --offload-arch=gfx950 -O3 tools/xcd_sync_probe.hip -o tools/xcd_sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);      \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

__global__ void t1_kernel(unsigned *out) {
    extern __shared__ float pad[];
    if (threadIdx.x == 0) {
        pad[0] = 1.0f;
        out[blockIdx.x] = xcc_id();
    }
}

// POLL: 0 = sc1 atomic load, 1 = sc0 atomic load, 2 = fetch_add(0)
template <int POLL>
__device__ __forceinline__ unsigned poll(unsigned *p) {
    if (POLL == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (POLL == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// barrier among `n` workgroups on counter `c` (monotonic); returns false if the spin gave up
template <int POLL>
__device__ __forceinline__ bool group_barrier(unsigned *c, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (long spin = 0; spin < 20000000L; ++spin) {
            if (poll<POLL>(c) >= target) {
                good = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

template <int POLL>
__global__ void t2_kernel(unsigned *ctr, int W, int global_scope, int iters, unsigned long long *cycles, unsigned *err) {
    const int x = blockIdx.x & 7;
    unsigned *c = global_scope ? ctr : ctr + 32 * (1 + x);
    const unsigned n = global_scope ? 8u * W : (unsigned)W;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (!group_barrier<POLL>(c, n * (unsigned)(it + 1))) {
            if (threadIdx.x == 0) atomicAdd(err, 1u);
            return;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// INV: 0 none, 1 buffer_inv sc0, 2 buffer_inv sc1, 3 = sc1 loads (no invalidate)
template <int INV>
__global__ void t3_kernel(unsigned *ctr, float *buf, int W, int slice_floats, int iters, unsigned long long *cycles,
                          unsigned *err, unsigned *stale) {
    const int x = blockIdx.x & 7, wx = blockIdx.x >> 3;
    unsigned *c = ctr + 32 * (1 + x);
    float *xb = buf + (size_t)x * 2 * W * slice_floats;
    unsigned bad = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float *cur = xb + (size_t)(it & 1) * W * slice_floats;
        float4 *mine = (float4 *)(cur + (size_t)wx * slice_floats);
        for (int i = threadIdx.x; i < slice_floats / 4; i += blockDim.x) {
            const float v = (float)(it * 7 + wx * 3 + (i & 63));
            mine[i] = make_float4(v, v + 1.0f, v + 2.0f, v + 3.0f);
        }
        if (!group_barrier<0>(c, (unsigned)W * (unsigned)(it + 1))) {
            if (threadIdx.x == 0) atomicAdd(err, 1u);
            return;
        }
        if (INV == 1) asm volatile("buffer_inv sc0" ::: "memory");
        if (INV == 2) asm volatile("buffer_inv sc1" ::: "memory");
        for (int nb = 1; nb <= 4; ++nb) {
            const int o = (wx + nb * 7) % W;
            const float4 *theirs = (const float4 *)(cur + (size_t)o * slice_floats);
            for (int i = threadIdx.x; i < slice_floats / 4; i += blockDim.x) {
                float4 g;
                if (INV == 3) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g) : "v"(theirs + i) : "memory");
                } else {
                    g = theirs[i];
                }
                const float v = (float)(it * 7 + o * 3 + (i & 63));
                bad += (g.x != v) + (g.w != v + 3.0f);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (bad) atomicAdd(stale, bad);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    unsigned *d_u;
    unsigned long long *d_cyc;
    float *d_buf;
    const int slice = 2048;   // floats per workgroup slice (8 KiB)
    CK(hipMalloc(&d_u, 4096 * 4));
    CK(hipMalloc(&d_cyc, 8));
    CK(hipMalloc(&d_buf, (size_t)8 * 2 * 32 * slice * 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, prop.multiProcessorCount, ghz);
    CK(hipFuncSetAttribute((const void *)t1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));

    // T1
    for (int grid : {64, 256, 512}) {
        for (int lds : {1024, 140 * 1024}) {
            CK(hipMemset(d_u, 0xff, 4096 * 4));
            hipLaunchKernelGGL(t1_kernel, dim3(grid), dim3(256), lds, 0, d_u);
            CK(hipDeviceSynchronize());
            std::vector<unsigned> h(grid);
            CK(hipMemcpy(h.data(), d_u, grid * 4, hipMemcpyDeviceToHost));
            int mism = 0;
            for (int b = 0; b < grid; ++b) mism += (h[b] != (unsigned)(b & 7));
            printf("T1 grid=%d lds=%dK: %d of %d workgroups NOT on XCD b%%8\n", grid, lds / 1024, mism, grid);
        }
    }
    auto zero = [&]() { CK(hipMemset(d_u, 0, 4096 * 4)); CK(hipMemset(d_cyc, 0, 8)); };
    auto report = [&](const char *name, int iters) {
        CK(hipDeviceSynchronize());
        unsigned long long cyc;
        unsigned h[2];
        CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h, d_u, 8, hipMemcpyDeviceToHost));
        printf("%s: %.0f cycles (%.3f us at shader clock) per iteration; spin timeouts=%u stale words=%u\n", name,
               (double)cyc / iters, (double)cyc / iters / (ghz * 1e3), h[0], h[1]);
    };
    // counters: d_u[0] = err, d_u[1] = stale, d_u[32*(1+x)] = XCD counters, d_u[0 + 32*9] global
    const int iters = 2000;
    for (int W : {32, 8}) {
        char nm[128];
        zero();
        hipLaunchKernelGGL(t2_kernel<0>, dim3(8 * W), dim3(256), 0, 0, d_u + 0, W, 0, iters, d_cyc, d_u);
        snprintf(nm, sizeof nm, "T2 XCD-local barrier W=%d poll=sc1-load", W);
        report(nm, iters);
        zero();
        hipLaunchKernelGGL(t2_kernel<1>, dim3(8 * W), dim3(256), 0, 0, d_u + 0, W, 0, iters, d_cyc, d_u);
        snprintf(nm, sizeof nm, "T2 XCD-local barrier W=%d poll=sc0-load", W);
        report(nm, iters);
        zero();
        hipLaunchKernelGGL(t2_kernel<2>, dim3(8 * W), dim3(256), 0, 0, d_u + 0, W, 0, iters, d_cyc, d_u);
        snprintf(nm, sizeof nm, "T2 XCD-local barrier W=%d poll=rmw", W);
        report(nm, iters);
        zero();
        hipLaunchKernelGGL(t2_kernel<0>, dim3(8 * W), dim3(256), 0, 0, d_u + 32 * 9, W, 1, iters, d_cyc, d_u);
        snprintf(nm, sizeof nm, "T2 device-wide barrier %d WGs poll=sc1-load", 8 * W);
        report(nm, iters);
        zero();
        hipLaunchKernelGGL(t2_kernel<2>, dim3(8 * W), dim3(256), 0, 0, d_u + 32 * 9, W, 1, iters, d_cyc, d_u);
        snprintf(nm, sizeof nm, "T2 device-wide barrier %d WGs poll=rmw", 8 * W);
        report(nm, iters);
    }
    for (int W : {32, 8}) {
        char nm[128];
        zero();
        hipLaunchKernelGGL(t3_kernel<0>, dim3(8 * W), dim3(256), 0, 0, d_u, d_buf, W, slice, iters, d_cyc, d_u, d_u + 1);
        snprintf(nm, sizeof nm, "T3 exchange W=%d no invalidate, plain loads", W);
        report(nm, iters);
        zero();
        hipLaunchKernelGGL(t3_kernel<1>, dim3(8 * W), dim3(256), 0, 0, d_u, d_buf, W, slice, iters, d_cyc, d_u, d_u + 1);
        snprintf(nm, sizeof nm, "T3 exchange W=%d buffer_inv sc0, plain loads", W);
        report(nm, iters);
        zero();
        hipLaunchKernelGGL(t3_kernel<2>, dim3(8 * W), dim3(256), 0, 0, d_u, d_buf, W, slice, iters, d_cyc, d_u, d_u + 1);
        snprintf(nm, sizeof nm, "T3 exchange W=%d buffer_inv sc1, plain loads", W);
        report(nm, iters);
        zero();
        hipLaunchKernelGGL(t3_kernel<3>, dim3(8 * W), dim3(256), 0, 0, d_u, d_buf, W, slice, iters, d_cyc, d_u, d_u + 1);
        snprintf(nm, sizeof nm, "T3 exchange W=%d sc1 loads", W);
        report(nm, iters);
    }
    return 0;
}
