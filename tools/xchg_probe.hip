// xchg_probe.hip -- the GGS cross-workgroup exchange alone (development probe, not part of the library): K workgroups of 512 threads,
// every wave publishes 12 tagged 8-byte granules {epoch, value} per iteration, every workgroup gathers all K x 8 x 12 of them (the data is
// the flag) -- the all-gather pd_ggs_kernel runs once per iteration at more than one workgroup per sequence.  What does an iteration of
// it cost, and what do placement (workgroups spread over the XCDs / all on one XCD) and the cache policy of the accesses change?
//   hipcc --offload-arch=gfx950 -O3 tools/xchg_probe.hip -o tools/xchg_probe && tools/xchg_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LINE 16

// LOADS: 0 sc1 (agent scope: what the engine uses), 1 sc0, 2 sc0 sc1, 3 plain load after buffer_inv sc0, 4 plain load after buffer_inv sc1, 5 nt
template <int LOADS>
__device__ __forceinline__ void load3(u32x4 &v0, u32x4 &v1, u32x4 &v2, const u64 *a0, const u64 *a1, const u64 *a2) {
    if (LOADS == 0)
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
    else if (LOADS == 1)
        asm volatile("global_load_dwordx4 %0, %3, off sc0\n\tglobal_load_dwordx4 %1, %4, off sc0\n\tglobal_load_dwordx4 %2, %5, off sc0\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
    else if (LOADS == 2)
        asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off sc0 sc1\n\tglobal_load_dwordx4 %2, %5, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
    else if (LOADS == 3)
        asm volatile("buffer_inv sc0\n\tglobal_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
    else if (LOADS == 4)
        asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %3, off nt\n\tglobal_load_dwordx4 %1, %4, off nt\n\tglobal_load_dwordx4 %2, %5, off nt\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
}

// STORES: 0 agent-scope atomic store (sc1: what the engine uses), 1 plain store, 2 store sc0
template <int STORES>
__device__ __forceinline__ void store1(u64 *g, u64 v) {
    if (STORES == 0) __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (STORES == 1) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(g), "v"(v) : "memory");
}

template <int LOADS, int STORES>
__global__ __launch_bounds__(512) void xchg(u64 *buf, size_t stride, int K, int iters, int stride_blocks, unsigned *bad, long long *cyc) {
    __shared__ float item[256 * 12];
    if (blockIdx.x % stride_blocks != 0) return;                 // placement: only every stride_blocks-th block works (8: all on XCD 0)
    const int wg = blockIdx.x / stride_blocks, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_items = K * 8, n_piece = n_items * 6;
    unsigned nbad = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (unsigned epoch = 1; epoch <= (unsigned)iters; ++epoch) {
        u64 *slot = buf + (size_t)(epoch & 1) * stride;
        const int my_item = wg * 8 + wave;
        if (lane < 12) store1<STORES>(slot + (size_t)my_item * LINE + lane, ((u64)epoch << 32) | (u64)__float_as_uint((float)(my_item * 16 + lane) + (float)epoch));
        for (int p0 = tid; p0 < n_piece; p0 += 3 * 512) {
            const u64 *a[3];
            int pi_[3];
            for (int u = 0; u < 3; ++u) {
                const int pc = p0 + u * 512;
                pi_[u] = pc < n_piece ? pc : p0;
                a[u] = slot + (size_t)(pi_[u] / 6) * LINE + (pi_[u] % 6) * 2;
            }
            u32x4 v0, v1, v2;
            unsigned spins = 0;
            for (;;) {
                load3<LOADS>(v0, v1, v2, a[0], a[1], a[2]);
                if (v0[1] == epoch && v0[3] == epoch && v1[1] == epoch && v1[3] == epoch && v2[1] == epoch && v2[3] == epoch) break;
                if (++spins > (1u << 14)) { nbad |= 0x80000000u; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const u32x4 vv[3] = {v0, v1, v2};
            for (int u = 0; u < 3; ++u)
                if (p0 + u * 512 < n_piece) {
                    const int it = pi_[u] / 6, g = (pi_[u] % 6) * 2;
                    item[it * 12 + g] = __uint_as_float(vv[u][0]);
                    item[it * 12 + g + 1] = __uint_as_float(vv[u][2]);
                    if (__uint_as_float(vv[u][0]) != (float)(it * 16 + g) + (float)epoch || __uint_as_float(vv[u][2]) != (float)(it * 16 + g + 1) + (float)epoch) ++nbad;
                }
        }
        if (__syncthreads_or((nbad & 0x80000000u) != 0)) break;     // a timed-out exchange ends the run (never visible with this policy)
    }
    const long long t1 = __builtin_readcyclecounter();
    if (nbad) atomicOr(bad, nbad);
    if (tid == 0 && wg == 0) cyc[0] = t1 - t0;
    if (item[tid] == -1.0f) bad[1] = 1;
}

template <int LOADS, int STORES>
static void run(const char *name, u64 *buf, size_t stride, int K, int stride_blocks, unsigned *bad, long long *cyc) {
    const int iters = 2000;
    hipMemset(buf, 0, 2 * stride * sizeof(u64));
    hipMemset(bad, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((xchg<LOADS, STORES>), dim3(K * stride_blocks), dim3(512), 0, 0, buf, stride, K, iters, stride_blocks, bad, cyc);
    hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned hb[2];
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("%-44s K=%2d %-9s %8.3f us / iteration   %s%s\n", name, K, stride_blocks == 8 ? "one XCD" : "spread", ms * 1e3 / iters,
           e != hipSuccess ? hipGetErrorString(e) : (hb[0] & 0x80000000u ? "TIMEOUT " : (hb[0] ? "WRONG DATA " : "ok")), "");
}

int main() {
    const size_t stride = 64 * 8 * LINE;
    u64 *buf;
    unsigned *bad;
    long long *cyc;
    hipMalloc(&buf, 2 * stride * sizeof(u64));
    hipMalloc(&bad, 8);
    hipMalloc(&cyc, 8);
    for (int K : {24, 8}) {
        for (int sb : {1, 8}) {
            run<0, 0>("loads sc1, stores agent atomic (engine)", buf, stride, K, sb, bad, cyc);
            run<2, 0>("loads sc0 sc1, stores agent atomic", buf, stride, K, sb, bad, cyc);
            run<1, 0>("loads sc0, stores agent atomic", buf, stride, K, sb, bad, cyc);
            run<1, 1>("loads sc0, stores plain", buf, stride, K, sb, bad, cyc);
            run<1, 2>("loads sc0, stores sc0", buf, stride, K, sb, bad, cyc);
            run<0, 1>("loads sc1, stores plain", buf, stride, K, sb, bad, cyc);
            run<3, 1>("buffer_inv sc0 + plain loads, stores plain", buf, stride, K, sb, bad, cyc);
            run<5, 1>("loads nt, stores plain", buf, stride, K, sb, bad, cyc);
        }
    }
    return 0;
}
