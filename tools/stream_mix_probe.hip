// stream_mix_probe.hip -- can a HEAD of every CU's private, cyclically re-read region be kept in the XCD's L2 (4 MiB per 32 CUs = 128 KiB per
// CU) by loading the rest of the region with a cache policy that does not allocate there?  (The GGS match stream: ~728 KB per CU and
// iteration, the same bytes every iteration; with one policy for all of it an LRU cache of 4 MiB under a 23 MiB cycle never hits.)
// Every CU re-reads R KB; the first H KB with plain loads, the remaining R - H KB with policy P.  Development probe, not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_mix_probe.hip -o tools/stream_mix_probe && tools/stream_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int P>
__device__ __forceinline__ f4 ld(const f4 *p) {
    f4 v;
    if (P == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if (P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if (P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if (P == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if (P == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int P>
__global__ __launch_bounds__(512) void stream(const f4 *base, size_t region_f4, size_t head_f4, int iters, float *out) {
    extern __shared__ float pad[];
    const f4 *p = base + (size_t)blockIdx.x * region_f4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int it = 0; it < iters; ++it) {
        for (size_t i = threadIdx.x; i < region_f4; i += (size_t)512 * U) {      // head_f4 and region_f4 are multiples of 512 * U
            f4 v[U];
            if (i < head_f4) {
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ld<0>(p + i + (size_t)u * 512);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ld<P>(p + i + (size_t)u * 512);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + pad[threadIdx.x & 7];
}

typedef void (*kern_t)(const f4 *, size_t, size_t, int, float *);
int main() {
    const size_t region_f4 = (size_t)11 * 512 * 8;            // 11 rounds of 64 KiB = 704 KiB per CU
    f4 *buf;
    float *out;
    hipMalloc(&buf, region_f4 * 16 * 256);
    hipMalloc(&out, 256 * 512 * 4);
    hipMemset(buf, 0, region_f4 * 16 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 96 * 1024;
    kern_t ks[] = {stream<0>, stream<1>, stream<2>, stream<3>, stream<4>, stream<5>};
    const char *kn[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0"};
    for (int k = 0; k < 6; ++k) hipFuncSetAttribute((const void *)ks[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("region 704 KiB per CU, 256 CUs, 512 threads x 8 loads in flight; head = the part loaded with plain loads\n");
    printf("%-12s %8s %10s %10s\n", "tail policy", "head KiB", "ms/pass", "TB/s");
    const int iters = 60;
    for (int k = 0; k < 6; ++k)
        for (int head : {0, 64, 128}) {
            if (k == 0 && head) continue;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(ks[k], dim3(256), dim3(512), lds, 0, buf, region_f4, (size_t)head * 64, iters, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)region_f4 * 16 * 256 * iters;
            printf("%-12s %8d %10.4f %10.2f\n", kn[k], head, ms / iters, bytes / (ms * 1e-3) / 1e12);
        }
    return 0;
}
