// stream_probe.hip -- how fast can every CU re-read a PRIVATE region over and over (the GGS match stream: 912 KB per sequence and
// iteration, one sequence per CU => 233 MB per chip, Infinity-Cache resident)?  Development probe, not part of the library.
// One workgroup per CU (LDS-padded), W waves per SIMD, U dwordx4 loads in flight per wave; region sizes put the working set in
// the L2s (25 MB), the Infinity Cache (233 MB) or HBM (934 MB).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4v __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void stream(const float4 *base, size_t region_f4, int iters, float *out) {
    extern __shared__ float pad[];
    const float4 *p = base + (size_t)blockIdx.x * region_f4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nt = blockDim.x;
    for (int it = 0; it < iters; ++it) {
        size_t off = 0;
        asm volatile("" : "+s"(off));   // the same addresses every pass: keep the compiler from hoisting the loads
        const float4 *q = p + off;
        for (size_t i = threadIdx.x; i < region_f4; i += (size_t)nt * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (NT) {
                    const f4v t = __builtin_nontemporal_load((const f4v *)(q + i + (size_t)u * nt));
                    v[u] = make_float4(t.x, t.y, t.z, t.w);
                } else {
                    v[u] = q[i + (size_t)u * nt];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + pad[threadIdx.x & 7];
}

typedef void (*kern_t)(const float4 *, size_t, int, float *);
int main() {
    const size_t max_bytes = (size_t)256 * 3648 * 1024;
    float4 *buf;
    float *out;
    if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&out, 256 * 1024 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, max_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 96 * 1024;
    kern_t ks[] = {stream<4, false>, stream<8, false>, stream<8, true>};
    const char *kn[] = {"4 in flight", "8 in flight", "8 in flight nt"};
    for (int k = 0; k < 3; ++k) hipFuncSetAttribute((const void *)ks[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("%-10s %-16s %6s %10s %10s\n", "region/CU", "loads", "W/SIMD", "ms/pass", "TB/s");
    for (int kb : {96, 456, 912, 3648})
        for (int k = 0; k < 3; ++k)
            for (int w : {2, 3, 4}) {
                const int nt = 256 * w;
                // region rounded down to a whole number of (threads x 8 loads) rounds
                const size_t round_f4 = (size_t)nt * 8;
                const size_t region_f4 = ((size_t)kb * 1024 / 16) / round_f4 * round_f4;
                const int iters = kb >= 3648 ? 6 : (kb >= 456 ? 40 : 200);
                float ms = 0;
                for (int pass = 0; pass < 2; ++pass) {
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(ks[k], dim3(256), dim3(nt), lds, 0, buf, region_f4, iters, out);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double bytes = (double)region_f4 * 16 * 256 * iters;
                printf("%6d KB  %-16s %6d %10.4f %10.2f\n", kb, kn[k], w, ms / iters, bytes / (ms * 1e-3) / 1e12);
            }
    return 0;
}
