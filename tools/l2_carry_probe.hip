// l2_carry_probe.hip -- does a line that kernel A pulled into an XCD's L2 survive the kernel boundary, so that a DEPENDENT kernel B on the same
// stream finds it there?  (Round 5: the small-batch denoiser's GEMM kernels wait ~1.5 us for their first weight bytes -- tools/den_small_legs.py --;
// the weights are not data dependent, so kernel i could pull kernel i + 1's slice into the right L2 while its own MFMA chain runs.)
// 256 blocks x 256 threads, block b runs on XCD b % 8 (observed).  A slice of S KB per block is read by `touch` (plain loads, results discarded
// through a never-true store) and then by `timed`, which records with s_memrealtime (100 MHz) the time from its first load to all S KB landed, and
// from entry to landed.  Scenarios: cold (another 512 MB region streamed in between), touched by the same block index (same XCD), touched by
// block index + 1 (a NEIGHBOUR XCD: the slice is then in the wrong L2), and touched through LDS-DMA instead of plain loads.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_carry_probe.hip -o tools/l2_carry_probe && tools/l2_carry_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(256) void touch(const float4 *base, int slice_f4, int shift, float *sink) {
    const float4 *p = base + (size_t)((blockIdx.x + shift) % gridDim.x) * slice_f4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < slice_f4; i += 256) {
        const float4 v = p[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) sink[blockIdx.x] = a.x;
}
// the same through LDS-DMA into a 1 KiB sink (what a GEMM kernel would issue beside its MFMA chain: no registers, no data use)
__global__ __launch_bounds__(256) void touch_dma(const float4 *base, int slice_f4, int shift, float *sink) {
    __shared__ __attribute__((aligned(1024))) float4 dump[4 * 64];
    const float4 *p = base + (size_t)((blockIdx.x + shift) % gridDim.x) * slice_f4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(dump + wave * 64));
    for (int i = wave * 64; i < slice_f4; i += 256) {
        const unsigned off = (unsigned)(i + lane) * 16u;
        unsigned keep;
        asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o], %[b]\n\ts_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep) : [d] "s"(dst), [b] "s"(p), [o] "v"(off) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dump[threadIdx.x].x == 12345.678f) sink[blockIdx.x] = 1.0f;
}
__global__ __launch_bounds__(256) void timed(const float4 *base, int slice_f4, long long *out, float *sink) {
    const long long t_in = (long long)__builtin_amdgcn_s_memrealtime();
    const float4 *p = base + (size_t)blockIdx.x * slice_f4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    // up to 16 loads in flight per thread (16 KB per wave): the shape of a GEMM kernel's first weight batch
    for (int i0 = threadIdx.x; i0 < slice_f4; i0 += 256 * 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (i0 + u * 256 < slice_f4) ? p[i0 + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 16; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x)::"memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = t1 - t_in;
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) sink[blockIdx.x] = a.x;
}
__global__ void sweep(const float4 *p, size_t n, float *sink) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        a.x += v.x; a.y += v.y;
    }
    if (a.x + a.y == 12345.678f) sink[0] = a.x;
}

int main() {
    const int blocks = 256;
    const size_t evict_bytes = (size_t)512 << 20;
    float4 *w, *evict;
    float *sink;
    long long *out;
    if (hipMalloc(&w, (size_t)blocks * 256 * 1024) != hipSuccess || hipMalloc(&evict, evict_bytes) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess ||
        hipMalloc(&out, sizeof(long long) * 2 * blocks) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(w, 0, (size_t)blocks * 256 * 1024);
    (void)hipMemset(evict, 0, evict_bytes);
    std::vector<long long> h(2 * blocks);
    auto report = [&](const char *what, int kb) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), out, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
        std::vector<long long> a, b;
        for (int i = 0; i < blocks; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        printf("%3d KB/block  %-44s first load -> all landed: median %5lld ns (p10 %5lld, p90 %5lld); entry -> landed: median %5lld ns\n", kb, what,
               a[blocks / 2] * 10, a[blocks / 10] * 10, a[blocks * 9 / 10] * 10, b[blocks / 2] * 10);
        fflush(stdout);
    };
    for (int kb : {16, 32, 96}) {
        const int slice_f4 = kb * 1024 / 16;
        for (int rep = 0; rep < 2; ++rep) {
            // cold: stream 512 MB (twice the Infinity Cache) in between
            hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, evict, evict_bytes / 16, sink);
            hipLaunchKernelGGL(timed, dim3(blocks), dim3(256), 0, 0, w, slice_f4, out, sink);
            if (rep) report("after streaming 512 MB (HBM)", kb);
            // warm in the Infinity Cache but not in any L2: touch everything, then stream 64 MB (2 x the L2s) of another region
            hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, 0, w, slice_f4, 0, sink);
            hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, evict, ((size_t)64 << 20) / 16, sink);
            hipLaunchKernelGGL(timed, dim3(blocks), dim3(256), 0, 0, w, slice_f4, out, sink);
            if (rep) report("touched, then 64 MB streamed (Infinity Cache)", kb);
            hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, 0, w, slice_f4, 0, sink);
            hipLaunchKernelGGL(timed, dim3(blocks), dim3(256), 0, 0, w, slice_f4, out, sink);
            if (rep) report("touched by the SAME block index just before", kb);
            hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, 0, w, slice_f4, 1, sink);
            hipLaunchKernelGGL(timed, dim3(blocks), dim3(256), 0, 0, w, slice_f4, out, sink);
            if (rep) report("touched by block index - 1 (neighbour XCD)", kb);
            hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, 0, w, slice_f4, 8, sink);
            hipLaunchKernelGGL(timed, dim3(blocks), dim3(256), 0, 0, w, slice_f4, out, sink);
            if (rep) report("touched by block index - 8 (same XCD, other CU)", kb);
            hipLaunchKernelGGL(touch_dma, dim3(blocks), dim3(256), 0, 0, w, slice_f4, 8, sink);
            hipLaunchKernelGGL(timed, dim3(blocks), dim3(256), 0, 0, w, slice_f4, out, sink);
            if (rep) report("LDS-DMA-touched by block index - 8 (same XCD)", kb);
        }
    }
    return 0;
}
