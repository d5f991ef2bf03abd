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

// The lane-per-item GGS kernel's own pattern (csrc/pd_ggs_lane.inc; VERDICT / ADVICE round 4: the plain-load rows above were beaten by the
// kernel they were meant to bound): one workgroup of 8 waves per CU, every wave streams ITS contiguous share of the private region in
// steps of 2 KiB (two global_load_lds_dwordx4 of 1 KiB) through a ring of RING slots in LDS, RING steps in flight at all times, a slot is
// read out with two ds_read_b128 per lane and refilled at once; the stream is periodic (pass after pass without draining).  `steps` =
// steps per wave and pass; wave w's share starts at w * steps * 2 KiB.
// ZIGZAG (round 6): the passes alternate direction (0 .. steps-1, steps-1 .. 0, ...): what a pass touched last the next one touches first, so the
// tail of every wave's stream can still sit in the XCD's L2 (4 MB per 32 CUs = 64 steps of 2 KiB per CU) instead of coming from the Infinity Cache
template <int RING, bool ZIGZAG = false>
__global__ __launch_bounds__(512, 2) void stream_ring(const float4 *base, size_t region_f4, int steps, int iters, float *out) {
    extern __shared__ __attribute__((aligned(1024))) float4 ring[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float4 *src = base + (size_t)blockIdx.x * region_f4 + (size_t)wave * steps * 128;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ring + wave * RING * 128));
    const float4 *rd = ring + wave * RING * 128 + lane;
    const unsigned lane16 = lane * 16u;
    auto fetch = [&](int slot, int idx) {
        const unsigned off0 = (unsigned)idx * 2048u + lane16, off1 = off0 + 1024u;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)slot * 2048u);
        unsigned keep;
        asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o0], %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %[o1], %[b]\n\ts_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep) : [d] "s"(dst), [b] "s"(src), [o0] "v"(off0), [o1] "v"(off1) : "memory", "scc");
    };
    int next = 0, dir = 1;
    auto advance = [&]() {
        if (ZIGZAG) {
            if (next + dir < 0 || next + dir >= steps) dir = -dir;      // the turning step is fetched twice in a row (it is in the L2 by then)
            else next += dir;
        } else {
            next = (next + 1 == steps) ? 0 : next + 1;
        }
    };
    for (int u = 0; u < RING; ++u) {
        fetch(u, next);
        advance();
    }
    int slot = 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long total = (long long)steps * iters;
    for (long long t = 0; t < total; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RING - 2) : "memory");       // the oldest step has landed
        float4 q0 = rd[slot * 128], q1 = rd[slot * 128 + 64];
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0.x), "+v"(q1.x) :: "memory");  // read out before the DMA may overwrite the slot
        fetch(slot, next);
        advance();
        slot = (slot + 1 == RING) ? 0 : slot + 1;
        acc.x += q0.x + q1.x; acc.y += q0.y + q1.y; acc.z += q0.z + q1.z; acc.w += q0.w + q1.w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
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
                // a launch the device refused (resource limits) is not a measurement: say so instead of printing its 0.0001 ms
                if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("%6d KB  %-16s %6d  launch failed\n", kb, kn[k], w); continue; }
                const double bytes = (double)region_f4 * 16 * 256 * iters;
                printf("%6d KB  %-16s %6d %10.4f %10.2f\n", kb, kn[k], w, ms / iters, bytes / (ms * 1e-3) / 1e12);
            }
    // the lane kernel's pattern: LDS-DMA rings, 8 waves per CU; 708 KB per CU and pass = what a 57 000-match sequence streams per iteration
    // beside its register-resident steps (354 steps of 2 KiB: here 44 per wave = 704 KB), and the whole 912 KB for comparison
    typedef void (*ring_t)(const float4 *, size_t, int, int, float *);
    ring_t rk[] = {stream_ring<4>, stream_ring<6>, stream_ring<8>};
    const int rdepth[] = {4, 6, 8};
    for (int k = 0; k < 3; ++k) hipFuncSetAttribute((const void *)rk[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-10s %-16s %6s %10s %10s\n", "region/CU", "LDS-DMA ring", "waves", "ms/pass", "TB/s");
    for (int steps : {44, 57})
        for (int k = 0; k < 3; ++k) {
            const size_t region_f4 = (size_t)8 * steps * 128;
            const int iters = 60;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                hipEventRecord(e0, 0);
                // 160 KiB of LDS per workgroup (the kernel's own footprint): exactly one workgroup per CU
                hipLaunchKernelGGL(rk[k], dim3(256), dim3(512), 160 * 1024, 0, buf, region_f4, steps, iters, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("FAILED ring of %d, %d steps\n", rdepth[k], steps); continue; }
            const double bytes = (double)region_f4 * 16 * 256 * iters;
            fflush(stdout); printf("RING %4zu KB  ring of %d x 2 KiB  %6d %10.4f %10.2f\n", region_f4 * 16 / 1024, rdepth[k], 8, ms / iters, bytes / (ms * 1e-3) / 1e12);
        }
    // round 6: the same rings with passes of alternating direction (rows "ZIGZAG": not a bench.py ceiling row), ring of 3 and 4, 40 steps per wave = the
    // streamed part of a sequence under the [75 x 4, 38 x 4] cuts (58 + 21 steps per wave pair)
    ring_t zk[] = {stream_ring<3, false>, stream_ring<3, true>, stream_ring<4, false>, stream_ring<4, true>};
    const char *zn[] = {"ring 3 periodic", "ring 3 zigzag", "ring 4 periodic", "ring 4 zigzag"};
    for (int k = 0; k < 4; ++k) hipFuncSetAttribute((const void *)zk[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int steps : {40, 57})
        for (int rep = 0; rep < 2; ++rep)
            for (int k = 0; k < 4; ++k) {
                const size_t region_f4 = (size_t)8 * steps * 128;
                const int iters = 60;
                float ms = 0;
                for (int pass = 0; pass < 2; ++pass) {
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(zk[k], dim3(256), dim3(512), 160 * 1024, 0, buf, region_f4, steps, iters, out);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("FAILED %s\n", zn[k]); continue; }
                const double bytes = (double)region_f4 * 16 * 256 * iters;
                printf("ZIGZAG %4zu KB  %-16s %10.4f ms/pass %10.2f TB/s\n", region_f4 * 16 / 1024, zn[k], ms / iters, bytes / (ms * 1e-3) / 1e12);
            }
    return 0;
}
