// Micro-benchmark 5: what a wave's L1-resident vector load costs on gfx950's address (TA) / data-return (TD) path.
// Every CU runs 16 waves (4 per SIMD) that issue back-to-back global loads into a 16 KB (L1-resident) buffer:
//   width   dword / dwordx2 / dwordx4
//   pattern "gather": lane i reads its own 128-B line (a BVH node fetch: 64 lanes -> 64 lines)
//           "line":   the 64 lanes read one contiguous run of 64 x width bytes
//   lanes   all 64, even lanes, lanes 0..31, one lane in four, 16 contiguous lanes
// Prints shader cycles per wave-load-instruction per CU (2.4 GHz nominal) and the bytes per clock per CU that implies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int WIDTH>
__global__ __launch_bounds__(256) void k_load(const char* __restrict__ buf, float* out, int iters, uint32_t stride, unsigned long long lanes, uint32_t rot)
{
    const uint32_t lane = threadIdx.x & 63;
    float acc = 0.f;
    if ((lanes >> lane) & 1ull) {
        uint32_t off = (lane * stride + (threadIdx.x >> 6) * 1024u) & 16383u;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (WIDTH == 4) { float4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(buf)); asm volatile("s_waitcnt vmcnt(6)"); acc += v.x; }
                if constexpr (WIDTH == 2) { float2 v; asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(buf)); asm volatile("s_waitcnt vmcnt(6)"); acc += v.x; }
                if constexpr (WIDTH == 1) { float v;  asm volatile("global_load_dword %0, %1, %2"   : "=v"(v) : "v"(off), "s"(buf)); asm volatile("s_waitcnt vmcnt(6)"); acc += v; }
                off = (off + rot) & 16383u;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class K> double run(K kern, const char* buf, float* out, int iters, uint32_t stride, unsigned long long lanes, uint32_t rot)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, buf, out, 10, stride, lanes, rot); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, buf, out, iters, stride, lanes, rot);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main()
{
    char* buf; float* out;
    (void)hipMalloc(&buf, 1 << 16); (void)hipMemset(buf, 0, 1 << 16); (void)hipMalloc(&out, 1024 * 256 * sizeof(float));
    const int iters = 4000;
    struct Mask { const char* name; unsigned long long m; } masks[] = {
        { "all 64", ~0ull }, { "even lanes", 0x5555555555555555ull }, { "lanes 0..31", 0xFFFFFFFFull },
        { "1 lane in 4", 0x1111111111111111ull }, { "lanes 0..15", 0xFFFFull }, { "quads 0,2,4..", 0x0F0F0F0F0F0F0F0Full } };
    printf("cycles per wave-load per CU (1024 blocks x 4 waves, 16 waves resident per CU on 256 CUs), 2.4 GHz nominal\n");
    printf("%-8s %-8s %-14s %10s %12s\n", "width", "pattern", "lanes", "cyc/load", "B/clk/CU");
    for (int w : { 4, 2, 1 })
        for (int pat = 0; pat < 2; ++pat)
            for (const Mask& mk : masks) {
                const uint32_t stride = pat == 0 ? 128u : (uint32_t)w * 4u;
                const uint32_t rot = pat == 0 ? 16u : 1024u;          // gather: next 16 B of the same line (then wraps); line: next KB
                double ms = w == 4 ? run(k_load<4>, buf, out, iters, stride, mk.m, rot)
                          : w == 2 ? run(k_load<2>, buf, out, iters, stride, mk.m, rot) : run(k_load<1>, buf, out, iters, stride, mk.m, rot);
                // per CU: 4 blocks (1024 / 256 CUs) x 4 waves x iters x 8 loads
                const double loads_per_cu = 4.0 * 4.0 * iters * 8.0;
                const double cyc = ms * 1e-3 * 2.4e9 / loads_per_cu;
                const int nl = __builtin_popcountll(mk.m);
                printf("x%-7d %-8s %-14s %10.2f %12.1f\n", w, pat == 0 ? "gather" : "line", mk.name, cyc, nl * w * 4.0 / cyc);
            }
    return 0;
}
