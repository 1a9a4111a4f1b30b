// verify.hip — verification on gfx950 of the short correctly-rounded sequences in csrc/rt_math.hpp against the compiler's IEEE
// operations:  rtm::rcp_ and rtm::sqrt_ on EVERY float32 bit pattern, rtm::div_mid on 4e10 random and structured quotients inside
// its guard, rtm::normalize on 2e10 random vectors (components from 0 / -0 / denormal to 2^70, NaN and inf included), and rtm::log_unit
// against rtm::log_ on every value RandomValue can return.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-fast-math -fno-slp-vectorize verify.hip -o verify          (the library's own flags)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../ray-tracing-extended_amd/csrc/rt_math.hpp"
using namespace rtm;
__device__ __forceinline__ bool same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }

__global__ void k_unary(unsigned long long* bad, uint32_t* first_bad)
{
    const uint64_t n = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += n) {
        const float x = u2f((uint32_t)i);
        if (!same(rcp_(x), 1.0f / x)) { if (atomicAdd(&bad[0], 1ull) == 0) first_bad[0] = (uint32_t)i; }
        if (!same(sqrt_(x), __builtin_sqrtf(x))) { if (atomicAdd(&bad[1], 1ull) == 0) first_bad[1] = (uint32_t)i; }
        if (x >= 1e-6f && x <= 0x1p100f && !same(rcp_mid(x), 1.0f / x)) atomicAdd(&bad[0], 1ull);     // the RayTriangle use
        // log_unit == log_ on everything RandomValue returns: +0 and [2^-32, 1]
        if (((uint32_t)i == 0u || (x >= 0x1p-32f && x <= 1.0f)) && !same(log_unit(x), log_(x))) { if (atomicAdd(&bad[4], 1ull) == 0) first_bad[7] = (uint32_t)i; }
    }
}
__device__ __forceinline__ uint32_t pcg(uint32_t& s) { s = s * 747796405u + 2891336453u; uint32_t r = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (r >> 22) ^ r; }
__device__ __forceinline__ uint32_t with_exp(uint32_t u, uint32_t e) { return (u & 0x807FFFFFu) | (e << 23); }
__global__ void k_div(unsigned long long* bad, uint32_t* first_bad, uint32_t rounds, uint32_t seed)
{
    uint32_t s = seed + (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    unsigned long long tested = 0;
    for (uint32_t it = 0; it < rounds; ++it) {
        uint32_t ua = with_exp(pcg(s), 67u + pcg(s) % 121u), ub = with_exp(pcg(s), 67u + pcg(s) % 121u);   // [2^-60, 2^60]
        float a = u2f(ua), b = u2f(ub);
        const uint32_t mode = it & 3u;
        if (mode == 1) a = b * (float)(int)(1u + pcg(s) % 4096u);                           // exact multiples
        else if (mode == 2) a = u2f(with_exp(ua, (ub >> 23) & 0xFFu));                        // same exponent: quotient in (0.5, 2)
        else if (mode == 3) { const float k = (float)(int)(1u + pcg(s) % 1000u); a = u2f(f2u(b * k) + (pcg(s) % 5u) - 2u); }   // next to multiples
        if ((it & 63u) == 63u) a = u2f(ua & 0x80000000u);                                     // +0 / -0 numerators (sign through the product)
        const float aa = __builtin_fabsf(a), ab = __builtin_fabsf(b);
        if (!((a == 0.0f || (aa >= 0x1p-60f && aa <= 0x1p60f)) && ab >= 0x1p-60f && ab <= 0x1p60f)) continue;
        ++tested;
        const float r = rcp_mid(b);
        if (!same(a == 0.0f ? a * r : div_mid(a, b, r), a / b)) { if (atomicAdd(&bad[2], 1ull) == 0) { first_bad[2] = f2u(a); first_bad[3] = f2u(b); } }
    }
    atomicAdd(&bad[5], tested);
}
__global__ void k_normalize(unsigned long long* bad, uint32_t* first_bad, uint32_t rounds, uint32_t seed)
{
    uint32_t s = seed + (blockIdx.x * blockDim.x + threadIdx.x) * 2246822519u;
    for (uint32_t it = 0; it < rounds; ++it) {
        float c[3];
        for (int k = 0; k < 3; ++k) {
            const uint32_t kind = pcg(s) % 16u, bits = pcg(s);
            if (kind == 0) c[k] = u2f(bits & 0x80000000u);                                  // +0 / -0
            else if (kind == 1) c[k] = u2f(bits & 0x807FFFFFu);                             // denormal
            else if (kind == 2) c[k] = u2f(with_exp(bits, 1u + pcg(s) % 254u));             // any normal exponent
            else if (kind == 3) c[k] = u2f(bits);                                           // any bit pattern (NaN, inf, ...)
            else if (kind < 8) c[k] = u2f(with_exp(bits, 60u + pcg(s) % 140u));             // 2^-67 .. 2^72
            else c[k] = u2f(with_exp(bits, 117u + pcg(s) % 12u));                           // around unit length, as in the tracer
        }
        const v3 a = mk(c[0], c[1], c[2]);
        const float len = __builtin_sqrtf(dot(a, a));
        const v3 got = normalize(a);
        if (!(same(got.x, a.x / len) && same(got.y, a.y / len) && same(got.z, a.z / len))) {
            if (atomicAdd(&bad[3], 1ull) == 0) { first_bad[4] = f2u(a.x); first_bad[5] = f2u(a.y); first_bad[6] = f2u(a.z); }
        }
    }
}
int main()
{
    unsigned long long* bad; uint32_t* fb;
    (void)hipMalloc(&bad, 8 * sizeof *bad); (void)hipMalloc(&fb, 8 * sizeof *fb);
    (void)hipMemset(bad, 0, 8 * sizeof *bad); (void)hipMemset(fb, 0, 8 * sizeof *fb);
    hipLaunchKernelGGL(k_unary, dim3(4096), dim3(256), 0, 0, bad, fb);
    const uint32_t rounds = 40000, nrounds = 20000;
    hipLaunchKernelGGL(k_div, dim3(4096), dim3(256), 0, 0, bad, fb, rounds, 12345u);
    hipLaunchKernelGGL(k_normalize, dim3(4096), dim3(256), 0, 0, bad, fb, nrounds, 777u);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; uint32_t hf[8];
    (void)hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, fb, sizeof hf, hipMemcpyDeviceToHost);
    printf("rcp_      mismatches over all 2^32 inputs: %llu (first bad input 0x%08x)\n", h[0], hf[0]);
    printf("sqrt_     mismatches over all 2^32 inputs: %llu (first bad input 0x%08x)\n", h[1], hf[1]);
    printf("div_mid   mismatches: %llu of %llu quotients inside the guard (first bad a=0x%08x b=0x%08x)\n", h[2], h[5], hf[2], hf[3]);
    printf("normalize mismatches: %llu of %.3g vectors (first bad 0x%08x 0x%08x 0x%08x)\n", h[3], 4096.0 * 256 * nrounds, hf[4], hf[5], hf[6]);
    printf("log_unit  mismatches against log_ over +0 and every float in [2^-32, 1]: %llu (first bad input 0x%08x)\n", h[4], hf[7]);
    return (h[0] | h[1] | h[2] | h[3] | h[4]) ? 1 : 0;
}
