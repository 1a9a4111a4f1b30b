// rt_math.hpp — device-side float32 arithmetic of the path tracer (gfx950).
//
// The reference's tracer is HLSL (Assets/Scripts/Shaders/RayTracing.shader); the meaning of its intrinsics
// (normalize, lerp, reflect, smoothstep, pow, log, cos, sin, min/max, saturate) is frozen in DESIGN.md
// "Frozen float semantics".  Everything here is written with + - * /, sqrt, floor and integer bit operations
// only, is compiled with -ffp-contract=off (no FMA formation) and IEEE-correct division / square root
// (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt, denormals kept), so the kernels produce the same
// bits on the GPU as a scalar IEEE-754 evaluation of the same expressions on any host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtm {

struct v3 { float x, y, z; };

__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float    u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

__device__ __forceinline__ v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// ---- correctly rounded 1/x, sqrt(x), a/b from the hardware estimates + FMA corrections ---------------------------------
// The frozen semantics are IEEE `/` and `sqrt`.  hipcc's correctly rounded expansions cost ~12 and ~17 VALU instructions; inside
// the guard ranges the short sequences below return the same bits — verified on gfx950 for EVERY float32 input of rcp_mid and
// sqrt_mid and for 4e10 random and structured quotients of div_mid (tools/exactmath/verify.hip; Markstein's quotient theorem
// covers the rest once r = RN(1/b)); outside the guards the compiler's IEEE sequence runs, so every input gives IEEE bits.
__device__ __forceinline__ bool mid_range(float x)          // |x| in [2^-100, 2^100]; false for NaN, inf, 0, denormals
{
    const float ax = __builtin_fabsf(x);
    return ax >= 0x1p-100f && ax <= 0x1p100f;
}
__device__ __forceinline__ float rcp_mid(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_mid(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
// The IEEE fallbacks must stay behind real branches: both sides are speculatable, and if-converted they would run for every
// lane next to the short sequence.  An (empty) volatile asm cannot be speculated, so the block it sits in stays a block.
#define RT_COLD_PATH() asm volatile("; RTCOLD")       /* (the comment marks the block for tools/static_valu.py; it assembles to nothing) */
__device__ __forceinline__ float rcp_(float x)                                                         // == 1.0f / x
{
    float r = rcp_mid(x);
    if (!mid_range(x)) { RT_COLD_PATH(); r = 1.0f / x; }
    return r;
}
__device__ __forceinline__ float sqrt_(float x)                                                        // == sqrtf(x)
{
    float r = sqrt_mid(x);
    if (!(x >= 0x1p-100f && x <= 0x1p100f)) { RT_COLD_PATH(); r = __builtin_sqrtf(x); }
    return r;
}
// a / b with r = rcp_mid(b): valid for |b| in [2^-60, 2^60] and |a| in [2^-60, 2^60] (quotient in [2^-120, 2^120])
__device__ __forceinline__ float div_mid(float a, float b, float r)
{
    const float q = a * r;
    const float e = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ v3 normalize(v3 a)              // == a / sqrt(dot(a, a)), three IEEE quotients
{
    const float len = sqrt_(dot(a, a));
    // a zero component gives the same signed zero (len > 0); a non-zero one must not be below 2^-60 (|bits| - 1 wraps for 0)
    const uint32_t kLo = (67u << 23) - 1u;
    const uint32_t mx = (f2u(a.x) & 0x7FFFFFFFu) - 1u, my = (f2u(a.y) & 0x7FFFFFFFu) - 1u, mz = (f2u(a.z) & 0x7FFFFFFFu) - 1u;
    const float r = rcp_mid(len);
    const float qx = div_mid(a.x, len, r), qy = div_mid(a.y, len, r), qz = div_mid(a.z, len, r);
    v3 q = mk(a.x == 0.0f ? a.x : qx, a.y == 0.0f ? a.y : qy, a.z == 0.0f ? a.z : qz);
    if (!(len >= 0x1p-60f && len <= 0x1p60f && min(min(mx, my), mz) >= kLo)) {
        RT_COLD_PATH();
        q = mk(a.x / len, a.y / len, a.z / len);
    }
    return q;
}
__device__ __forceinline__ v3 lerp(v3 a, v3 b, float t)
{
    return mk(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z));
}
__device__ __forceinline__ v3 reflect(v3 i, v3 n)
{
    float k = 2.0f * dot(i, n);
    return mk(i.x - k * n.x, i.y - k * n.y, i.z - k * n.z);
}

// min/max that return the non-NaN operand (the frozen meaning of HLSL min/max; v_min_f32/v_max_f32 order
// signed zeros differently, so the selects are spelled out).
__device__ __forceinline__ float fmin_(float a, float b) { return (a < b) ? a : ((b != b) ? a : b); }
__device__ __forceinline__ float fmax_(float a, float b) { return (a > b) ? a : ((b != b) ? a : b); }
__device__ __forceinline__ float saturate(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }

// ---- sin / cos: Cody-Waite quadrant reduction + Cephes single-precision minimax polynomials -----------
__device__ __forceinline__ float reduce_quadrant(float x, int& quadrant)
{
    const float TWO_OVER_PI = 0.636619772f;
    const float C1 = 1.5703125f;
    const float C2 = 4.837512969970703125e-4f;
    const float C3 = 7.54978995489188216e-8f;
    float q = __builtin_floorf(x * TWO_OVER_PI + 0.5f);
    float r = ((x - q * C1) - q * C2) - q * C3;
    quadrant = (int)q & 3;
    return r;
}
__device__ __forceinline__ float sin_poly(float r)
{
    float z = r * r;
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
}
__device__ __forceinline__ float cos_poly(float r)
{
    float z = r * r;
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
           - 0.5f * z + 1.0f;
}
__device__ __forceinline__ float sin_(float x)
{
    int n; float r = reduce_quadrant(x, n);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (n & 1) ? c : s;
    return (n & 2) ? -v : v;
}
__device__ __forceinline__ float cos_(float x)
{
    int n; float r = reduce_quadrant(x, n);
    float s = sin_poly(r), c = cos_poly(r);
    float v = (n & 1) ? s : c;
    return ((n + 1) & 2) ? -v : v;
}
// both at once (RandomPointInCircle needs the pair)
__device__ __forceinline__ void sincos_(float x, float& sn, float& cs)
{
    int n; float r = reduce_quadrant(x, n);
    float s = sin_poly(r), c = cos_poly(r);
    float vs = (n & 1) ? c : s;
    float vc = (n & 1) ? s : c;
    sn = (n & 2) ? -vs : vs;
    cs = ((n + 1) & 2) ? -vc : vc;
}

// ---- natural log (Cephes logf scheme) --------------------------------------------------------------------
__device__ __forceinline__ float log_(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return u2f(0xff800000u);
    uint32_t bits = f2u(x);
    if (bits == 0x7f800000u) return x;
    int e = 0;
    if (bits < 0x00800000u) { x = x * 8388608.0f; bits = f2u(x); e = -23; }
    e += (int)(bits >> 23) - 127;
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f
                   - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f - 1.6668057665e-1f) * f
                + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = f + y;
    r = r + 0.693359375f * fe;
    return r;
}
__device__ __forceinline__ float log2_(float x) { return log_(x) * 1.44269504f; }
// log_ for the values RandomValue returns — 0 and [2^-32, 1]: never NaN, negative, infinite or denormal.  The same operations in the
// same order as log_ on the path those inputs take (so the same bits: tools/exactmath/verify.hip compares the two for every such
// float); the special-case tests that can never fire are gone (a dozen VALU instructions per draw, three draws per hit).
__device__ __forceinline__ float log_unit(float x)
{
    const uint32_t bits = f2u(x);
    int e = (int)(bits >> 23) - 127;
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f
                   - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f - 1.6668057665e-1f) * f
                + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = f + y;
    r = r + 0.693359375f * fe;
    return x == 0.0f ? u2f(0xff800000u) : r;
}

// ---- 2^x ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float exp2_(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return u2f(0x7f800000u);
    if (x < -150.0f) return 0.0f;
    float k = __builtin_floorf(x + 0.5f);
    float f = x - k;
    float p = ((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f
                + 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f;
    float r = p * f + 1.0f;
    int ki = (int)k;
    int k1 = ki >> 1, k2 = ki - k1;
    r = r * u2f((uint32_t)(k1 + 127) << 23);
    r = r * u2f((uint32_t)(k2 + 127) << 23);
    return r;
}
__device__ __forceinline__ float pow_(float x, float y)
{
    if (x == 0.0f) return 0.0f;
    return exp2_(y * log2_(x));
}
__device__ __forceinline__ float smoothstep(float e0, float e1, float x)
{
    float t = saturate((x - e0) / (e1 - e0));
    return t * t * (3.0f - 2.0f * t);
}

// ---- PCG hash RNG — RayTracing.shader:193-204 ------------------------------------------------------------
__device__ __forceinline__ uint32_t next_random(uint32_t& state)
{
    state = state * 747796405u + 2891336453u;
    uint32_t result = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}
__device__ __forceinline__ float random_value(uint32_t& state)
{
    return (float)next_random(state) * 2.3283064365386963e-10f;   // r / 2^32 (the shader's 4294967295.0 is a float32 literal)
}
// ---- counter-based alternative (rt_params.rngMode = RT_RNG_PHILOX): Philox4x32-10, key (pixelIndex, Frame), counter (block, sample,
// 0, 0).  Every draw is addressed by what it is for — block 0: the sample's camera ray (4 draws); blocks 1 + 2b and 2 + 2b: the hit at
// loop index b of Trace (8 draws) — so the samples of a pixel are independent and can sit on different lanes (k_stream<.., PHILOX>;
// the definition is include/rt.h RT_RNG_PHILOX).  A PhiloxScope serves the draws of ONE such site: it lives in a straight-line
// region, n is a compile-time constant at every use after inlining, and nothing of the generator survives the region (no RNG state
// in the persistent loop's registers).
struct PhiloxScope {
    uint32_t k0, k1, c1, block, n;
    uint32_t w0, w1, w2, w3;
    __device__ __forceinline__ void begin(uint32_t pixelIndex, uint32_t frame, uint32_t sample, uint32_t first_block)
    {
        k0 = pixelIndex; k1 = frame; c1 = sample; block = first_block; n = 0; w0 = w1 = w2 = w3 = 0;
    }
    __device__ __forceinline__ void gen(uint32_t blk)
    {
        uint32_t c0 = blk, d1 = c1, c2 = 0, c3 = 0, a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            // the two 32 x 32 -> 64 products of a round, one v_mad_u64_u32 each: it issues like v_mul_lo_u32 or v_mul_hi_u32 alone (4.4 against
            // 4.7 + 4.2 cycles, profiles/ubench_r03_valu_rate6.txt) and the compiler emits the pair: a block 308 -> 237 cycles
            // (written as 64-bit products so that the compiler emits one v_mad_u64_u32 each; __umulhi + the 32-bit product gave two instructions)
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
            c0 = (uint32_t)(p1 >> 32) ^ d1 ^ a; d1 = (uint32_t)p1; c2 = (uint32_t)(p0 >> 32) ^ c3 ^ b; c3 = (uint32_t)p0;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        w0 = c0; w1 = d1; w2 = c2; w3 = c3;
    }
};
__device__ __forceinline__ float random_value(PhiloxScope& g)
{
    const uint32_t q = g.n & 3u;
    if (q == 0u) g.gen(g.block + (g.n >> 2));
    const uint32_t r = q == 0u ? g.w0 : q == 1u ? g.w1 : q == 2u ? g.w2 : g.w3;
    g.n++;
    return (float)r * 2.3283064365386963e-10f;
}

// RayTracing.shader:207-213
template <class R>
__device__ __forceinline__ float random_normal(R& state)
{
    const float TWO_PI = 2.0f * 3.1415926f;
    float theta = TWO_PI * random_value(state);
    float rho = sqrt_(-2.0f * log_unit(random_value(state)));        // (== log_ for every value RandomValue can return)
    return rho * cos_(theta);
}
// RayTracing.shader:216-223
template <class R>
__device__ __forceinline__ v3 random_direction(R& state)
{
    float x = random_normal(state);
    float y = random_normal(state);
    float z = random_normal(state);
    return normalize(mk(x, y, z));
}
// RayTracing.shader:225-230, PI = 3.1415 (:35)
template <class R>
__device__ __forceinline__ void random_point_in_circle(R& state, float& px, float& py)
{
    const float PI = 3.1415f;
    float angle = random_value(state) * 2.0f * PI;
    float sn, cs;
    sincos_(angle, sn, cs);
    float s = sqrt_(random_value(state));
    px = cs * s;
    py = sn * s;
}

} // namespace rtm
