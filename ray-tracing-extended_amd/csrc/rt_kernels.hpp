// rt_kernels.hpp — the CDNA4 path-trace megakernel and its validation twin.
//
// Path replaced: RayTracing.shader `frag` (:356-389) -> `Trace` (:300-352) -> `CalculateRayCollision`
// (:256-297) -> RaySphere / RayTriangle / RayBoundingBox (:120-187), followed by Accumulate.shader `frag`
// (:43-54), i.e. the two material blits of RayTracingManager.OnRenderImage (RayTracingManager.cs:74-81).
//
// Execution model (wave64, gfx950):
//   * one lane = one pixel, one wave = one 8x8 pixel tile; waves are persistent and pull tiles from an
//     atomic queue (wave granularity, no workgroup barrier anywhere);
//   * the PCG state of a pixel is a serial chain over all samples and bounces (RayTracing.shader:362,374-385),
//     so a lane keeps its pixel for the whole frame; inside the frame the lane runs a flat state machine —
//     every loop iteration is exactly one CalculateRayCollision plus its shading, and a lane whose path ended
//     immediately starts the pixel's next sample instead of idling until its neighbours finish theirs;
//   * closest hit = uniform sphere loop (scalar loads) + 4-wide BVH traversal, "while-while" shaped: lanes
//     descend internal nodes until each holds a leaf, then the wave tests leaves together.  The per-lane
//     traversal stack lives in LDS (stack[entry][lane], one bank per lane, conflict-free);
//   * accumulate (Accumulate.shader:43-54) is fused into the epilogue: 16 B read + 2 x 16 B write per pixel.
#pragma once
#include <cstddef>
#include <type_traits>
#include "rt_math.hpp"
#include "bvh.hpp"
#include "../../include/rt.h"

namespace rtk {

using rtm::v3;

struct DeviceScene {
    const float4* sph_geom;     // [ns]      (centre.xyz, radius)                         16 B
    const float4* sph_mat;      // [ns*4]    rt_material as 4 float4                      64 B
    const float4* nodes;        // [nn*8]    rtbvh::Node4                                128 B
    const float4* nodes_h;      // [nn*8]    rtbvh::Node4h (f16 planes around a per-node origin: 5 loads per visit instead of 7)
    const float4* tri_geo;      // [nt*3]    BVH order: (A, e1.x) (e1.yz, e2.xy) (e2.z, n.xyz) 48 B
    const float4* tri_nrm;      // [nt*3]    BVH order: (nA, chunk) (nB, visiting rank = tie-break key) (nC, -)   48 B
    const float4* chunk_mat;    // [nm*4]    rt_material of the chunk                     64 B
    const float4* chunk_box;    // [nm*2]    (boundsMin, -) (boundsMax, -)                32 B
    // raw reference buffers for the flat validation kernel
    const float*  raw_tris;     // [nt*18]
    const uint32_t* raw_chunk_range; // [nm*2] first, count
    int ns, nn, nt, nm;
};

struct FrameArgs {
    rt_params p;
    int frame;
    int row0, nrows;            // rows rendered by this launch: local row ly -> global row row0 + (ly/8)*row_stride + ly%8
    int row_stride;             // 8: one contiguous strip; 8*N: every N-th 8-row band (interleaved decomposition)
    int tiles_x, tiles_y;
    int frames_in_launch;       // k_trace: > 1 = this launch traces frames frame .. frame+n-1 into out_frame[f * frame_stride + pixel]
    unsigned int frame_stride;  //          and leaves the (ordered) accumulation to k_accumulate
    const uint32_t* tile_order; // k_trace: item -> tile permutation (costliest tiles first: LPT scheduling of the persistent waves); may be null
    uint32_t* tile_cost;        // k_trace: per-tile shader-clock cost of this launch (summed over its frames); may be null
    int tile_w_log2;            // k_trace: a wave's tile is 2^tile_w_log2 pixels wide and 64 >> tile_w_log2 rows tall (3: 8x8)
    int stack_cap;              // LDS stack entries per lane
    int full_sort;              // 1: sort all four children of a node by distance, 0: nearest first only
    int fixed_origin;           // 1: every camera ray starts exactly at worldSpaceCameraPos (defocusStrength is +-0 and the camera basis is
                                //    finite): camera_ray skips the arithmetic of the defocus jitter, whose result is pos + (+-0) = pos
    uint32_t* gstack;           // overflow of the traversal stack beyond stack_cap ([entry][lane of the launch]); may be null
    unsigned int gstack_stride; // lanes of the launch
    float4* out_frame;          // [nrows*W] currentFrame
    float4* accum;              // [nrows*W] resultTexture
    float* park;                // k_stream, Philox mode: per wave [items of a group][3][64] sub-stream sums waiting for the estimator's tree; else null
    const uint4* primary;       // k_stream: per local pixel, up to four leaf references a camera ray of that pixel starts with instead of the root
                                // (rt_primary.hpp: .x = 0xFFFFFFFE no list, kNone-terminated, all kNone = certain miss); null = every ray starts at the root
    const float4* focus;        // k_stream: per local pixel its focus point, computed once with the lists (the same float operations as the camera block's); null = compute
    unsigned int* tile_counter;
    unsigned long long* counters;   // [kNumCounters] rays, sphereTests, nodeVisits, triTests, hits, phase lanes[5], phase execs[5], region[32]
};

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kTriBit = 0x80000000u;

struct Hit {
    float    t;
    uint32_t id;      // kNone | sphere index | kTriBit + triangle index in BVH order
    float    u, v;    // barycentrics of the winning triangle
};

// Work counters (rays always; the rest in the counting build).  phase_lanes[k] / (64 * phase_execs[k]) is the lane
// utilisation of phase k (0 node step, 1 triangle test, 2 hit shading, 3 environment, 4 camera ray).
// Per-lane traversal stack: the first `cap` entries in LDS (stack[entry][lane]: one bank per lane, conflict-free), deeper
// entries — rare: the LDS part is sized for the depths rays actually reach — in a global overflow area.
struct TravStack {
    uint32_t* lds; uint32_t* glb; int cap; unsigned int stride;
    __device__ __forceinline__ void push(int& sp, uint32_t v) const
    {
        if (sp < cap) lds[sp * 64] = v; else glb[(size_t)(sp - cap) * stride] = v;
        ++sp;
    }
    __device__ __forceinline__ uint32_t pop(int& sp) const
    {
        --sp;
        return sp < cap ? lds[sp * 64] : glb[(size_t)(sp - cap) * stride];
    }
};

// wave-wide vote straight from the i1 (HIP's __ballot(int) round-trips the predicate through a VGPR: two half-rate VALU ops)
__device__ __forceinline__ unsigned long long ballot_(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
// vote on a && b: two compares into SGPR pairs and a scalar AND (the vote of an i1 that is not itself a compare is
// lowered through a VGPR again)
__device__ __forceinline__ unsigned long long ballot2_(bool a, bool b) { return __builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b); }

// Regions of k_stream (rt_stream.hpp).  The counting build counts how often a wave executes each of them (Counters::region,
// rt_stats.regionExecs); tools/static_valu.py counts the VALU instructions of each in the code object's assembly — the product of the two
// is the launch's VALU instruction count without a profiler pass (bench.py roofline.valu_model).  A region is what lies between
// RT_REGION_BEGIN and RT_REGION_END in the source; the innermost region owns an instruction.  `loop` is everything outside the others.
#define RT_REGION_LIST(X) X(loop) X(fetch) X(shade) X(hit) X(hit_sphere) X(hit_checker) X(hit_scatter) X(env) X(env_sun) X(camera) X(pixel_done) \
                          X(take) X(refill) X(setup) X(setup_list) X(burst) X(burstiter) X(nodeloop) X(node) X(node_spill) X(node_pop) X(leaf) X(tri) \
                          X(tri_accept) X(tri_tie) X(tri_chunk) X(prologue) X(verify) X(epilogue) X(camera_dof) X(camera_focus) X(setup_spheres)
enum Region : int {
#define RT_X(NAME) R_##NAME,
    RT_REGION_LIST(RT_X)
#undef RT_X
    kRegionsUsed
};
constexpr int kNumRegions = 32;
static_assert(kRegionsUsed <= kNumRegions, "rt_stats.regionExecs holds 32 regions");
struct Counters { uint32_t rays, sph, nodes, tris, hits; uint32_t phase_lanes[5], phase_execs[5]; uint32_t region[kNumRegions]; };
constexpr int kNumCounters = 15 + kNumRegions;

// Region markers for tools/static_valu.py: an assembler comment in a side-effect asm statement (it assembles to nothing).  Only the -S
// compile of that tool defines RT_MARKERS; the product library is built without them.
#ifdef RT_MARKERS
#define RT_MARK(TEXT) asm volatile("; RTMARK " TEXT)
#define RT_RARE_PATH() asm volatile("; RTRARE")      /* a block the benchmark configurations do not execute (option off, stack within its LDS part) */
#define RT_RARE_PATH_EXPR() ({ asm volatile("; RTRARE"); 0; })                     /* the same inside a condition: (RT_RARE_PATH_EXPR(), test) */
#else
#define RT_MARK(TEXT) do { } while (0)
#define RT_RARE_PATH() do { } while (0)
#define RT_RARE_PATH_EXPR() 0
#endif
// (COUNT and cnt are the enclosing kernel's; the tick is one increment on the first active lane, in the counting build only)
#define RT_REGION_BEGIN(NAME) do { RT_MARK("begin " #NAME); if (COUNT) { const unsigned long long rm_ = rtk::ballot_(true); \
                                   if ((unsigned)__builtin_ctzll(rm_) == (threadIdx.x & 63u)) cnt.region[rtk::R_##NAME]++; } } while (0)
#define RT_REGION_END(NAME) RT_MARK("end " #NAME)

template <bool COUNT>
__device__ __forceinline__ void phase_tick(Counters& cnt, int k)
{
    if (COUNT) {
        cnt.phase_lanes[k]++;
        const unsigned long long m = ballot_(1);
        if ((unsigned)(__builtin_ctzll(m)) == (threadIdx.x & 63u)) cnt.phase_execs[k]++;
    }
}

__device__ __forceinline__ v3 ld3(const float* p) { return rtm::mk(p[0], p[1], p[2]); }

// RayBoundingBox — RayTracing.shader:177-187, literal arithmetic (used for the chunk filter)
__device__ __forceinline__ bool ray_bounding_box(v3 o, v3 inv, v3 bmin, v3 bmax)
{
    float tminx = (bmin.x - o.x) * inv.x, tminy = (bmin.y - o.y) * inv.y, tminz = (bmin.z - o.z) * inv.z;
    float tmaxx = (bmax.x - o.x) * inv.x, tmaxy = (bmax.y - o.y) * inv.y, tmaxz = (bmax.z - o.z) * inv.z;
    float t1x = rtm::fmin_(tminx, tmaxx), t1y = rtm::fmin_(tminy, tmaxy), t1z = rtm::fmin_(tminz, tmaxz);
    float t2x = rtm::fmax_(tminx, tmaxx), t2y = rtm::fmax_(tminy, tmaxy), t2z = rtm::fmax_(tminz, tmaxz);
    float tNear = rtm::fmax_(rtm::fmax_(t1x, t1y), t1z);
    float tFar  = rtm::fmin_(rtm::fmin_(t2x, t2y), t2z);
    return tNear <= tFar;
}

// RaySphere — RayTracing.shader:120-146.  a = dot(dir, dir) and the denominator 2a are the same for every sphere a ray meets:
// its correctly rounded reciprocal is taken once per ray, and the quotient (-b - sqrt(disc)) / (2a) is Markstein's (rt_math.hpp)
// when both operands are in the verified range, the compiler's IEEE division otherwise — the same bits either way.
struct SphereA { float a, den, rden; bool den_ok; };
__device__ __forceinline__ SphereA sphere_a(float a)
{
    SphereA s; s.a = a; s.den = 2.0f * a;
    s.den_ok = s.den >= 0x1p-60f && s.den <= 0x1p60f;
    s.rden = rtm::rcp_mid(s.den);
    return s;
}
__device__ __forceinline__ bool ray_sphere(v3 o, v3 d, const SphereA& sa, v3 centre, float radius, float& dst)
{
    v3 oc = o - centre;
    float b = 2.0f * rtm::dot(oc, d);
    float c = rtm::dot(oc, oc) - radius * radius;
    float disc = b * b - 4.0f * sa.a * c;
    if (disc >= 0.0f) {
        const float num = -b - rtm::sqrt_(disc);
        const float an = __builtin_fabsf(num);
        const float q0 = num * sa.rden;
        float q = __builtin_fmaf(__builtin_fmaf(-q0, sa.den, num), sa.rden, q0);
        q = (num == 0.0f) ? q0 : q;                               // a zero numerator keeps its sign through the product
        if (!(sa.den_ok && (num == 0.0f || (an >= 0x1p-60f && an <= 0x1p60f)))) { RT_COLD_PATH(); q = num / sa.den; }
        dst = q;
        return dst >= 0.0f;
    }
    return false;
}

// RayTriangle — RayTracing.shader:150-174 with edgeAB, edgeAC and their cross product precomputed on the host
// by the same float operations (upload re-layout).
__device__ __forceinline__ bool ray_triangle(v3 o, v3 d, v3 A, v3 eAB, v3 eAC, v3 n, float& dst, float& u, float& v)
{
    v3 ao = o - A;
    v3 dao = rtm::cross(ao, d);
    float det = -rtm::dot(d, n);
    // 1/det: exact for det in [2^-100, 2^100]; a det below 1e-6 (or NaN) never hits, so its reciprocal is never looked at
    float inv = rtm::rcp_mid(det);
    if (det > 0x1p100f) { RT_COLD_PATH(); inv = 1.0f / det; }
    dst = rtm::dot(ao, n) * inv;
    u = rtm::dot(eAC, dao) * inv;
    v = -rtm::dot(eAB, dao) * inv;
    float w = 1.0f - u - v;
    return det >= 1e-6f && dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
}


// ---- BVH4 node step ---------------------------------------------------------------------------------------
// Per ray: the byte offsets (inside a 128-B node) of the float4 holding each axis' NEAR planes, chosen by the
// sign of the direction component (far = the other one); t = plane * inv - o*inv is one FMA per plane — the
// hierarchy only prunes, so its arithmetic is free to differ from the reference's (boxes are padded for it:
// bvh.cpp pad_box).  Measured on MI355X (tools/ubench/valu_rate.hip): v_fma/v_mul/v_add issue at ~2.5 cycles
// per wave, v_min/v_max/v_cmp/v_cndmask at ~4.2 — so the slab test avoids per-axis min/max altogether.
template <bool H> struct RaySlabT;
template <> struct RaySlabT<false> {
    v3 inv;             // 1 / d  (RayBoundingBox :179 — also used by the literal chunk filter)
    v3 oinv;            // o * inv
    uint32_t nx, ny, nz;   // byte offsets of the near-plane float4s (x: 0|48, y: 16|64, z: 32|80)
    uint32_t fx, fy, fz;   // ... and of the far-plane ones (the other of each pair)
};
// Node4h (bvh.hpp): the ray's signs pick one of four 16-B (x, y) plane sets as its near planes, the complementary set as its
// far planes, and one of two 16-B (near z, far z) sets
template <> struct RaySlabT<true> {
    v3 inv, oinv;
    uint32_t nxy, fxy, zo; // byte offsets inside the node: near (x,y) set, far (x,y) set, z set
};
using RaySlab = RaySlabT<false>;

template <bool H = false>
__device__ __forceinline__ RaySlabT<H> make_slab(v3 o, v3 d)
{
    RaySlabT<H> r;
    r.inv = rtm::mk(rtm::rcp_(d.x), rtm::rcp_(d.y), rtm::rcp_(d.z));
    r.oinv = rtm::mk(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
    if constexpr (H) {
        r.nxy = (d.x < 0.0f ? 16u : 0u) + (d.y < 0.0f ? 32u : 0u);
        r.fxy = 48u - r.nxy;
        r.zo = d.z < 0.0f ? 80u : 64u;
    } else {
        r.nx = d.x < 0.0f ? 48u : 0u;
        r.ny = d.y < 0.0f ? 64u : 16u;
        r.nz = d.z < 0.0f ? 80u : 32u;
        r.fx = 48u - r.nx; r.fy = 80u - r.ny; r.fz = 112u - r.nz;
    }
    return r;
}

// Tests the four child boxes of node `cur`, returns them sorted by entry distance (t0 <= t1 <= t2 <= t3, misses
// carry t = +inf).  Empty slots hold (+inf, -inf) boxes and can never be hit.
typedef _Float16 rt_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rt_half2 as_half2(uint32_t u) { return __builtin_bit_cast(rt_half2, u); }

template <bool H>
__device__ __forceinline__ void node_step(const float4* __restrict__ nodes, uint32_t cur, const RaySlabT<H>& r, float best_t, bool full_sort,
                                          float& t0, float& t1, float& t2, float& t3,
                                          uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3)
{
    // 32-bit byte offsets from the (wave-uniform) node array: one shift and a few adds per node instead of 64-bit address
    // arithmetic per load (the host refuses BVHs of 2^25 nodes or more)
    const char* nb = reinterpret_cast<const char*>(nodes);
    const uint32_t base = cur << 7;
    const float INF = __builtin_inff();
    if constexpr (H) {
        const uint4 na = *reinterpret_cast<const uint4*>(nb + (base + r.nxy)), fa = *reinterpret_cast<const uint4*>(nb + (base + r.fxy));
        const uint4 zz = *reinterpret_cast<const uint4*>(nb + (base + r.zo));
        const uint4 ch = *reinterpret_cast<const uint4*>(nb + (base + 96u));
        const float4 og = *reinterpret_cast<const float4*>(nb + (base + 112u));
        c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
#ifdef RT_PROBE_LOADS       // sensitivity probes (never in the product build; tools/build_variant.py): extra dwordx4 loads per node step ...
        // (volatile C++ loads, not inline asm: the compiler's s_waitcnt bookkeeping does not see a load inside an asm, the registers it
        // returns into are handed to other values at once and the late write-back corrupts them — the round-2 load probe had that fault)
        for (int k_ = 0; k_ < RT_PROBE_LOADS; ++k_) (void)*reinterpret_cast<const volatile uint32_t*>(nb + (base + 124u));
#endif
#ifdef RT_PROBE_VALU        // ... extra full-rate VALU instructions (v_fma_f32) ...
        { float pa_ = r.inv.x; for (int k_ = 0; k_ < RT_PROBE_VALU; ++k_) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pa_)); }
#endif
#ifdef RT_PROBE_VALU_HALF   // ... or extra half-rate ones (v_max_f32)
        { float pa_ = r.inv.x; for (int k_ = 0; k_ < RT_PROBE_VALU_HALF; ++k_) asm volatile("v_max_f32 %0, %0, %0" : "+v"(pa_)); }
#endif
        // t = (origin + offset) * inv - o*inv = offset * inv + (origin * inv - o*inv): one f32 FMA per axis and node, then one
        // v_fma_mix_f32 (f16 offset x f32 + f32) per plane
        const float kx = __builtin_fmaf(og.x, r.inv.x, -r.oinv.x), ky = __builtin_fmaf(og.y, r.inv.y, -r.oinv.y), kz = __builtin_fmaf(og.z, r.inv.z, -r.oinv.z);
#define RT_SLABH(XW, YW, ZNW, ZFW, E, TK)                                                                                 \
        {                                                                                                                \
            const float nx_ = __builtin_fmaf((float)as_half2(na.XW).E, r.inv.x, kx), fx_ = __builtin_fmaf((float)as_half2(fa.XW).E, r.inv.x, kx); \
            const float ny_ = __builtin_fmaf((float)as_half2(na.YW).E, r.inv.y, ky), fy_ = __builtin_fmaf((float)as_half2(fa.YW).E, r.inv.y, ky); \
            const float nz_ = __builtin_fmaf((float)as_half2(zz.ZNW).E, r.inv.z, kz), fz_ = __builtin_fmaf((float)as_half2(zz.ZFW).E, r.inv.z, kz); \
            const float tn_ = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx_, ny_), nz_), 0.0f);                    \
            const float tf_ = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx_, fy_), fz_), best_t);                  \
            TK = (tn_ <= tf_) ? tn_ : INF;                                                                               \
        }
        RT_SLABH(x, z, x, z, x, t0) RT_SLABH(x, z, x, z, y, t1) RT_SLABH(y, w, y, w, x, t2) RT_SLABH(y, w, y, w, y, t3)
#undef RT_SLABH
    } else {
    const float4 px = *reinterpret_cast<const float4*>(nb + (base + r.nx)), qx = *reinterpret_cast<const float4*>(nb + (base + r.fx));
    const float4 py = *reinterpret_cast<const float4*>(nb + (base + r.ny)), qy = *reinterpret_cast<const float4*>(nb + (base + r.fy));
    const float4 pz = *reinterpret_cast<const float4*>(nb + (base + r.nz)), qz = *reinterpret_cast<const float4*>(nb + (base + r.fz));
    const uint4 ch = *reinterpret_cast<const uint4*>(nb + (base + 96u));
    c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
#ifdef RT_PROBE_LOADS       // sensitivity probe (never in the product build): RT_PROBE_LOADS extra dwordx4 loads per node step
    for (int k_ = 0; k_ < RT_PROBE_LOADS; ++k_) (void)*reinterpret_cast<const volatile uint32_t*>(nb + (base + 124u));
#endif
#ifdef RT_PROBE_VALU        // ... or RT_PROBE_VALU extra full-rate VALU instructions
    { float pa_ = r.inv.x; for (int k_ = 0; k_ < RT_PROBE_VALU; ++k_) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pa_)); }
#endif
#define RT_SLAB(K, TK)                                                                                                   \
    {                                                                                                                    \
        const float nx_ = __builtin_fmaf(px.K, r.inv.x, -r.oinv.x), fx_ = __builtin_fmaf(qx.K, r.inv.x, -r.oinv.x);      \
        const float ny_ = __builtin_fmaf(py.K, r.inv.y, -r.oinv.y), fy_ = __builtin_fmaf(qy.K, r.inv.y, -r.oinv.y);      \
        const float nz_ = __builtin_fmaf(pz.K, r.inv.z, -r.oinv.z), fz_ = __builtin_fmaf(qz.K, r.inv.z, -r.oinv.z);      \
        const float tn_ = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx_, ny_), nz_), 0.0f);                        \
        const float tf_ = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx_, fy_), fz_), best_t);                      \
        TK = (tn_ <= tf_) ? tn_ : INF;                                                                                   \
    }
    RT_SLAB(x, t0) RT_SLAB(y, t1) RT_SLAB(z, t2) RT_SLAB(w, t3)
#undef RT_SLAB
    }
#define RT_CSWAP(TA, CA, TB, CB) { const bool s_ = TB < TA; const float tt_ = s_ ? TB : TA, tu_ = s_ ? TA : TB;         \
                                   const uint32_t ct_ = s_ ? CB : CA, cu_ = s_ ? CA : CB; TA = tt_; TB = tu_; CA = ct_; CB = cu_; }
    // the nearest child must come first; a full sort of the other three (2 more exchanges) only refines the order in
    // which they are popped later
    RT_CSWAP(t0, c0, t1, c1) RT_CSWAP(t2, c2, t3, c3) RT_CSWAP(t0, c0, t2, c2)
    if (full_sort) { RT_RARE_PATH(); RT_CSWAP(t1, c1, t3, c3) RT_CSWAP(t1, c1, t2, c2) }
#undef RT_CSWAP
}

// The three float4s of BVH-order triangle `ti` (A, eAB, eAC, n) through a 32-bit byte offset from the wave-uniform array
// (the host refuses scenes of 2^32 / 48 triangles or more).
__device__ __forceinline__ void load_tri(const float4* __restrict__ tri_geo, uint32_t ti, float4& g0, float4& g1, float4& g2)
{
    const char* tb = reinterpret_cast<const char*>(tri_geo);
    const uint32_t off = ti * 48u;
    g0 = *reinterpret_cast<const float4*>(tb + off);
    g1 = *reinterpret_cast<const float4*>(tb + (off + 16u));
    g2 = *reinterpret_cast<const float4*>(tb + (off + 32u));
#ifdef RT_PROBE_LEAF_LOADS  // the same probes for a triangle test
    for (int k_ = 0; k_ < RT_PROBE_LEAF_LOADS; ++k_) (void)*reinterpret_cast<const volatile uint32_t*>(tb + (off + 44u));
#endif
#ifdef RT_PROBE_LEAF_VALU
    { float pa_ = g0.x; for (int k_ = 0; k_ < RT_PROBE_LEAF_VALU; ++k_) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pa_)); }
#endif
}

// A ray with a NaN in it, or a zero direction, can hit nothing (every RaySphere / RayTriangle comparison is false), but its
// slab tests are all NaN too and would "enter" every child, empty slots included: such a query is complete without traversal.
__device__ __forceinline__ bool ray_traceable(v3 o, v3 d, float dd)
{
    return o.x == o.x && o.y == o.y && o.z == o.z && dd == dd && !(d.x == 0.0f && d.y == 0.0f && d.z == 0.0f);
}

// ---- closest hit: spheres, then BVH ---------------------------------------------------------------------
template <bool COUNT, bool H = false>
__device__ __forceinline__ Hit closest_hit(const DeviceScene& S, int intersect_mode, bool full_sort, v3 o, v3 d,
                                           const TravStack& stk, Counters& cnt)
{
    Hit best; best.t = __builtin_inff(); best.id = kNone; best.u = 0.f; best.v = 0.f;
    cnt.rays++;                                  // rays are always counted (one add per cast)

    // CalculateRayCollision :263-273 — buffer order, strict '<' (first sphere wins ties)
    const float a = rtm::dot(d, d);
    const SphereA sa = sphere_a(a);
    // (four, then two sphere records per trip: the compiler waits for every load right where it issues it, so a sphere test otherwise
    // pays the full latency of its own 16 bytes.  Sphere workload: 40.23 one by one, 41.28 by two, 41.59 Grays/s by four)
    auto test = [&](const float4 s, int i) {
        float dst;
        if (COUNT) cnt.sph++;
        if (ray_sphere(o, d, sa, rtm::mk(s.x, s.y, s.z), s.w, dst) && dst < best.t) { best.t = dst; best.id = (uint32_t)i; }
    };
    int i = 0;
    for (; i + 3 < S.ns; i += 4) {
        const float4 s0 = S.sph_geom[i], s1 = S.sph_geom[i + 1], s2 = S.sph_geom[i + 2], s3 = S.sph_geom[i + 3];
        test(s0, i); test(s1, i + 1); test(s2, i + 2); test(s3, i + 3);
    }
    for (; i + 1 < S.ns; i += 2) {
        const float4 s0 = S.sph_geom[i], s1 = S.sph_geom[i + 1];
        test(s0, i);
        test(s1, i + 1);
    }
    for (; i < S.ns; ++i) test(S.sph_geom[i], i);

    if (S.nn > 0 && ray_traceable(o, d, a)) {
        const RaySlabT<H> slab = make_slab<H>(o, d);
        const v3 inv = slab.inv;                                        // RayBoundingBox :179
        int sp = 0;
        uint32_t cur = 0;                                               // root
        // The top of the stack lives in a register: a pop hands it out at once and refills it from LDS, so the LDS read's
        // latency falls before the *next* pop instead of in front of the next node's address.
        uint32_t top = kNone;
#define RT_PUSH(X) { if (top != kNone) stk.push(sp, top); top = (X); }
#define RT_POP()   { cur = top; top = (sp > 0) ? stk.pop(sp) : kNone; }
        while (cur != kNone) {
            // ---- descend internal nodes until this lane holds a leaf (or ran dry)
            while ((int)cur >= 0) {
                if (COUNT) cnt.nodes++;
                phase_tick<COUNT>(cnt, 0);
                float t0, t1, t2, t3;
                uint32_t c0, c1, c2, c3;
                node_step<H>(H ? S.nodes_h : S.nodes, cur, slab, best.t, full_sort, t0, t1, t2, t3, c0, c1, c2, c3);
                const float INF = __builtin_inff();
                if (t3 < INF) RT_PUSH(c3)
                if (t2 < INF) RT_PUSH(c2)
                if (t1 < INF) RT_PUSH(c1)
                if (t0 < INF) cur = c0;
                else RT_POP()
            }
            // ---- leaf
            if (cur != kNone) {
                const uint32_t first = (cur & 0x7FFFFFFFu) >> 2, count = (cur & 3u) + 1u;
                for (uint32_t j = 0; j < count; ++j) {
                    const uint32_t ti = first + j;
                    float4 g0, g1, g2;
                    load_tri(S.tri_geo, ti, g0, g1, g2);
                    float dst, u, v;
                    if (COUNT) cnt.tris++;
                    phase_tick<COUNT>(cnt, 1);
                    bool hit = ray_triangle(o, d, rtm::mk(g0.x, g0.y, g0.z), rtm::mk(g0.w, g1.x, g1.y),
                                            rtm::mk(g1.z, g1.w, g2.x), rtm::mk(g2.y, g2.z, g2.w), dst, u, v);
                    if (hit && dst <= best.t) {
                        bool take = dst < best.t;
                        if (!take && (best.id & kTriBit) && best.id != kNone) {
                            // equal dst: the reference keeps the triangle that comes first in the buffer
                            uint32_t oc = __float_as_uint(S.tri_nrm[(size_t)ti * 3 + 1].w);
                            uint32_t ob = __float_as_uint(S.tri_nrm[(size_t)(best.id & ~kTriBit) * 3 + 1].w);
                            take = oc < ob;
                        }
                        if (take && intersect_mode == RT_INTERSECT_FLAT_CHUNKS) {
                            // the reference only reaches this triangle if its chunk's box test passes (:279)
                            uint32_t chunk = __float_as_uint(S.tri_nrm[(size_t)ti * 3].w);
                            float4 bmn = S.chunk_box[(size_t)chunk * 2], bmx = S.chunk_box[(size_t)chunk * 2 + 1];
                            take = ray_bounding_box(o, inv, rtm::mk(bmn.x, bmn.y, bmn.z), rtm::mk(bmx.x, bmx.y, bmx.z));
                        }
                        if (take) { best.t = dst; best.id = kTriBit | ti; best.u = u; best.v = v; }
                    }
                }
                RT_POP()
            }
        }
#undef RT_PUSH
#undef RT_POP
    }
    if (COUNT && best.id != kNone) cnt.hits++;
    return best;
}

// ---- the reference's own flat loop (validation twin): every lane walks all chunks / triangles ------------
__device__ __forceinline__ Hit closest_hit_flat(const DeviceScene& S, int intersect_mode, v3 o, v3 d,
                                                v3& nrm_out, uint32_t& chunk_out)
{
    Hit best; best.t = __builtin_inff(); best.id = kNone; best.u = 0.f; best.v = 0.f;
    const float a = rtm::dot(d, d);
    const SphereA sa = sphere_a(a);
    for (int i = 0; i < S.ns; ++i) {
        float4 s = S.sph_geom[i];
        float dst;
        if (ray_sphere(o, d, sa, rtm::mk(s.x, s.y, s.z), s.w, dst) && dst < best.t) { best.t = dst; best.id = (uint32_t)i; }
    }
    const v3 inv = rtm::mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    nrm_out = rtm::mk(0.f, 0.f, 0.f); chunk_out = 0;
    for (int m = 0; m < S.nm; ++m) {
        if (intersect_mode == RT_INTERSECT_FLAT_CHUNKS) {
            float4 bmn = S.chunk_box[(size_t)m * 2], bmx = S.chunk_box[(size_t)m * 2 + 1];
            if (!ray_bounding_box(o, inv, rtm::mk(bmn.x, bmn.y, bmn.z), rtm::mk(bmx.x, bmx.y, bmx.z))) continue;
        }
        const uint32_t first = S.raw_chunk_range[2 * m], count = S.raw_chunk_range[2 * m + 1];
        for (uint32_t j = 0; j < count; ++j) {
            const float* t = S.raw_tris + (size_t)(first + j) * 18;
            v3 A = ld3(t), eAB = ld3(t + 3) - A, eAC = ld3(t + 6) - A;
            v3 n = rtm::cross(eAB, eAC);
            float dst, u, v;
            if (ray_triangle(o, d, A, eAB, eAC, n, dst, u, v) && dst < best.t) {
                best.t = dst; best.id = kTriBit | (first + j); best.u = u; best.v = v;
                float w = 1.0f - u - v;
                nrm_out = (ld3(t + 9) * w + ld3(t + 12) * u) + ld3(t + 15) * v;
                chunk_out = (uint32_t)m;
            }
        }
    }
    return best;
}

// GetEnvironmentLight — RayTracing.shader:238-251
__device__ __forceinline__ v3 environment_light(const rt_params& p, v3 d)
{
    if (!p.environmentEnabled) return rtm::mk(0.f, 0.f, 0.f);
    float skyGradientT = rtm::pow_(rtm::smoothstep(0.0f, 0.4f, d.y), 0.35f);
    float groundToSkyT = rtm::smoothstep(-0.01f, 0.0f, d.y);
    v3 skyGradient = rtm::lerp(ld3(p.skyColourHorizon), ld3(p.skyColourZenith), skyGradientT);
    // sunIntensity == +0 (e.g. Chess.unity:30185): the base max(0, dot) is finite, >= 0 and at most 1 + a few ulp, so for
    // 1 <= sunFocus <= 1e6 pow() is finite and >= 0 and the product is exactly +0; skip the transcendentals, keep the adds.
    float sun = 0.0f;
    if (!(__float_as_uint(p.sunIntensity) == 0u && p.sunFocus >= 1.0f && p.sunFocus <= 1.0e6f)) {     // (wave-uniform: parameters only)
        RT_MARK("begin env_sun");
        sun = rtm::pow_(rtm::fmax_(0.0f, rtm::dot(d, ld3(p.worldSpaceLightPos0))), p.sunFocus) * p.sunIntensity;
        RT_MARK("end env_sun");
    }
    v3 composite = rtm::lerp(ld3(p.groundColour), skyGradient, groundToSkyT);
    float sunTerm = sun * ((groundToSkyT >= 1.0f) ? 1.0f : 0.0f);
    return rtm::mk(composite.x + sunTerm, composite.y + sunTerm, composite.z + sunTerm);
}

__device__ __forceinline__ float mod2(float x) { return x - 2.0f * __builtin_floorf(x / 2.0f); }

struct Camera { v3 focusPoint, right, up, pos; float W; };

// A *fresh* view of a kernel's argument segment (KA = the kernel's parameters as one struct, in order).  The megakernels are
// persistent loops with regions that need different arguments (traversal: node / triangle arrays; shading: materials, camera,
// environment; scheduling: queue and tile tables).  Read through the kernel's own parameters they are all loop invariants: the
// compiler keeps every one of them in SGPRs across the whole loop, runs out (k_stream: 86 SGPRs spilled to VGPR lanes) and pays
// a VALU v_readlane_b32 per use — on the port that bounds the kernel.  A region that starts with fresh_kernargs() re-reads what it
// needs from the kernarg segment with scalar loads instead (the pointer goes through an empty asm, so nothing read through it can
// be hoisted out of the region) and the values die with the region.
template <class KA>
__device__ __forceinline__ const KA& fresh_kernargs()
{
    auto kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const KA*)(const char*)kp;
}
static_assert(alignof(DeviceScene) <= 8 && alignof(FrameArgs) <= 8, "kernarg layout = struct layout");
struct TraceKernArgs { DeviceScene S; FrameArgs F; };                     // k_trace(DeviceScene, FrameArgs)
static_assert(offsetof(TraceKernArgs, F) == ((sizeof(DeviceScene) + 7) & ~size_t(7)), "kernarg layout = struct layout");
// (k_trace's pixel loop is one shading-dominated region: the same treatment measured -0.8 % on the sphere workload and is not applied.)

// frag :377-382 — one camera ray (4 RNG draws)
// fixed_origin (wave-uniform, decided by the host: FrameArgs::fixed_origin): defocusStrength is +-0, so the jitter is
// (finite * +-0) / W = +-0, right * +-0 and up * +-0 are +-0 (finite basis) and pos + (+-0) = pos bit for bit (pos finite and no
// component -0, the one value a +0 would change).  The two draws of the jitter are still taken from the stream.
template <class R>
__device__ __forceinline__ void camera_ray(const rt_params& p, const Camera& c, R& rng, v3& o, v3& d, bool fixed_origin = false)
{
    float jx, jy;
    if (fixed_origin) {
        (void)rtm::random_value(rng); (void)rtm::random_value(rng);
        o = c.pos;
    } else {
        RT_MARK("begin camera_dof");        // (wave-uniform: the host's fixed_origin decision)
        rtm::random_point_in_circle(rng, jx, jy);
        jx = jx * p.defocusStrength / c.W; jy = jy * p.defocusStrength / c.W;
        o = (c.pos + c.right * jx) + c.up * jy;
        RT_MARK("end camera_dof");
    }
    rtm::random_point_in_circle(rng, jx, jy);
    jx = jx * p.divergeStrength / c.W; jy = jy * p.divergeStrength / c.W;
    v3 jfp = (c.focusPoint + c.right * jx) + c.up * jy;
    d = rtm::normalize(jfp - o);
}

// One pixel of one frame: frag :356-389 as a flat state machine (see file header).
template <bool COUNT, bool FLAT, bool H = false>
__device__ __forceinline__ v3 render_pixel(const DeviceScene& S, const rt_params& p, bool full_sort, bool fixed_origin, int frame, int x, int y,
                                           const TravStack& stk, Counters& cnt)
{
    const float* M = p.camLocalToWorld;
    const uint32_t W = (uint32_t)p.width;
    Camera cam;
    cam.W = (float)W;
    const float uvx = ((float)x + 0.5f) / cam.W, uvy = ((float)y + 0.5f) / (float)(uint32_t)p.height;
    uint32_t rng = ((uint32_t)y * W + (uint32_t)x) + (uint32_t)frame * 719393u;             // :361-362 (the reference's PCG stream; the
                                                                                            // counter-based mode is k_stream's, rt_stream.hpp)
    {
        float lx = (uvx - 0.5f) * p.viewParams[0], ly = (uvy - 0.5f) * p.viewParams[1], lz = 1.0f * p.viewParams[2];
        cam.focusPoint = rtm::mk(((M[0] * lx + M[1] * ly) + M[2]  * lz) + M[3]  * 1.0f,
                                 ((M[4] * lx + M[5] * ly) + M[6]  * lz) + M[7]  * 1.0f,
                                 ((M[8] * lx + M[9] * ly) + M[10] * lz) + M[11] * 1.0f);
    }
    cam.right = rtm::mk(M[0], M[4], M[8]);
    cam.up    = rtm::mk(M[1], M[5], M[9]);
    cam.pos   = ld3(p.worldSpaceCameraPos);

    v3 total = rtm::mk(0.f, 0.f, 0.f);
    v3 o, d;
    v3 rayColour = rtm::mk(1.f, 1.f, 1.f), light = rtm::mk(0.f, 0.f, 0.f);
    int sample = 0, bounce = 0;
    bool alive = p.numRaysPerPixel > 0;
    if (alive) camera_ray(p, cam, rng, o, d, fixed_origin);

    while (alive) {
        Hit h; v3 nrm_flat; uint32_t chunk_flat;
        if (FLAT) { h = closest_hit_flat(S, p.intersectMode, o, d, nrm_flat, chunk_flat); cnt.rays++; }
        else      h = closest_hit<COUNT, H>(S, p.intersectMode, full_sort, o, d, stk, cnt);

        bool path_done;
        if (h.id != kNone) {
            // ---- hit: Trace :309-343
            phase_tick<COUNT>(cnt, 2);
            const v3 hitPoint = o + d * h.t;
            v3 normal; const float4* mat;
            if (h.id & kTriBit) {
                const uint32_t ti = h.id & ~kTriBit;
                if (FLAT) {
                    normal = rtm::normalize(nrm_flat);
                    mat = S.chunk_mat + (size_t)chunk_flat * 4;
                } else {
                    const float4* tn = S.tri_nrm + (size_t)ti * 3;
                    const float4 n0 = tn[0], n1 = tn[1], n2 = tn[2];
                    const float w = 1.0f - h.u - h.v;
                    normal = rtm::normalize((rtm::mk(n0.x, n0.y, n0.z) * w + rtm::mk(n1.x, n1.y, n1.z) * h.u)
                                            + rtm::mk(n2.x, n2.y, n2.z) * h.v);
                    mat = S.chunk_mat + (size_t)__float_as_uint(n0.w) * 4;
                }
            } else {
                const float4 s = S.sph_geom[h.id];
                normal = rtm::normalize(hitPoint - rtm::mk(s.x, s.y, s.z));
                mat = S.sph_mat + (size_t)h.id * 4;
            }
            const float4 mcol = mat[0], memi = mat[1], mspec = mat[2], mprm = mat[3];
            const int flag = (int)__float_as_uint(mprm.w);
            v3 colour = rtm::mk(mcol.x, mcol.y, mcol.z);
            bool skip = false;
            if (flag == 1) {                                                           // CheckerPattern :313-317
                float cx = mod2(__builtin_floorf(hitPoint.x)), cz = mod2(__builtin_floorf(hitPoint.z));
                if (!(cx == cz)) colour = rtm::mk(memi.x, memi.y, memi.z);
            } else if (flag == 2 && bounce == 0) {                                     // InvisibleLightSource :318-322
                o = hitPoint + d * 0.001f;
                skip = true;
            }
            path_done = false;
            if (!skip) {
                const bool isSpecular = mprm.z >= rtm::random_value(rng);              // :325
                const float specF = isSpecular ? 1.0f : 0.0f;
                o = hitPoint;                                                          // :327
                v3 diffuseDir = rtm::normalize(normal + rtm::random_direction(rng));
                v3 specularDir = rtm::reflect(d, normal);
                d = rtm::normalize(rtm::lerp(diffuseDir, specularDir, mprm.y * specF));

                v3 emitted = rtm::mk(memi.x, memi.y, memi.z) * mprm.x;                 // :333-335
                light = light + emitted * rayColour;
                rayColour = rayColour * rtm::lerp(colour, rtm::mk(mspec.x, mspec.y, mspec.z), specF);

                float pr = rtm::fmax_(rayColour.x, rtm::fmax_(rayColour.y, rayColour.z));   // :338-342
                if (rtm::random_value(rng) >= pr) path_done = true;
                else { float ip = rtm::rcp_(pr); rayColour = rayColour * ip; }
            }
            ++bounce;
            if (bounce > p.maxBounceCount) path_done = true;                           // loop bound :305
        } else {
            phase_tick<COUNT>(cnt, 3);
            light = light + environment_light(p, d) * rayColour;                       // :346-347
            path_done = true;
        }

        if (path_done) {
            total = total + light;                                                     // :384
            ++sample;
            if (sample >= p.numRaysPerPixel) alive = false;
            else {
                phase_tick<COUNT>(cnt, 4);
                camera_ray(p, cam, rng, o, d, fixed_origin);
                bounce = 0;
                rayColour = rtm::mk(1.f, 1.f, 1.f); light = rtm::mk(0.f, 0.f, 0.f);
            }
        }
    }
    const float n = (float)p.numRaysPerPixel;
    return rtm::mk(total.x / n, total.y / n, total.z / n);                             // :387
}

constexpr int kBlock = 256;         // 4 waves
constexpr int kGroupMax = 32;       // k_stream: most work items in a wave's group (the LDS table of their decoded positions: 256 B per wave)
constexpr int kWavesPerBlock = kBlock / 64;

// WAVES = waves per SIMD the register allocation aims at.  With a BVH the LDS stacks allow four workgroups per CU and the 95 VGPRs
// the kernel takes by itself (five waves) are left alone (0); a scene of spheres only has no stack to speak of and is shading-bound:
// six waves per SIMD (80 VGPRs) measure 37.3 Grays/s on the sphere workload against 34.9 at five and 33.7 at eight.
template <bool COUNT, bool FLAT, bool H = false, int WAVES = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WAVES == 0 ? 1 : WAVES, WAVES == 0 ? 8 : WAVES))) void k_trace(DeviceScene S, FrameArgs F)
{
    extern __shared__ uint32_t lds_stack[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TravStack stk;
    stk.lds = lds_stack + (size_t)wave * F.stack_cap * 64 + lane;
    stk.cap = F.stack_cap; stk.stride = F.gstack_stride;
    stk.glb = F.gstack ? F.gstack + (blockIdx.x * kBlock + threadIdx.x) : nullptr;
    Counters cnt = {};
    const int ntiles = F.tiles_x * F.tiles_y;
    const bool batched = F.frames_in_launch > 1;
    const unsigned int nitems = (unsigned)ntiles * (unsigned)(batched ? F.frames_in_launch : 1);
    const float weight = 1.0f / (float)(F.frame + 1);                                  // Accumulate.shader:48
    const float omw = 1.0f - weight;
    const rt_params& P = F.p;
    const bool full_sort = F.full_sort != 0, fixed_origin = F.fixed_origin != 0;

    for (;;) {
        // (what a tile needs before and after its pixels is read fresh — fresh_kernargs — so that it does not sit in SGPRs during
        // the pixel loop; the pixel loop itself keeps the kernel's plain parameters, see there)
        const TraceKernArgs& KA = fresh_kernargs<TraceKernArgs>();
        const FrameArgs& F = KA.F;
        // work item = (frame, tile), frame-major: with several frames per launch the wave count no longer has to be
        // matched by the tile count for the persistent waves to balance (frames are independent: frag :362 seeds by Frame)
        unsigned int item = 0;
        if (lane == 0) item = atomicAdd(F.tile_counter, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nitems) break;
        const unsigned int fi = item / (unsigned)ntiles;
        unsigned int tile = item - fi * (unsigned)ntiles;
        if (F.tile_order) tile = F.tile_order[tile];
        const unsigned long long t_begin = F.tile_cost ? __builtin_readcyclecounter() : 0ull;
        const int tx = tile % F.tiles_x, ty = tile / F.tiles_x;
        const int tw = F.tile_w_log2, th = 6 - tw;
        const int x = (tx << tw) + (lane & ((1 << tw) - 1)), ly = (ty << th) + (lane >> tw);
        if (x < F.p.width && ly < F.nrows) {
            const int y = F.row0 + (ly >> 3) * F.row_stride + (ly & 7);
            v3 c = render_pixel<COUNT, FLAT, H>(S, P, full_sort, fixed_origin, F.frame + (int)fi, x, y, stk, cnt);
            const TraceKernArgs& KB = fresh_kernargs<TraceKernArgs>();
            const FrameArgs& F = KB.F;
            const size_t pi = (size_t)ly * F.p.width + x;
            F.out_frame[(size_t)fi * F.frame_stride + pi] = make_float4(c.x, c.y, c.z, 1.0f);     // frag :388
            if (!batched) {
                float4 prev = F.accum[pi];                                             // Accumulate.shader:45-50
                float4 acc;
                acc.x = rtm::saturate(prev.x * omw + c.x * weight);
                acc.y = rtm::saturate(prev.y * omw + c.y * weight);
                acc.z = rtm::saturate(prev.z * omw + c.z * weight);
                acc.w = rtm::saturate(prev.w * omw + 1.0f * weight);
                F.accum[pi] = acc;
            }
        }
        if (F.tile_cost && lane == 0)
            atomicAdd(&F.tile_cost[tile], (uint32_t)((__builtin_readcyclecounter() - t_begin) >> 6));
    }
    {
        unsigned long long v[kNumCounters] = { cnt.rays, cnt.sph, cnt.nodes, cnt.tris, cnt.hits };
        for (int k = 0; k < 5; ++k) { v[5 + k] = cnt.phase_lanes[k]; v[10 + k] = cnt.phase_execs[k]; }
        for (int k = 0; k < kNumRegions; ++k) v[15 + k] = cnt.region[k];
        for (int k = 0; k < (COUNT ? kNumCounters : 1); ++k) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) atomicAdd(&F.counters[k], s);
        }
    }
}

// Accumulate.shader:43-54 for a batch of frames traced by one launch: per pixel, frames first .. first+n-1 in order.
// (a template only so that this header can be included by more than one translation unit)
template <int = 0>
__global__ __launch_bounds__(256) void k_accumulate(const float4* __restrict__ frames, float4* __restrict__ accum,
                                                    float4* __restrict__ last_frame, size_t pixels, unsigned int frame_stride,
                                                    int first_frame, int n_frames)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pixels; i += (size_t)gridDim.x * blockDim.x) {
        float4 acc = accum[i], cur = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int f = 0; f < n_frames; ++f) {
            cur = frames[(size_t)f * frame_stride + i];
            const float weight = 1.0f / (float)(first_frame + f + 1);
            const float omw = 1.0f - weight;
            acc.x = rtm::saturate(acc.x * omw + cur.x * weight);
            acc.y = rtm::saturate(acc.y * omw + cur.y * weight);
            acc.z = rtm::saturate(acc.z * omw + cur.z * weight);
            acc.w = rtm::saturate(acc.w * omw + cur.w * weight);
        }
        accum[i] = acc;
        last_frame[i] = cur;
    }
}

} // namespace rtk
