// Micro-benchmark: the achievable issue rate of k_stream's traversal instruction mix on gfx950.
//
// k_stream's two hot loops are the BVH4 node step (node_step<true>: 5 x global_load_dwordx4 of one f16 node, 3 + 24 FMAs for the slab
// planes, 16 min/max, compares and selects for the nearest-first exchange, three branch-free LDS pushes and the pop) and the triangle
// test (load_tri + ray_triangle: 3 x global_load_dwordx4, ~60 VALU).  This runs exactly that code (the product's own headers, the
// product's flags) in lockstep on data that stays in the L1 — every lane active, no divergence, no cache miss — at 1..6 waves per SIMD,
// and reports cycles per step and per VALU instruction (the VALU count of the loop body comes from the ISA: tools/ubench/README).
// What it measures is the ceiling of the mix: the SPEC rate is one VALU wave-instruction per 2 cycles per SIMD, the mix with its loads
// and LDS traffic reaches less, and k_stream's issue fraction should be read against this number as well (bench.py roofline.issue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -I ray-tracing-extended_amd/csrc tools/ubench/node_mix.hip -o tools/ubench/node_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "rt_kernels.hpp"

using namespace rtk;

constexpr int kNodes = 64;          // 8 KB of nodes: L1-resident

template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k_node(const float4* __restrict__ nodes, float* out, int iters, float seed)
{
    extern __shared__ uint32_t lds[];
    using lds_u32 = __attribute__((address_space(3))) uint32_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t stk0 = (uint32_t)(uintptr_t)(lds_u32*)(lds + (size_t)wave * 8 * 64 + lane);
    auto slot = [](uint32_t a) -> lds_u32& { return *reinterpret_cast<lds_u32*>((uintptr_t)a); };
    v3 o = rtm::mk(0.1f * lane * seed, 0.2f * seed, -3.0f * seed), d = rtm::mk(0.01f * (lane & 7) * seed, 0.02f * (lane >> 3) * seed, seed);   // (seed = 1: run-time values, nothing folds)
    RaySlabT<true> slab = make_slab<true>(o, d);
    uint32_t cur = (uint32_t)lane % kNodes, top = stk0;
    float best = 1e30f, acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        float t0, t1, t2, t3; uint32_t c0, c1, c2, c3;
        node_step<true>(nodes, cur, slab, best, false, t0, t1, t2, t3, c0, c1, c2, c3);
        const float INF = __builtin_inff();
        slot(top) = c3; top = (t3 < INF) ? top + 256u : top;
        slot(top) = c2; top = (t2 < INF) ? top + 256u : top;
        slot(top) = c1; top = (t1 < INF) ? top + 256u : top;
        if (t0 < INF) cur = c0 & (kNodes - 1);
        else { top = (top != stk0) ? top - 256u : top; cur = slot(top) & (kNodes - 1); }
        if (top - stk0 > 4u * 256u) top = stk0;                     // keep the synthetic stack inside its 8 entries
        acc += t0 < INF ? t0 : 0.f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)cur;
}

// Round 4, verdict item 1(a): the same step with FOUR loads — the child references are not loaded (bytes 96..111) but computed: the
// header at bytes 112..127 holds (origin.x, origin.y, origin.z, D) where D is the reference of child 0 and the low mantissa byte of each
// origin component the offset of children 1 / 2 / 3 from it (children of a node contiguous in one address space; the origin is any point
// near the node, so the builder is free to choose those eight bits): 1 + 3 x 2 full-rate VALU instead of a fifth dwordx4 load.
__device__ __forceinline__ void node_step4(const float4* __restrict__ nodes, uint32_t cur, const RaySlabT<true>& r, float best_t,
                                           float& t0, float& t1, float& t2, float& t3, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3)
{
    const char* nb = reinterpret_cast<const char*>(nodes);
    const uint32_t base = cur << 7;
    const float INF = __builtin_inff();
    const uint4 na = *reinterpret_cast<const uint4*>(nb + (base + r.nxy)), fa = *reinterpret_cast<const uint4*>(nb + (base + r.fxy));
    const uint4 zz = *reinterpret_cast<const uint4*>(nb + (base + r.zo));
    const uint4 hd = *reinterpret_cast<const uint4*>(nb + (base + 112u));
    const uint32_t d0 = hd.w & ~3u;
    c0 = hd.w; c1 = d0 + (hd.x & 0xFFu); c2 = d0 + (hd.y & 0xFFu); c3 = d0 + (hd.z & 0xFFu);
    const float ogx = __uint_as_float(hd.x), ogy = __uint_as_float(hd.y), ogz = __uint_as_float(hd.z);
    const float kx = __builtin_fmaf(ogx, r.inv.x, -r.oinv.x), ky = __builtin_fmaf(ogy, r.inv.y, -r.oinv.y), kz = __builtin_fmaf(ogz, r.inv.z, -r.oinv.z);
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
#define RT_CSWAP(TA, CA, TB, CB) { const bool s_ = TB < TA; const float tt_ = s_ ? TB : TA, tu_ = s_ ? TA : TB;         \
                                   const uint32_t ct_ = s_ ? CB : CA, cu_ = s_ ? CA : CB; TA = tt_; TB = tu_; CA = ct_; CB = cu_; }
    RT_CSWAP(t0, c0, t1, c1) RT_CSWAP(t2, c2, t3, c3) RT_CSWAP(t0, c0, t2, c2)
#undef RT_CSWAP
}

template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k_node4(const float4* __restrict__ nodes, float* out, int iters, float seed)
{
    extern __shared__ uint32_t lds[];
    using lds_u32 = __attribute__((address_space(3))) uint32_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t stk0 = (uint32_t)(uintptr_t)(lds_u32*)(lds + (size_t)wave * 8 * 64 + lane);
    auto slot = [](uint32_t a) -> lds_u32& { return *reinterpret_cast<lds_u32*>((uintptr_t)a); };
    v3 o = rtm::mk(0.1f * lane * seed, 0.2f * seed, -3.0f * seed), d = rtm::mk(0.01f * (lane & 7) * seed, 0.02f * (lane >> 3) * seed, seed);
    RaySlabT<true> slab = make_slab<true>(o, d);
    uint32_t cur = (uint32_t)lane % kNodes, top = stk0;
    float best = 1e30f, acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        float t0, t1, t2, t3; uint32_t c0, c1, c2, c3;
        node_step4(nodes, cur, slab, best, t0, t1, t2, t3, c0, c1, c2, c3);
        const float INF = __builtin_inff();
        slot(top) = c3; top = (t3 < INF) ? top + 256u : top;
        slot(top) = c2; top = (t2 < INF) ? top + 256u : top;
        slot(top) = c1; top = (t1 < INF) ? top + 256u : top;
        if (t0 < INF) cur = c0 & (kNodes - 1);
        else { top = (top != stk0) ? top - 256u : top; cur = slot(top) & (kNodes - 1); }
        if (top - stk0 > 4u * 256u) top = stk0;
        acc += t0 < INF ? t0 : 0.f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)cur;
}

template <int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k_tri(const float4* __restrict__ tris, float* out, int iters, float seed)
{
    const int lane = threadIdx.x & 63;
    v3 o = rtm::mk(0.1f * lane * seed, 0.2f * seed, -3.0f * seed), d = rtm::mk(0.01f * (lane & 7) * seed, 0.02f * (lane >> 3) * seed, seed);   // (seed = 1: run-time values, nothing folds)
    uint32_t ti = (uint32_t)lane % kNodes;
    float best = 1e30f, acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        float4 g0, g1, g2;
        load_tri(tris, ti, g0, g1, g2);
        float dst, u, v;
        const bool hit = ray_triangle(o, d, rtm::mk(g0.x, g0.y, g0.z), rtm::mk(g0.w, g1.x, g1.y), rtm::mk(g1.z, g1.w, g2.x), rtm::mk(g2.y, g2.z, g2.w), dst, u, v);
        if (hit && dst < best) { best = dst; acc += u + v; }
        ti = (ti + 1u + (hit ? 1u : 0u)) & (kNodes - 1);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + best;
}

template <class K> double run(K kern, const float4* data, float* out, int blocks, int iters, size_t lds)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, data, out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, data, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv)
{
    const int valu_node = argc > 1 ? atoi(argv[1]) : 78, valu_tri = argc > 2 ? atoi(argv[2]) : 60;   // VALU instructions per loop body (from the ISA)
    const int valu_node4 = argc > 3 ? atoi(argv[3]) : valu_node + 7;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    // nodes: four children each, every box hit by every ray (offsets +-60 around origin 0 in f16), children = the next nodes
    std::vector<uint32_t> h(kNodes * 32);
    auto f16 = [](float x) -> uint32_t { _Float16 v = (_Float16)x; uint16_t b; __builtin_memcpy(&b, &v, 2); return b; };
    for (int n = 0; n < kNodes; ++n) {
        uint32_t* w = &h[n * 32];
        const uint32_t lo = f16(-60.f), hi = f16(60.f);
        for (int c = 0; c < 4; ++c) {
            const uint32_t x = (c & 1) ? hi : lo, y = (c & 2) ? hi : lo;
            w[4 * c + 0] = x | (x << 16); w[4 * c + 1] = x | (x << 16); w[4 * c + 2] = y | (y << 16); w[4 * c + 3] = y | (y << 16);
        }
        w[16] = w[17] = lo | (lo << 16); w[18] = w[19] = hi | (hi << 16); w[20] = w[21] = hi | (hi << 16); w[22] = w[23] = lo | (lo << 16);
        for (int k = 0; k < 4; ++k) w[24 + k] = (uint32_t)((n * 4 + k + 1) % kNodes);
        // header for the four-load variant: origin (0, 0, 0) with the child offsets 1, 2, 3 in the low mantissa bytes (denormal-sized
        // origins: exact enough here), D = the first child
        w[28] = 1u; w[29] = 2u; w[30] = 3u; w[31] = (uint32_t)((n * 4 + 1) % kNodes);
    }
    std::vector<float> t(kNodes * 12);
    for (int n = 0; n < kNodes; ++n) { float* p = &t[n * 12]; const float z = 1.0f + n; float v[12] = { -50, -50, z, 100, 0, 0, 0, 100, 0, 0, 0, 10000 }; for (int k = 0; k < 12; ++k) p[k] = v[k]; }
    float4 *dn, *dt; float* out;
    hipMalloc(&dn, h.size() * 4); hipMalloc(&dt, t.size() * 4); hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    hipMemcpy(dn, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    const size_t lds = 4 * 8 * 64 * 4;
    printf("cycles at 2.4 GHz nominal; one 256-thread workgroup per CU per wave/SIMD; %d VALU per node step, %d per triangle test (ISA)\n", valu_node, valu_tri);
    printf("%-10s %6s %14s %18s %22s\n", "loop", "w/SIMD", "cycles/step", "cycles/VALU instr", "VALU wave-instr/cycle/SIMD");
#define ROW(NAME, KERN, DATA, LDS, VALU, W)                                                                                  \
    { const double ms = run(KERN<W>, DATA, out, cus * W, iters, LDS);                                                           \
      const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * W);      /* per step per SIMD: W waves share a SIMD */             \
      printf("%-10s %6d %14.1f %18.2f %22.3f\n", NAME, W, cyc, cyc / VALU, VALU / cyc); }
    ROW("node", k_node, dn, lds, valu_node, 1) ROW("node", k_node, dn, lds, valu_node, 2) ROW("node", k_node, dn, lds, valu_node, 4) ROW("node", k_node, dn, lds, valu_node, 5) ROW("node", k_node, dn, lds, valu_node, 6)
    ROW("node4", k_node4, dn, lds, valu_node4, 1) ROW("node4", k_node4, dn, lds, valu_node4, 2) ROW("node4", k_node4, dn, lds, valu_node4, 4) ROW("node4", k_node4, dn, lds, valu_node4, 5) ROW("node4", k_node4, dn, lds, valu_node4, 6)
    ROW("triangle", k_tri, dt, 0, valu_tri, 1) ROW("triangle", k_tri, dt, 0, valu_tri, 2) ROW("triangle", k_tri, dt, 0, valu_tri, 4) ROW("triangle", k_tri, dt, 0, valu_tri, 5) ROW("triangle", k_tri, dt, 0, valu_tri, 6)
    return 0;
}
