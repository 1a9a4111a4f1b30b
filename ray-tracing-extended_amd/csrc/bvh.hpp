// bvh.hpp — host-side build of the 4-wide BVH that replaces the reference's flat chunk list.
//
// The reference has no hierarchy: every ray tests every chunk AABB and brute-forces the triangles of the
// passing chunks (Assets/Scripts/Shaders/RayTracing.shader:276-294; chunks come from
// Assets/Scripts/Helpers/MeshSplitter.cs:65-99).  A BVH only prunes: the closest hit is still decided by the
// reference's own triangle arithmetic and the (dst, buffer index) order, so any conservative hierarchy
// returns the same bits.  Boxes are padded (see pad_box) so that float rounding in the slab test can never
// reject a triangle the reference arithmetic accepts.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rtbvh {

// 128-byte node = one L2 line: four child boxes in SoA form (float4 loads) + four child references.
struct alignas(16) Node4 {
    float    minx[4], miny[4], minz[4];
    float    maxx[4], maxy[4], maxz[4];
    uint32_t child[4];      // < 0x80000000: node index; >= 0x80000000: leaf; 0xFFFFFFFF: empty slot
    uint32_t meta[4];       // meta[0] = number of used slots
};
static_assert(sizeof(Node4) == 128, "Node4 must be one 128-byte line");

// Compact form of the same node for the traversal kernels (Node4h, also one 128-byte line), derived from a Node4 on the
// device (rt_geom.hpp k_compact_nodes).  Measured on MI355X (tools/ubench/vmem_rate.hip): an L1-resident wave load costs the
// CU's vector-memory path ~16 cycles per *instruction* — dwordx2 and dwordx4 alike, 64 active lanes or 16 — and a node
// visit of the f32 form needs seven of them; adding two more per visit costs the tracer 12 % (profiles/ab_probe_r02.txt).
// Node4h needs five:
//   bytes   0.. 63  four (x, y) plane sets of 16 B = 4 x-planes + 4 y-planes as f16 offsets from the node's origin:
//                   set c = (dir.x < 0) + 2 (dir.y < 0) holds the NEAR planes of a ray with those signs
//                   (x: min or max per bit 0, y: per bit 1); the FAR planes of that ray are set 3 - c
//   bytes  64.. 95  (minz[4], maxz[4]) then (maxz[4], minz[4]) as f16: near z then far z for dir.z >= 0 / < 0
//   bytes  96..111  child[4]
//   bytes 112..127  origin.xyz (f32), unused
// plane = origin + offset; every min offset is rounded towards -inf and every max offset towards +inf from the padded f32
// box, so the compact box contains the f32 one (the hierarchy only prunes; the closest hit is unchanged).  Offsets too
// large for f16 become +-inf, which keeps the box conservative.  No f16 denormal is ever stored.
struct alignas(16) Node4h { uint32_t w[32]; };
static_assert(sizeof(Node4h) == 128, "Node4h must be one 128-byte line");

#if defined(__HIP__)
#define RTBVH_HD __host__ __device__
#else
#define RTBVH_HD
#endif
// largest f16 (as bits; no denormals besides zero) that is <= x.  +-inf map to themselves, NaN to +inf / -inf (the widest box).
RTBVH_HD inline uint16_t f16_round_down(float x)
{
    uint32_t u; __builtin_memcpy(&u, &x, 4);
    const uint32_t sign = u >> 31, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return a > 0x7F800000u ? (uint16_t)0xFC00u : (uint16_t)((sign << 15) | 0x7C00u);   // NaN -> -inf (a lower bound of anything)
    const bool away = sign != 0;                                   // negative values round away from zero
    uint32_t h;
    if (a >= 0x47800000u) h = away ? 0x7C00u : 0x7BFFu;            // |x| >= 65536
    else if (a < 0x38800000u) h = (away && a != 0) ? 0x0400u : 0u; // |x| < 2^-14: zero, or the smallest normal
    else {
        h = (((a >> 23) - 112u) << 10) | ((a >> 13) & 0x3FFu);
        if (away && (a & 0x1FFFu)) ++h;                            // may carry up to 0x7C00 = inf: |x| > 65504
    }
    return (uint16_t)((sign << 15) | h);
}
RTBVH_HD inline uint16_t f16_round_up(float x) { return (uint16_t)(f16_round_down(-x) ^ 0x8000u); }
// the next float32 below / above x (x finite): covers the rounding of the subtraction plane - origin
RTBVH_HD inline float f32_below(float x)
{
    uint32_t u; __builtin_memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return x;
    if ((u & 0x7FFFFFFFu) == 0u) u = 0x80000001u; else u = (u >> 31) ? u + 1u : u - 1u;
    __builtin_memcpy(&x, &u, 4); return x;
}
RTBVH_HD inline float f32_above(float x) { return -f32_below(-x); }

constexpr uint32_t kEmpty   = 0xFFFFFFFFu;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr int      kMaxLeaf = 4;
// leaf reference = kLeafBit | (firstTriangleInBvhOrder << 2) | (count - 1)
inline uint32_t make_leaf(uint32_t first, uint32_t count) { return kLeafBit | (first << 2) | (count - 1u); }

struct Bvh {
    std::vector<Node4>    nodes;       // nodes[0] is the root (empty when there are no triangles)
    std::vector<uint32_t> order;       // BVH order -> index into the uploaded triangle buffer
    int                   maxStack = 0; // worst-case number of entries the traversal stack can hold
    int                   depth = 0;
    float                 magnitude = 0.f;  // G used for the box padding (scene coordinates and ray origins)
    std::vector<uint32_t> levelStart;       // nodes of tree level L are [levelStart[L], levelStart[L+1]) (breadth-first order)
};

// Builder tuning, per call (two contexts on two host threads build independently): SAH bins per axis (2..128), the exponent,
// in percent, of the triangle count in the SAH's subtree-cost model area * count^e, passes of insertion-based optimisation of
// the binary tree, triangles per leaf (1..kMaxLeaf).  The hierarchy changes, the closest hit it returns does not.
struct Tuning { int bins = 32; int cost_exp_percent = 100; int reinsert_passes = 0; int max_leaf = 2;
                // collapse of the binary tree to 4-wide nodes: 0 = greedy (open the child of largest area; default — measured best on the
                // headline scene, profiles/bvh_collapse_r04.txt), 1 = cost-driven (dynamic programme over "at most i children" per binary
                // node, leaves formed by it), 2 = cost-driven over the split search's own leaves; node_cost_percent = cost of a node step in percent of a
                // triangle test's (measured on MI355X: 345 against 265 SIMD cycles per wave-level execution)
                int collapse_dp = 0; int node_cost_percent = 130; };

// tri_pos: 9 floats per triangle (posA, posB, posC) every stride_floats.  origin_magnitude = largest |coordinate| a ray origin
// outside the triangles can have (the camera, sphere surfaces): it widens the absolute part of the box padding.
void build(const float* tri_pos, size_t stride_floats, uint32_t n_tris, float origin_magnitude, const Tuning& tuning, Bvh& out);

// The same split search over n boxes (6 floats each: lo.xyz, hi.xyz), one leaf per box: the top of the device builder's tree over the
// clusters its bottom-up rounds have formed.  out = pairs (left, right) of the internal nodes, out[0..1] the root; a child >= 0 is a box
// index, a child < 0 is ~(internal node index).
void build_over_boxes(const float* boxes, uint32_t n, const Tuning& tuning, std::vector<int32_t>& out);

} // namespace rtbvh
