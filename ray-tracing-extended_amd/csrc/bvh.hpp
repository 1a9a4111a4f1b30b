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

// tri_pos: 9 floats per triangle (posA, posB, posC).  origin_magnitude = largest |coordinate| a ray origin outside
// the geometry can have (the camera): it widens the absolute part of the box padding.
// max_leaf: triangles per leaf (1..kMaxLeaf).
// Builder tuning (process-wide): SAH bins per axis (2..128, default 32) and the exponent, in percent, of the triangle count in
// the SAH's subtree-cost model area * count^e (default 100); passes of insertion-based optimisation of the binary tree.  The hierarchy changes, the closest hit it returns does not.
void set_tuning(int bins, int cost_exp_percent, int reinsert_passes);
void build(const float* tri_pos, size_t stride_floats, uint32_t n_tris, float origin_magnitude, int max_leaf, Bvh& out);

} // namespace rtbvh
