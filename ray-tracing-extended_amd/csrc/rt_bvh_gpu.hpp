// rt_bvh_gpu.hpp — BVH build on the device: the step the reference does not have at all (it loops over a flat chunk list,
// RayTracing.shader:276-294, re-created on the CPU every frame: RayTracingManager.cs:135-164, TODO at RayTracedMesh.cs:37).
//
// World-space triangles (the reference's 72-byte layout, already on the device) -> BVH4 nodes + triangle order, without a host
// round trip of the geometry:
//   1. k_tri_bounds      scene bounds over triangle-box centroids + the largest |coordinate| (box padding), block reduce + atomics
//   2. k_morton          63-bit Morton code of each triangle's box centre, sorted with hipcub::DeviceRadixSort (the one library call)
//   3. PLOC              (Meister & Bittner, "Parallel Locally-Ordered Clustering for BVH Construction", TVCG 2018): clusters in
//                        Morton order; every round each cluster finds, among its 2R neighbours in the array, the one whose
//                        union box has the smallest area; mutual pairs merge into a new binary node; the array is compacted
//                        in order (block-local scan, scan of the block totals, gather).  Rounds run until one cluster is left;
//                        the last <= 512 clusters finish inside a single workgroup without further launches.
//   3c. treelets         (round 4) every maximal subtree of the clustering with <= 64 triangles is rebuilt by one wave with an exact
//                        sweep SAH (all three axes, every split position) over its triangles, in the node slots it already has:
//                        the clustering decides which triangles belong together, the split search how they are arranged
//   4. collapse          breadth-first, one launch per level: a binary node becomes a 4-wide node by opening its larger internal
//                        children; two sibling triangles form one leaf (the tracer's leaf size); children of one node are
//                        allocated contiguously, levels are contiguous ranges (what the refit kernels need); child boxes get
//                        the same padding as the host builder's (bvh.cpp pad_box)
//   5. k_stack_need      bottom-up, one launch per level: the exact worst-case traversal stack depth
// The hierarchy only prunes: whichever tree is built, the closest hit is decided by the reference's triangle arithmetic and the
// visiting-rank tie-break, so images do not change (tested bitwise against the host builder's tree and the oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "bvh.hpp"

namespace rtgb {

constexpr int kMaxRadius = 64;          // PLOC search radius (neighbours looked at on each side): run-time, up to this
constexpr int kPlocBlock = 512;         // clusters per workgroup and round
constexpr int kCtrNodes = 0, kCtrNode4 = 1, kCtrOrder = 2, kCtrFrontier = 3, kCtrClusters = 4, kCtrBounds = 8, kCtrTreelets = 16;   // (kCtrTreelets + pass: up to 16 passes)   // counter slots (uint32)

// monotone float <-> uint map for atomicMin / atomicMax on floats
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(uint32_t o) { const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; __builtin_memcpy(&f, &u, 4); return f; }

struct Box3 { float lo[3], hi[3]; };

__device__ __forceinline__ Box3 tri_box(const float* __restrict__ p)
{
    Box3 b;
    for (int a = 0; a < 3; ++a) {
        // fminf / fmaxf return the non-NaN operand: a NaN coordinate does not poison the box (as std::min / std::max in bvh.cpp)
        b.lo[a] = __builtin_fminf(__builtin_fminf(p[a], p[3 + a]), p[6 + a]);
        b.hi[a] = __builtin_fmaxf(__builtin_fmaxf(p[a], p[3 + a]), p[6 + a]);
    }
    return b;
}

__device__ __forceinline__ float union_half_area(float4 amin, float4 amax, float4 bmin, float4 bmax)
{
    const float dx = __builtin_fmaxf(amax.x, bmax.x) - __builtin_fminf(amin.x, bmin.x);
    const float dy = __builtin_fmaxf(amax.y, bmax.y) - __builtin_fminf(amin.y, bmin.y);
    const float dz = __builtin_fmaxf(amax.z, bmax.z) - __builtin_fminf(amin.z, bmin.z);
    const float a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : __builtin_inff();            // (inf - inf, 0 * inf: never the better candidate)
}
__device__ __forceinline__ float half_area(float4 mn, float4 mx)
{
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    const float a = dx * dy + dy * dz + dz * dx;
    return (dx >= 0.f && a == a) ? a : 0.f;
}

// ---- 1. scene bounds of the box centres, and the largest finite |coordinate| ------------------------------------------------
__global__ __launch_bounds__(256) void k_tri_bounds(const float* __restrict__ tris, uint32_t nt, uint32_t* __restrict__ ctr)
{
    float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -lo[0], -lo[0], -lo[0] }, mag = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += gridDim.x * blockDim.x) {
        const Box3 b = tri_box(tris + (size_t)i * 18);
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
            if (c > -3.0e38f && c < 3.0e38f) { lo[a] = __builtin_fminf(lo[a], c); hi[a] = __builtin_fmaxf(hi[a], c); }
            const float m = __builtin_fmaxf(__builtin_fabsf(b.lo[a]), __builtin_fabsf(b.hi[a]));
            if (m < 3.0e38f) mag = __builtin_fmaxf(mag, m);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int a = 0; a < 3; ++a) { lo[a] = __builtin_fminf(lo[a], __shfl_down(lo[a], off, 64)); hi[a] = __builtin_fmaxf(hi[a], __shfl_down(hi[a], off, 64)); }
        mag = __builtin_fmaxf(mag, __shfl_down(mag, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) { atomicMin(&ctr[kCtrBounds + a], f2ord(lo[a])); atomicMax(&ctr[kCtrBounds + 3 + a], f2ord(hi[a])); }
        atomicMax(&ctr[kCtrBounds + 6], f2ord(mag));
    }
}

// ---- 2. Morton codes ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread21(uint32_t v)
{
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ __launch_bounds__(256) void k_morton(const float* __restrict__ tris, uint32_t nt, const uint32_t* __restrict__ ctr,
                                                uint64_t* __restrict__ keys, uint32_t* __restrict__ ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const Box3 b = tri_box(tris + (size_t)i * 18);
    // one scale for the three axes (the scene's largest extent): cubic cells.  Normalising each axis by its own extent makes the
    // cells of a flat scene (a 150 x 2 x 90 floor of chess sets) as flat as the scene, and Morton neighbours would then be the
    // upper halves of pieces tens of units apart rather than the two halves of one piece (measured: 1.9x the tree area).
    float ext = 0.f;
    for (int a = 0; a < 3; ++a) ext = __builtin_fmaxf(ext, ord2f(ctr[kCtrBounds + 3 + a]) - ord2f(ctr[kCtrBounds + a]));
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(ctr[kCtrBounds + a]);
        const float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
        float t = (ext > 0.f) ? (c - lo) / ext : 0.f;
        t = (t == t) ? __builtin_fminf(__builtin_fmaxf(t, 0.f), 1.f) : 0.f;
        q[a] = (uint32_t)(t * 2097151.0f);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    ids[i] = i;
}

// leaves of the binary tree = triangles in Morton order; the first cluster array is 0 .. nt-1
__global__ __launch_bounds__(256) void k_leaf_boxes(const float* __restrict__ tris, const uint32_t* __restrict__ ids, uint32_t nt,
                                                    float4* __restrict__ bmin, float4* __restrict__ bmax, uint32_t* __restrict__ clusters)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const Box3 b = tri_box(tris + (size_t)ids[i] * 18);
    bmin[i] = make_float4(b.lo[0], b.lo[1], b.lo[2], __uint_as_float(1u));        // .w = triangles below the node (as an integer)
    bmax[i] = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.f);
    clusters[i] = i;
}

// ---- 3. PLOC --------------------------------------------------------------------------------------------------------------------
// nearest neighbour of local element e inside [lo, hi) of the staged window: smallest union area, ties to the smaller index
// (with that rule the globally cheapest pair is always mutual, so every round merges at least one pair)
__device__ __forceinline__ int ploc_nearest(const float4* smin, const float4* smax, int e, int lo, int hi, int kRadius)
{
    float best = __builtin_inff(); int bj = -1;
    const float4 amin = smin[e], amax = smax[e];
    for (int j = max(lo, e - kRadius); j < min(hi, e + kRadius + 1); ++j) {
        if (j == e) continue;
        const float c = union_half_area(amin, amax, smin[j], smax[j]);
        if (bj < 0 || c < best) { best = c; bj = j; }
    }
    return bj;
}

// One round over clusters [0, m): block b owns [b * kPlocBlock, ...), stages its range plus 2R on each side, finds the nearest
// neighbours of its range plus R on each side (so both sides of a block boundary see the same decisions), merges the mutual
// pairs it owns the lower element of, and writes its surviving clusters, compacted in order, to tmp[b * kPlocBlock ...].
__global__ __launch_bounds__(kPlocBlock) void k_ploc_round(const uint32_t* __restrict__ cin, uint32_t m, uint32_t nt,
                                                           float4* __restrict__ bmin, float4* __restrict__ bmax, int2* __restrict__ child,
                                                           uint32_t* __restrict__ ctr, uint32_t* __restrict__ tmp, uint32_t* __restrict__ block_count,
                                                           int kRadius)
{
    constexpr int WMAX = kPlocBlock + 4 * kMaxRadius;
    const int W = kPlocBlock + 4 * kRadius;
    __shared__ float4 smin[WMAX], smax[WMAX];
    __shared__ uint32_t sid[WMAX];
    __shared__ int snn[WMAX];
    __shared__ uint32_t scan[kPlocBlock];
    const int start = (int)(blockIdx.x * kPlocBlock);
    const int w0 = start - 2 * kRadius;                     // global index of window slot 0
    for (int s = threadIdx.x; s < W; s += kPlocBlock) {
        const int g = w0 + s;
        if (g >= 0 && g < (int)m) { const uint32_t id = cin[g]; sid[s] = id; smin[s] = bmin[id]; smax[s] = bmax[id]; }
    }
    __syncthreads();
    const int lo = max(0, -w0), hi = min(W, (int)m - w0);   // valid window slots
    for (int s = threadIdx.x; s < W; s += kPlocBlock) {
        const int g = w0 + s;
        snn[s] = (g >= start - kRadius && g < start + kPlocBlock + kRadius && s >= lo && s < hi) ? ploc_nearest(smin, smax, s, lo, hi, kRadius) : -1;
    }
    __syncthreads();
    const int e = (int)threadIdx.x + 2 * kRadius, g = start + (int)threadIdx.x;
    uint32_t keep = 0, value = 0;
    if (g < (int)m) {
        const int j = snn[e];
        const bool mutual = j >= 0 && snn[j] == e;
        if (mutual && e < j) {
            const uint32_t node = nt + atomicAdd(&ctr[kCtrNodes], 1u);
            const float4 amin = smin[e], amax = smax[e], cmin = smin[j], cmax = smax[j];
            bmin[node] = make_float4(__builtin_fminf(amin.x, cmin.x), __builtin_fminf(amin.y, cmin.y), __builtin_fminf(amin.z, cmin.z),
                                     __uint_as_float(__float_as_uint(amin.w) + __float_as_uint(cmin.w)));
            bmax[node] = make_float4(__builtin_fmaxf(amax.x, cmax.x), __builtin_fmaxf(amax.y, cmax.y), __builtin_fmaxf(amax.z, cmax.z), 0.f);
            child[node] = make_int2((int)sid[e], (int)sid[j]);
            keep = 1; value = node;
        } else if (!mutual) { keep = 1; value = sid[e]; }
    }
    scan[threadIdx.x] = keep;
    __syncthreads();
    for (int off = 1; off < kPlocBlock; off <<= 1) {
        const uint32_t add = (int)threadIdx.x >= off ? scan[threadIdx.x - off] : 0u;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    if (keep) tmp[(size_t)start + scan[threadIdx.x] - 1u] = value;
    if (threadIdx.x == kPlocBlock - 1) block_count[blockIdx.x] = scan[threadIdx.x];
}

// exclusive scan of the block totals (one workgroup), total -> ctr[kCtrClusters]
__global__ __launch_bounds__(1024) void k_ploc_scan(const uint32_t* __restrict__ block_count, uint32_t nb, uint32_t* __restrict__ block_offset,
                                                    uint32_t* __restrict__ ctr)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023u) / 1024u;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t i = threadIdx.x * per + k; if (i < nb) sum += block_count[i]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t i = threadIdx.x * per + k; if (i < nb) { block_offset[i] = run; run += block_count[i]; } }
    if (threadIdx.x == 1023) ctr[kCtrClusters] = part[1023];
}

__global__ __launch_bounds__(kPlocBlock) void k_ploc_gather(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ block_count,
                                                            const uint32_t* __restrict__ block_offset, uint32_t* __restrict__ cout)
{
    if (threadIdx.x < block_count[blockIdx.x]) cout[block_offset[blockIdx.x] + threadIdx.x] = tmp[(size_t)blockIdx.x * kPlocBlock + threadIdx.x];
}

// the last m <= kPlocBlock clusters: all remaining rounds in one workgroup; the root's id ends up in ctr[kCtrClusters + 1]
__global__ __launch_bounds__(kPlocBlock) void k_ploc_tail(const uint32_t* __restrict__ cin, uint32_t m, uint32_t nt,
                                                          float4* __restrict__ bmin, float4* __restrict__ bmax, int2* __restrict__ child,
                                                          uint32_t* __restrict__ ctr, int kRadius)
{
    __shared__ float4 smin[kPlocBlock], smax[kPlocBlock];
    __shared__ uint32_t sid[kPlocBlock];
    __shared__ int snn[kPlocBlock];
    __shared__ uint32_t scan[kPlocBlock];
    const int t = (int)threadIdx.x;
    if (t < (int)m) { const uint32_t id = cin[t]; sid[t] = id; smin[t] = bmin[id]; smax[t] = bmax[id]; }
    __syncthreads();
    int cnt = (int)m;
    while (cnt > 1) {
        if (t < cnt) snn[t] = ploc_nearest(smin, smax, t, 0, cnt, kPlocBlock);     // the last clusters (the top of the tree) search all of them
        __syncthreads();
        uint32_t keep = 0, value = 0; float4 nmin = make_float4(0, 0, 0, 0), nmax = nmin;
        if (t < cnt) {
            const int j = snn[t];
            const bool mutual = j >= 0 && snn[j] == t;
            if (mutual && t < j) {
                const uint32_t node = nt + atomicAdd(&ctr[kCtrNodes], 1u);
                nmin = make_float4(__builtin_fminf(smin[t].x, smin[j].x), __builtin_fminf(smin[t].y, smin[j].y), __builtin_fminf(smin[t].z, smin[j].z),
                                   __uint_as_float(__float_as_uint(smin[t].w) + __float_as_uint(smin[j].w)));
                nmax = make_float4(__builtin_fmaxf(smax[t].x, smax[j].x), __builtin_fmaxf(smax[t].y, smax[j].y), __builtin_fmaxf(smax[t].z, smax[j].z), 0.f);
                bmin[node] = nmin; bmax[node] = nmax;
                child[node] = make_int2((int)sid[t], (int)sid[j]);
                keep = 1; value = node;
            } else if (!mutual) { keep = 1; value = sid[t]; nmin = smin[t]; nmax = smax[t]; }
        }
        scan[t] = keep;
        __syncthreads();
        for (int off = 1; off < kPlocBlock; off <<= 1) {
            const uint32_t add = t >= off ? scan[t - off] : 0u;
            __syncthreads();
            scan[t] += add;
            __syncthreads();
        }
        const int next = (int)scan[kPlocBlock - 1];
        __syncthreads();
        if (keep) { const int pos = (int)scan[t] - 1; sid[pos] = value; smin[pos] = nmin; smax[pos] = nmax; }
        __syncthreads();
        cnt = next;
    }
    if (t == 0) ctr[kCtrClusters + 1] = sid[0];
}

// ---- 3b. the top of the tree: the last `m` clusters go to the host as boxes, a binned-SAH split search builds the binary tree over
// them (bvh.cpp build_over_boxes: the upper levels decide most of a ray's node visits, and a top-down SAH places them better than
// bottom-up clustering does), and its nodes come back as (left, right) pairs.
__global__ __launch_bounds__(256) void k_cluster_boxes(const uint32_t* __restrict__ cin, uint32_t m, const float4* __restrict__ bmin,
                                                       const float4* __restrict__ bmax, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float4 lo = bmin[cin[i]], hi = bmax[cin[i]];
    float* o = out + 6 * (size_t)i;
    o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = hi.x; o[4] = hi.y; o[5] = hi.z;
}
// pairs: the host's internal nodes (child >= 0: index into cin, child < 0: ~(host node)); host node k becomes binary node base + k.
// One launch per depth is not needed: a node's box is the union of the boxes of the clusters below it, which the host has computed
// and sends along (boxes: 6 floats per host node).
__global__ __launch_bounds__(256) void k_top_nodes(const int32_t* __restrict__ pairs, const float* __restrict__ boxes, uint32_t n_top, uint32_t base,
                                                   const uint32_t* __restrict__ cin, float4* __restrict__ bmin, float4* __restrict__ bmax,
                                                   int2* __restrict__ child)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_top) return;
    const int32_t l = pairs[2 * k], r = pairs[2 * k + 1];
    child[base + k] = make_int2(l >= 0 ? (int)cin[l] : (int)(base + (uint32_t)~l), r >= 0 ? (int)cin[r] : (int)(base + (uint32_t)~r));
    const float* b = boxes + 6 * (size_t)k;
    bmin[base + k] = make_float4(b[0], b[1], b[2], 0.f);
    bmax[base + k] = make_float4(b[3], b[4], b[5], 0.f);
}

// ---- 3c. treelets: sweep SAH over the bottom of the tree ------------------------------------------------------------------------
// The clustering pairs nearest neighbours; which of a triangle's neighbours it ends up with depends on the order the rounds meet them,
// and the bottom levels come out less regular than a split search's (3.3x as many single-triangle leaves on the headline scene, 12 %
// more wave-level node steps).  So the bottom is rebuilt: a *treelet root* is a node of the clustering with at most kTreeletMax
// triangles below it whose parent has more (or that was handed to the top as a cluster); one wave takes one root, collects its
// triangles (one per lane) and the internal nodes of the subtree (their slots are reused), and builds the binary tree over them top
// down: at every level, for every segment in parallel, the lanes are sorted along each axis (bitonic network on a key of segment |
// centroid | lane), prefix and suffix boxes give the SAH cost area(L) * |L| + area(R) * |R| of EVERY split position (bvh.cpp's cost
// function; no bins), the cheapest of the three axes wins.  The split between lanes i and i + 1 is made exactly once, so internal node
// i of the treelet takes slot i of the collected ones (the root keeps its own: its parent's link stays valid).
constexpr int kTreeletMax = 64;

// roots: (1) the internal children with <= kTreeletMax triangles of a clustering node with more; (2) the candidates `extra`
// (the clusters handed to the top builder / the clustering's own root) when they are internal and small enough
__global__ __launch_bounds__(256) void k_treelet_roots(uint32_t nt, const float4* __restrict__ bmin, const int2* __restrict__ child,
                                                       const uint32_t* __restrict__ extra, uint32_t n_extra, uint32_t* __restrict__ ctr,
                                                       uint32_t* __restrict__ roots, uint32_t max_tris, uint32_t min_tris, int slot)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t made = ctr[kCtrNodes];
    if (i < made) {
        const uint32_t p = nt + i;
        if (__float_as_uint(bmin[p].w) > max_tris) {
            const int2 c = child[p];
            const uint32_t kids[2] = { (uint32_t)c.x, (uint32_t)c.y };
            for (int k = 0; k < 2; ++k)
                if (kids[k] >= nt && kids[k] < nt + made && __float_as_uint(bmin[kids[k]].w) <= max_tris && __float_as_uint(bmin[kids[k]].w) >= min_tris)
                    roots[atomicAdd(&ctr[kCtrTreelets + slot], 1u)] = kids[k];
        }
    }
    if (i < n_extra) {
        const uint32_t x = extra[i];
        if (x >= nt && x < nt + made && __float_as_uint(bmin[x].w) <= max_tris && __float_as_uint(bmin[x].w) >= min_tris)
            roots[atomicAdd(&ctr[kCtrTreelets + slot], 1u)] = x;
    }
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
// ascending bitonic sort of one key per lane over the wave
__device__ __forceinline__ uint64_t wave_sort(uint64_t key, int lane)
{
    for (int k = 2; k <= 64; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint64_t other = shfl_xor_u64(key, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            key = (lower == up) ? (other < key ? other : key) : (other > key ? other : key);
        }
    return key;
}
struct WBox { float lo[3], hi[3]; };
__device__ __forceinline__ WBox wbox_shfl(const WBox& b, int src)
{
    WBox r;
    for (int a = 0; a < 3; ++a) { r.lo[a] = __shfl(b.lo[a], src, 64); r.hi[a] = __shfl(b.hi[a], src, 64); }
    return r;
}
__device__ __forceinline__ void wbox_grow(WBox& b, const WBox& o)
{
    for (int a = 0; a < 3; ++a) { b.lo[a] = __builtin_fminf(b.lo[a], o.lo[a]); b.hi[a] = __builtin_fmaxf(b.hi[a], o.hi[a]); }
}
__device__ __forceinline__ float wbox_area(const WBox& b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
// inclusive scans of the boxes inside the segments [seg_lo, seg_hi): from the left and from the right
__device__ __forceinline__ void wbox_scan(const WBox& b, int lane, int seg_lo, int seg_hi, WBox& pre, WBox& suf)
{
    pre = b; suf = b;
    for (int off = 1; off < 64; off <<= 1) {
        WBox o;
        for (int a = 0; a < 3; ++a) { o.lo[a] = __shfl_up(pre.lo[a], off, 64); o.hi[a] = __shfl_up(pre.hi[a], off, 64); }
        if (lane - off >= seg_lo) wbox_grow(pre, o);
        for (int a = 0; a < 3; ++a) { o.lo[a] = __shfl_down(suf.lo[a], off, 64); o.hi[a] = __shfl_down(suf.hi[a], off, 64); }
        if (lane + off < seg_hi) wbox_grow(suf, o);
    }
}

__global__ __launch_bounds__(256) void k_treelet_sah(const uint32_t* __restrict__ roots, const uint32_t* __restrict__ ctr, uint32_t nt,
                                                     float4* __restrict__ bmin, float4* __restrict__ bmax, int2* __restrict__ child, int pair_cost,
                                                     uint32_t item_tris, int slot, int isolate)
{
    __shared__ uint32_t s_item[4][64], s_pool[4][64];
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    volatile uint32_t* items = s_item[wv]; volatile uint32_t* pool = s_pool[wv];
    const uint32_t n_roots = ctr[kCtrTreelets + slot];
    const uint64_t lt = (1ull << lane) - 1ull;
    const float INF = __builtin_inff();
    for (uint32_t r = blockIdx.x * 4u + (uint32_t)wv; r < n_roots; r += gridDim.x * 4u) {
        const uint32_t root = roots[r];
        // ---- the items below the root (one per lane, any order) — triangles, or in the later passes subtrees of at most item_tris
        // triangles — and the internal nodes above them (pool[0] = the root), by opening the frontier until everything is an item or
        // the wave is full (then the larger subtrees that are left count as items too)
        uint32_t item = lane == 0 ? root : 0u;
        uint32_t wt = lane == 0 ? __float_as_uint(bmin[root].w) : 0u;                 // triangles below the lane's item
        int cnt = 1, n_int = 0;
        for (;;) {
            const bool want = lane < cnt && item >= nt && wt > item_tris;
            const uint64_t wmask = __builtin_amdgcn_ballot_w64(want);
            const int room = kTreeletMax - cnt;
            if (wmask == 0ull || room <= 0) break;
            const bool open = want && (int)__popcll(wmask & lt) < room;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(open);
            const int rank = (int)__popcll(mask & lt), add = (int)__popcll(mask);
            if (open) { const int2 c = child[item]; pool[n_int + rank] = item; item = (uint32_t)c.x; items[cnt + rank] = (uint32_t)c.y; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (lane >= cnt && lane < cnt + add) item = items[lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (open || (lane >= cnt && lane < cnt + add)) wt = item < nt ? 1u : __float_as_uint(bmin[item].w);
            n_int += add; cnt += add;
        }
        const int n = cnt;
        if (n < 3) continue;
        WBox box;
        if (lane < n) { const float4 mn = bmin[item], mx = bmax[item]; box.lo[0] = mn.x; box.lo[1] = mn.y; box.lo[2] = mn.z; box.hi[0] = mx.x; box.hi[1] = mx.y; box.hi[2] = mx.z; }
        else for (int a = 0; a < 3; ++a) { box.lo[a] = INF; box.hi[a] = -INF; }
        int seg_lo = lane < n ? 0 : lane, seg_hi = lane < n ? n : lane + 1;    // lanes without a triangle: segments of their own, never split
        int par = -1, side = 0;                                                // the split (= slot) of the segment's parent; -1: the treelet's root
        bool first = true;
        for (;;) {
            const bool active = seg_hi - seg_lo >= 2;
            if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
            // ---- every split position of every segment, along each axis
            float best = INF; int best_axis = 0, best_pos = seg_lo + ((seg_hi - seg_lo) >> 1) - 1;      // (nothing finite: the middle, along x)
            int src_axis[3];                                    // the lane each position takes its item from, in the order along each axis
            WBox whole; uint32_t wsum = 0u;                     // the segment's own box and triangle count (what its node stores)
            float iso_cost = INF; int iso_pos = 0;
            for (int axis = 0; axis < 3; ++axis) {
                const float cen = 0.5f * box.lo[axis] + 0.5f * box.hi[axis];
                const uint64_t key = wave_sort(((uint64_t)(uint32_t)seg_lo << 40) | ((uint64_t)f2ord(cen) << 8) | (uint64_t)(uint32_t)lane, lane);
                const int src = (int)(key & 0xFFull);
                src_axis[axis] = src;
                const WBox sb = wbox_shfl(box, src);
                WBox pre, suf;
                wbox_scan(sb, lane, seg_lo, seg_hi, pre, suf);
                const WBox right = wbox_shfl(suf, min(lane + 1, 63));
                // triangles on the left of every position (the order along this axis): inclusive scan of the weights inside the segment
                const uint32_t w0 = (uint32_t)__shfl((int)wt, src, 64);
                uint32_t wl = w0;
                for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)wl, off, 64); if (lane - off >= seg_lo) wl += o; }
                const uint32_t wall = (uint32_t)__shfl((int)wl, seg_hi - 1, 64);
                if (axis == 0) {
                    whole = wbox_shfl(pre, seg_hi - 1); wsum = wall;
                    // large-box isolation (bvh.cpp's extra candidate): a sweep over centroids can never separate an item whose box spans
                    // the segment (a floor triangle among a thousand small ones) from the rest — it would sink to the bottom and inflate
                    // every box on its way.  Candidate: the item of largest area alone against everything else.
                    if (isolate) {
                        float ma = wbox_area(sb); ma = (lane < n && ma == ma) ? ma : -1.f;
                        int mp = lane;
                        for (int off = 1; off < 64; off <<= 1) {
                            const float oa = __shfl_up(ma, off, 64); const int op = __shfl_up(mp, off, 64);
                            if (lane - off >= seg_lo && oa > ma) { ma = oa; mp = op; }
                        }
                        ma = __shfl(ma, seg_hi - 1, 64); mp = __shfl(mp, seg_hi - 1, 64);
                        WBox rest = wbox_shfl(pre, max(mp - 1, 0));
                        const WBox after = wbox_shfl(suf, min(mp + 1, 63));
                        if (mp == seg_lo) rest = after; else if (mp + 1 < seg_hi) wbox_grow(rest, after);
                        const uint32_t wb = (uint32_t)__shfl((int)w0, mp, 64);
                        if (active && seg_hi - seg_lo > 2 && ma > 0.25f * wbox_area(whole)) {
                            const uint32_t nr = wall - wb;
                            const float c = pair_cost ? ma * (float)((wb + 1u) >> 1) + wbox_area(rest) * (float)((nr + 1u) >> 1) : ma * (float)wb + wbox_area(rest) * (float)nr;
                            iso_cost = (c == c) ? c : INF; iso_pos = mp;
                        }
                    }
                }
                float cost = INF;
                if (active && lane < seg_hi - 1) {
                    const uint32_t nl = wl, nr = wall - wl;
                    // pair_cost: a side's cost in LEAVES (two sibling triangles share one, step 4): odd / odd splits of an even count pay for it
                    cost = pair_cost ? wbox_area(pre) * (float)((nl + 1u) >> 1) + wbox_area(right) * (float)((nr + 1u) >> 1)
                                     : wbox_area(pre) * (float)nl + wbox_area(right) * (float)nr;
                    cost = (cost == cost) ? cost : INF;
                }
                // the segment's cheapest position (ties: the leftmost): min-scan from the left, read at the segment's last lane
                float mc = cost; int mp = lane;
                for (int off = 1; off < 64; off <<= 1) {
                    const float oc = __shfl_up(mc, off, 64); const int op = __shfl_up(mp, off, 64);
                    if (lane - off >= seg_lo && oc <= mc) { mc = oc; mp = op; }
                }
                mc = __shfl(mc, seg_hi - 1, 64); mp = __shfl(mp, seg_hi - 1, 64);
                if (mc < best) { best = mc; best_axis = axis; best_pos = mp; }
            }
            // ---- the lanes of every segment in the order of its best axis; an isolated item: the x order with that item moved to the end
            {
                int src = best_axis == 0 ? src_axis[0] : best_axis == 1 ? src_axis[1] : src_axis[2];
                if (iso_cost < best) {
                    const int next = __shfl(src_axis[0], min(lane + 1, 63), 64), big = __shfl(src_axis[0], iso_pos, 64);
                    src = lane < iso_pos ? src_axis[0] : lane < seg_hi - 1 ? next : big;
                    best_pos = seg_hi - 2;
                }
                box = wbox_shfl(box, src);
                item = (uint32_t)__shfl((int)item, src, 64);
                wt = (uint32_t)__shfl((int)wt, src, 64);
            }
            if (first) {                                    // the root keeps its slot: it becomes the slot of the root's split
                if (lane == 0) { const uint32_t t = pool[best_pos]; pool[best_pos] = pool[0]; pool[0] = t; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                first = false;
            }
            if (active) {
                const uint32_t node = pool[best_pos];
                if (lane == seg_lo) {
                    bmin[node] = make_float4(whole.lo[0], whole.lo[1], whole.lo[2], __uint_as_float(wsum));
                    bmax[node] = make_float4(whole.hi[0], whole.hi[1], whole.hi[2], 0.f);
                    if (par >= 0) reinterpret_cast<int*>(&child[pool[par]])[side] = (int)node;
                }
                par = best_pos;
                if (lane <= best_pos) { seg_hi = best_pos + 1; side = 0; } else { seg_lo = best_pos + 1; side = 1; }
                if (seg_hi - seg_lo == 1) reinterpret_cast<int*>(&child[pool[par]])[side] = (int)item;      // one item: final
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();        // (the next root reuses the wave's LDS)
    }
}

// ---- 4. collapse to 4-wide nodes, one level per launch --------------------------------------------------------------------
__device__ __forceinline__ bool leaf_unit(uint32_t x, uint32_t nt, const int2* __restrict__ child)
{
    if (x < nt) return true;                                   // one triangle
    const int2 c = child[x];
    return (uint32_t)c.x < nt && (uint32_t)c.y < nt;           // two sibling triangles: one leaf
}

// tin / tout (optional): the first position in `order` of the triangles below each frontier entry — with the triangle counts of the
// binary nodes (bmin.w) the records of a subtree land next to each other, in depth-first order (what the host builder's index array
// gives); without them the leaves take positions as the levels reach them.
// the (up to four) children a frontier entry's 4-wide node gets: the binary node's two, then the internal child of largest area opened
// until there are four or only leaves are left
__device__ __forceinline__ int collapse_kids(uint32_t X, uint32_t nt, const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                                             const int2* __restrict__ child, uint32_t kids[4])
{
    int nk = 0;
    if (leaf_unit(X, nt, child)) kids[nk++] = X;
    else {
        const int2 c = child[X];
        kids[nk++] = (uint32_t)c.x; kids[nk++] = (uint32_t)c.y;
        while (nk < 4) {                                       // open the internal child with the largest area
            int pick = -1; float pa = -1.f;
            for (int k = 0; k < nk; ++k)
                if (!leaf_unit(kids[k], nt, child)) {
                    const float ar = half_area(bmin[kids[k]], bmax[kids[k]]);
                    if (ar > pa) { pa = ar; pick = k; }
                }
            if (pick < 0) break;
            const int2 c2 = child[kids[pick]];
            kids[pick] = (uint32_t)c2.x; kids[nk++] = (uint32_t)c2.y;
        }
    }
    return nk;
}

// internal children per frontier entry: an exclusive scan of these (hipcub) gives every entry the place of its children in the next
// level — breadth-first order, the same on every run (positions handed out by atomics came out in the order the waves happened to
// run: the million-triangle scene, whose nodes do not fit the L2, was traced 3 % faster or slower from one build to the next)
__global__ __launch_bounds__(256) void k_collapse_count(const uint2* __restrict__ fin, uint32_t nin, uint32_t nt, const float4* __restrict__ bmin,
                                                        const float4* __restrict__ bmax, const int2* __restrict__ child, uint32_t* __restrict__ cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nin) return;
    uint32_t kids[4];
    const int nk = collapse_kids(fin[i].x, nt, bmin, bmax, child, kids);
    uint32_t n_int = 0;
    for (int k = 0; k < nk; ++k) n_int += (kids[k] >= nt && !leaf_unit(kids[k], nt, child)) ? 1u : 0u;
    cnt[i] = n_int;
}

__global__ __launch_bounds__(256) void k_collapse_level(const uint2* __restrict__ fin, uint32_t nin, uint2* __restrict__ fout, uint32_t nt,
                                                        const float4* __restrict__ bmin, const float4* __restrict__ bmax, const int2* __restrict__ child,
                                                        const uint32_t* __restrict__ sorted_ids, rtbvh::Node4* __restrict__ nodes,
                                                        uint32_t* __restrict__ order, uint32_t* __restrict__ ctr, float G,
                                                        const uint32_t* __restrict__ tin, uint32_t* __restrict__ tout,
                                                        const uint32_t* __restrict__ off, uint32_t node_base)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nin) return;
    const uint32_t X = fin[i].x, slot = fin[i].y;
    uint32_t kids[4];
    const int nk = collapse_kids(X, nt, bmin, bmax, child, kids);
    uint32_t n_int = 0, n_tri = 0;
    for (int k = 0; k < nk; ++k) {
        if (kids[k] < nt) n_tri += 1; else if (leaf_unit(kids[k], nt, child)) n_tri += 2; else n_int += 1;
    }
    const uint32_t fbase = off[i], nbase = node_base + fbase;
    if (i == nin - 1u) ctr[kCtrFrontier] = fbase + n_int;              // the next level's size
    uint32_t obase = tin ? tin[i] : n_tri ? atomicAdd(&ctr[kCtrOrder], n_tri) : 0u;
    rtbvh::Node4 N;
    uint32_t ri = 0;
    const float INF = __builtin_inff();
    for (int k = 0; k < 4; ++k) {
        if (k >= nk) {
            N.minx[k] = N.miny[k] = N.minz[k] = INF; N.maxx[k] = N.maxy[k] = N.maxz[k] = -INF; N.child[k] = rtbvh::kEmpty;
            continue;
        }
        const uint32_t x = kids[k];
        const float4 mn = bmin[x], mx = bmax[x];
        const float lo[3] = { mn.x, mn.y, mn.z }, hi[3] = { mx.x, mx.y, mx.z };
        float plo[3], phi[3];
        for (int a = 0; a < 3; ++a) {                          // bvh.cpp pad_box
            const float m = __builtin_fmaxf(__builtin_fabsf(lo[a]), __builtin_fabsf(hi[a]));
            const float e = 3e-5f * m + 2e-6f * G + 1e-30f;
            plo[a] = lo[a] - e; phi[a] = hi[a] + e;
        }
        N.minx[k] = plo[0]; N.miny[k] = plo[1]; N.minz[k] = plo[2];
        N.maxx[k] = phi[0]; N.maxy[k] = phi[1]; N.maxz[k] = phi[2];
        if (x < nt) { order[obase] = sorted_ids[x]; N.child[k] = rtbvh::kLeafBit | (obase << 2) | 0u; obase += 1; }
        else if (leaf_unit(x, nt, child)) {
            const int2 c = child[x];
            order[obase] = sorted_ids[c.x]; order[obase + 1] = sorted_ids[c.y];
            N.child[k] = rtbvh::kLeafBit | (obase << 2) | 1u; obase += 2;
        } else {
            N.child[k] = nbase + ri;
            fout[fbase + ri] = make_uint2(x, nbase + ri);
            if (tin) { tout[fbase + ri] = obase; obase += __float_as_uint(mn.w); }
            ++ri;
        }
    }
    N.meta[0] = (uint32_t)nk; N.meta[1] = N.meta[2] = N.meta[3] = 0u;
    nodes[slot] = N;
}

// ---- 5. worst-case traversal stack, bottom-up over the levels -----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stack_need(const rtbvh::Node4* __restrict__ nodes, uint32_t n0, uint32_t n1, int* __restrict__ need)
{
    const uint32_t i = n0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const int k = (int)nodes[i].meta[0];
    int deepest = 0;
    for (int s = 0; s < k; ++s) { const uint32_t c = nodes[i].child[s]; if (!(c & rtbvh::kLeafBit)) deepest = max(deepest, need[c]); }
    need[i] = (k - 1) + deepest;
}

// sum of the areas of all internal child boxes (the part of the SAH the topology decides): refit quality monitor
__global__ __launch_bounds__(256) void k_internal_area(const rtbvh::Node4* __restrict__ nodes, uint32_t nn, float* __restrict__ out)
{
    float s = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        const rtbvh::Node4& N = nodes[i];
        for (int k = 0; k < 4; ++k)
            if (N.child[k] != rtbvh::kEmpty) {
                const float dx = N.maxx[k] - N.minx[k], dy = N.maxy[k] - N.miny[k], dz = N.maxz[k] - N.minz[k];
                const float a = dx * dy + dy * dz + dz * dx;
                if (a == a && a < 3.0e38f) s += a;
            }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// ---- host driver ------------------------------------------------------------------------------------------------------------------
template <class T> struct Buf {
    T* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = std::max<size_t>(n, 1);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Workspace {
    Buf<float> top_boxes; Buf<int32_t> top_pairs;       // the host-built top of the tree (boxes out, nodes back)
    std::vector<float> h_boxes, h_top_boxes; std::vector<int32_t> h_pairs;
    Buf<uint64_t> keys0, keys1;
    Buf<uint32_t> ids0, ids1, c0, c1, tmp, block_count, block_offset, ctr;
    Buf<unsigned char> sort_tmp;
    Buf<float4> bmin, bmax;
    Buf<int2> child;
    Buf<uint2> f0, f1;
    Buf<uint32_t> t0, t1;                               // collapse: first triangle position of each frontier entry
    Buf<int> need;
    Buf<float> area;
    void release()
    {
        keys0.release(); keys1.release(); ids0.release(); ids1.release(); c0.release(); c1.release(); tmp.release(); block_count.release();
        block_offset.release(); ctr.release(); sort_tmp.release(); bmin.release(); bmax.release(); child.release(); f0.release(); f1.release();
        need.release(); area.release(); top_boxes.release(); top_pairs.release(); t0.release(); t1.release();
    }
};

struct Result {
    uint32_t n_nodes = 0;
    int max_stack = 0, rounds = 0, levels = 0;
    float magnitude = 0.f;                  // G used for the padding
    std::vector<uint32_t> level_start;      // nodes of level L are [level_start[L], level_start[L+1])
};

#define RTGB_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

// tris: nt world-space triangles (18 floats each) on the device.  nodes: room for nt + 1 Node4 records; order: nt entries.
// origin_magnitude: largest |coordinate| of a ray origin outside the triangles (camera, spheres).
// top_clusters: once the bottom-up rounds have left at most this many clusters, the rest of the tree — its top — is built by the
// host's binned-SAH split search over the clusters' boxes (0: the rounds run down to 512 clusters and one workgroup finishes).
// treelets: number of sweep-SAH passes over the clustering's tree (step 3c; 0 = none).
inline hipError_t build(hipStream_t stream, const float* tris, uint32_t nt, float origin_magnitude, int radius, Workspace& w,
                        rtbvh::Node4* nodes, uint32_t* order, Result& out, uint32_t top_clusters, const rtbvh::Tuning& tuning, int treelets = 6, int treelet_ratio = 8, int treelet_isolate = 1, int treelet_first = 1)
{
    const bool widen = radius > 0;              // a negative radius = that radius in every round (A/B of the schedule)
    radius = std::max(1, std::min(radius < 0 ? -radius : radius, kMaxRadius));
    out = Result();
    if (nt == 0) return hipSuccess;
    const uint32_t nb0 = (nt + kPlocBlock - 1) / kPlocBlock;
    RTGB_HIP(w.keys0.ensure(nt)); RTGB_HIP(w.keys1.ensure(nt)); RTGB_HIP(w.ids0.ensure(nt)); RTGB_HIP(w.ids1.ensure(nt));
    RTGB_HIP(w.c0.ensure(nt)); RTGB_HIP(w.c1.ensure(nt)); RTGB_HIP(w.tmp.ensure((size_t)nb0 * kPlocBlock));
    RTGB_HIP(w.block_count.ensure(nb0)); RTGB_HIP(w.block_offset.ensure(nb0)); RTGB_HIP(w.ctr.ensure(32));
    RTGB_HIP(w.bmin.ensure(2 * (size_t)nt)); RTGB_HIP(w.bmax.ensure(2 * (size_t)nt)); RTGB_HIP(w.child.ensure(2 * (size_t)nt));
    RTGB_HIP(w.f0.ensure(nt)); RTGB_HIP(w.f1.ensure(nt)); RTGB_HIP(w.t0.ensure(nt)); RTGB_HIP(w.t1.ensure(nt)); RTGB_HIP(w.need.ensure((size_t)nt + 1)); RTGB_HIP(w.area.ensure(1));

    uint32_t h_ctr[32] = {};
    for (int a = 0; a < 3; ++a) { h_ctr[kCtrBounds + a] = 0xFFFFFFFFu; h_ctr[kCtrBounds + 3 + a] = 0u; }
    h_ctr[kCtrNode4] = 1u;                                     // node 0 = the root
    RTGB_HIP(hipMemcpyAsync(w.ctr.p, h_ctr, sizeof h_ctr, hipMemcpyHostToDevice, stream));
    const int grid = (int)std::min<uint32_t>((nt + 255) / 256, 2048);
    hipLaunchKernelGGL(k_tri_bounds, dim3(grid), dim3(256), 0, stream, tris, nt, w.ctr.p);
    hipLaunchKernelGGL(k_morton, dim3((nt + 255) / 256), dim3(256), 0, stream, tris, nt, w.ctr.p, w.keys0.p, w.ids0.p);
    size_t tmp_bytes = 0;
    RTGB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, w.keys0.p, w.keys1.p, w.ids0.p, w.ids1.p, (int)nt, 0, 63, stream));
    RTGB_HIP(w.sort_tmp.ensure(tmp_bytes));
    RTGB_HIP(hipcub::DeviceRadixSort::SortPairs(w.sort_tmp.p, tmp_bytes, w.keys0.p, w.keys1.p, w.ids0.p, w.ids1.p, (int)nt, 0, 63, stream));
    hipLaunchKernelGGL(k_leaf_boxes, dim3((nt + 255) / 256), dim3(256), 0, stream, tris, w.ids1.p, nt, w.bmin.p, w.bmax.p, w.c0.p);

    // ---- PLOC rounds
    uint32_t m = nt;
    uint32_t* cin = w.c0.p; uint32_t* cout = w.c1.p;
    const uint32_t stop_at = std::max<uint32_t>((uint32_t)kPlocBlock, top_clusters);
    while (m > stop_at) {
        const uint32_t nb = (m + kPlocBlock - 1) / kPlocBlock;
        // wider search as the clusters get fewer and larger (the upper levels decide most of a ray's node visits, and cost little)
        const int r_now = (widen && m <= nt / 16) ? std::min(kMaxRadius, 4 * radius) : (widen && m <= nt / 4) ? std::min(kMaxRadius, 2 * radius) : radius;
        hipLaunchKernelGGL(k_ploc_round, dim3(nb), dim3(kPlocBlock), 0, stream, cin, m, nt, w.bmin.p, w.bmax.p, w.child.p, w.ctr.p, w.tmp.p, w.block_count.p, r_now);
        hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(1024), 0, stream, w.block_count.p, nb, w.block_offset.p, w.ctr.p);
        hipLaunchKernelGGL(k_ploc_gather, dim3(nb), dim3(kPlocBlock), 0, stream, w.tmp.p, w.block_count.p, w.block_offset.p, cout);
        uint32_t m_next = 0;
        RTGB_HIP(hipMemcpyAsync(&m_next, w.ctr.p + kCtrClusters, sizeof m_next, hipMemcpyDeviceToHost, stream));
        RTGB_HIP(hipStreamSynchronize(stream));
        if (m_next == 0 || m_next >= m) return hipErrorUnknown;           // (cannot happen: every round merges at least one pair)
        m = m_next; std::swap(cin, cout); ++out.rounds;
    }
    uint32_t root = 0; uint32_t h_bounds[8];
    if (top_clusters > 0 && m >= 2) {
        // ---- the top of the tree on the host: m boxes out (24 B each), m - 1 nodes back
        uint32_t made = 0;
        RTGB_HIP(w.top_boxes.ensure(6 * (size_t)m)); RTGB_HIP(w.top_pairs.ensure(2 * (size_t)m));
        hipLaunchKernelGGL(k_cluster_boxes, dim3((m + 255) / 256), dim3(256), 0, stream, cin, m, w.bmin.p, w.bmax.p, w.top_boxes.p);
        w.h_boxes.resize(6 * (size_t)m);
        RTGB_HIP(hipMemcpyAsync(w.h_boxes.data(), w.top_boxes.p, w.h_boxes.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        RTGB_HIP(hipMemcpyAsync(&made, w.ctr.p + kCtrNodes, sizeof made, hipMemcpyDeviceToHost, stream));
        RTGB_HIP(hipMemcpyAsync(h_bounds, w.ctr.p + kCtrBounds, sizeof h_bounds, hipMemcpyDeviceToHost, stream));
        RTGB_HIP(hipStreamSynchronize(stream));
        rtbvh::build_over_boxes(w.h_boxes.data(), m, tuning, w.h_pairs);
        const uint32_t n_top = (uint32_t)(w.h_pairs.size() / 2);            // = m - 1
        // boxes of the host's nodes, children before parents (pre-order numbering: a child's index is larger than its parent's)
        w.h_top_boxes.assign(6 * (size_t)n_top, 0.f);
        for (uint32_t k = n_top; k-- > 0; ) {
            float* b = &w.h_top_boxes[6 * (size_t)k];
            for (int a = 0; a < 3; ++a) { b[a] = __builtin_inff(); b[3 + a] = -__builtin_inff(); }
            for (int side = 0; side < 2; ++side) {
                const int32_t c = w.h_pairs[2 * (size_t)k + side];
                const float* cb = c >= 0 ? &w.h_boxes[6 * (size_t)c] : &w.h_top_boxes[6 * (size_t)~c];
                for (int a = 0; a < 3; ++a) { b[a] = std::fmin(b[a], cb[a]); b[3 + a] = std::fmax(b[3 + a], cb[3 + a]); }
            }
        }
        RTGB_HIP(w.top_boxes.ensure(6 * (size_t)std::max(m, n_top)));
        RTGB_HIP(hipMemcpyAsync(w.top_pairs.p, w.h_pairs.data(), w.h_pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        RTGB_HIP(hipMemcpyAsync(w.top_boxes.p, w.h_top_boxes.data(), w.h_top_boxes.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_top_nodes, dim3((n_top + 255) / 256), dim3(256), 0, stream, w.top_pairs.p, w.top_boxes.p, n_top, nt + made, cin,
                           w.bmin.p, w.bmax.p, w.child.p);
        root = nt + made;                                                   // the host's node 0
        RTGB_HIP(hipStreamSynchronize(stream));                             // (the staging vectors are reused by the next build)
    } else {
        hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(kPlocBlock), 0, stream, cin, m, nt, w.bmin.p, w.bmax.p, w.child.p, w.ctr.p, radius);
        RTGB_HIP(hipMemcpyAsync(&root, w.ctr.p + kCtrClusters + 1, sizeof root, hipMemcpyDeviceToHost, stream));
        RTGB_HIP(hipMemcpyAsync(h_bounds, w.ctr.p + kCtrBounds, sizeof h_bounds, hipMemcpyDeviceToHost, stream));
        RTGB_HIP(hipStreamSynchronize(stream));
    }
    const float G = std::max(origin_magnitude, ord2f(h_bounds[6]));
    out.magnitude = G;

    // ---- the bottom of the tree again, by the sweep SAH (the clustering's nodes only: the top's clusters are the candidates beside them)
    if (treelets && nt >= 3) {
        const bool has_top = top_clusters > 0 && m >= 2;
        const uint32_t n_extra = has_top ? m : 1u;
        const uint32_t* extra = has_top ? cin : w.ctr.p + kCtrClusters + 1;
        // pass k: items = subtrees of <= ratio^k triangles (as the pass before left them), roots = maximal subtrees of <= 64 * ratio^k:
        // every pass regroups what the one below it built, across the boundaries the clustering had drawn at that scale; with enough
        // passes (6 for a million triangles at ratio 8) the last one has the whole tree as its one treelet
        uint32_t item_tris = (uint32_t)treelet_first;
        for (int pass = 0; pass < treelets && pass < 16 && (pass == 0 || item_tris < nt); ++pass, item_tris *= (uint32_t)treelet_ratio) {
            const uint32_t max_tris = (uint32_t)kTreeletMax * item_tris;
            hipLaunchKernelGGL(k_treelet_roots, dim3((std::max(nt, n_extra) + 255) / 256), dim3(256), 0, stream, nt, w.bmin.p, w.child.p, extra, n_extra, w.ctr.p, w.tmp.p,
                               max_tris, 2u * item_tris + 1u, pass);
            const uint32_t blocks = std::min<uint32_t>(2048u, nt / (32u * item_tris) + 1u);
            hipLaunchKernelGGL(k_treelet_sah, dim3(blocks), dim3(256), 0, stream, w.tmp.p, w.ctr.p, nt, w.bmin.p, w.bmax.p, w.child.p,
                               item_tris == 1 ? 1 : 0, item_tris, pass, treelet_isolate);
        }
    }

    // ---- collapse, level by level
    const uint2 first = make_uint2(root, 0u);
    RTGB_HIP(hipMemcpyAsync(w.f0.p, &first, sizeof first, hipMemcpyHostToDevice, stream));
    uint2* fin = w.f0.p; uint2* fout = w.f1.p;
    // every binary node carries its triangle count unless the host built the top (its nodes do not): depth-first triangle order then
    const bool dfs_order = !(top_clusters > 0 && m >= 2);
    uint32_t* tin = dfs_order ? w.t0.p : nullptr; uint32_t* tout = dfs_order ? w.t1.p : nullptr;
    if (dfs_order) RTGB_HIP(hipMemsetAsync(w.t0.p, 0, sizeof(uint32_t), stream));
    uint32_t nin = 1, total = 1;
    out.level_start.push_back(0);
    uint32_t* const lvl_cnt = w.c0.p; uint32_t* const lvl_off = w.c1.p;       // (the cluster arrays are free by now)
    size_t scan_bytes = 0;
    RTGB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, lvl_cnt, lvl_off, (int)nt, stream));
    RTGB_HIP(w.sort_tmp.ensure(scan_bytes));
    while (nin > 0) {
        hipLaunchKernelGGL(k_collapse_count, dim3((nin + 255) / 256), dim3(256), 0, stream, fin, nin, nt, w.bmin.p, w.bmax.p, w.child.p, lvl_cnt);
        size_t sb = w.sort_tmp.cap;
        RTGB_HIP(hipcub::DeviceScan::ExclusiveSum(w.sort_tmp.p, sb, lvl_cnt, lvl_off, (int)nin, stream));
        hipLaunchKernelGGL(k_collapse_level, dim3((nin + 255) / 256), dim3(256), 0, stream, fin, nin, fout, nt, w.bmin.p, w.bmax.p, w.child.p,
                           w.ids1.p, nodes, order, w.ctr.p, G, tin, tout, lvl_off, total);
        uint32_t nout = 0;
        RTGB_HIP(hipMemcpyAsync(&nout, w.ctr.p + kCtrFrontier, sizeof nout, hipMemcpyDeviceToHost, stream));
        RTGB_HIP(hipStreamSynchronize(stream));
        out.level_start.push_back(total);
        total += nout; nin = nout; std::swap(fin, fout); std::swap(tin, tout); ++out.levels;
    }
    out.n_nodes = total;
    if (out.level_start.back() != total) out.level_start.push_back(total);
    // ---- exact worst-case stack depth
    for (int L = (int)out.level_start.size() - 2; L >= 0; --L) {
        const uint32_t n0 = out.level_start[L], n1 = out.level_start[L + 1];
        if (n1 > n0) hipLaunchKernelGGL(k_stack_need, dim3((n1 - n0 + 255) / 256), dim3(256), 0, stream, nodes, n0, n1, w.need.p);
    }
    int need0 = 0;
    RTGB_HIP(hipMemcpyAsync(&need0, w.need.p, sizeof need0, hipMemcpyDeviceToHost, stream));
    RTGB_HIP(hipStreamSynchronize(stream));
    out.max_stack = std::max(1, need0);
    return hipGetLastError();
}

// the same without the synchronisation: the sum arrives in *host_area once the stream has been synchronised
inline hipError_t internal_area_async(hipStream_t stream, const rtbvh::Node4* nodes, uint32_t nn, Workspace& w, float* host_area)
{
    *host_area = 0.f;
    if (!nn) return hipSuccess;
    RTGB_HIP(w.area.ensure(1));
    RTGB_HIP(hipMemsetAsync(w.area.p, 0, sizeof(float), stream));
    hipLaunchKernelGGL(k_internal_area, dim3((int)std::min<uint32_t>((nn + 255) / 256, 1024)), dim3(256), 0, stream, nodes, nn, w.area.p);
    return hipMemcpyAsync(host_area, w.area.p, sizeof(float), hipMemcpyDeviceToHost, stream);
}

// sum of internal child-box areas of the current nodes (synchronous, one float back)
inline hipError_t internal_area(hipStream_t stream, const rtbvh::Node4* nodes, uint32_t nn, Workspace& w, float& area)
{
    area = 0.f;
    if (!nn) return hipSuccess;
    RTGB_HIP(w.area.ensure(1));
    RTGB_HIP(hipMemsetAsync(w.area.p, 0, sizeof(float), stream));
    hipLaunchKernelGGL(k_internal_area, dim3((int)std::min<uint32_t>((nn + 255) / 256, 1024)), dim3(256), 0, stream, nodes, nn, w.area.p);
    RTGB_HIP(hipMemcpyAsync(&area, w.area.p, sizeof(float), hipMemcpyDeviceToHost, stream));
    return hipStreamSynchronize(stream);
}

#undef RTGB_HIP

} // namespace rtgb
