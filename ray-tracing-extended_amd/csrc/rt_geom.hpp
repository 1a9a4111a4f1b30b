// rt_geom.hpp — on-device geometry pipeline: the step immediately before the tracer.
//
// The reference re-transforms every triangle of every mesh to world space on the CPU and re-uploads the whole scene
// every frame (RayTracedMesh.GetSubMeshes / UpdateWorldChunkFromLocal, "Assets/Scripts/Render Types/RayTracedMesh.cs":
// 36-84; RayTracingManager.CreateMeshes, Assets/Scripts/RayTracingManager.cs:135-164; its own TODO at
// RayTracedMesh.cs:37: "upload matrices to gpu to avoid having to contantly upload all mesh data").  Here the local
// chunks are uploaded once and a frame only sends the per-mesh transforms (40 bytes each); these kernels then
//   k_transform      local -> world triangles: rot * Scale(p, lossyScale) + pos, normals rot * n  (:86-94), float32 in
//                    the operation order of UnityEngine's Quaternion * Vector3 so the bytes equal the host marshal's;
//   k_chunk_bounds   tight world AABB per chunk, stored the way MeshInfo reads it back from a UnityEngine.Bounds
//                    (centre/size round trip: RayTracedMesh.cs:82, MeshInfo.cs:16-17);
//   k_relayout       BVH-order tracer records (A, eAB, eAC, cross | normals, chunk, visiting rank);
//   k_refit_level    bottom-up refit of the BVH4 boxes (topology kept), one launch per tree level.
// All are streaming kernels: 240 B of HBM traffic per triangle end to end.
#pragma once
#include "rt_kernels.hpp"

namespace rtg {

using rtm::v3;

struct MeshXf { float px, py, pz, qx, qy, qz, qw, sx, sy, sz; };            // rt_mesh_transform

// UnityEngine Quaternion * Vector3 (same operation order as host.py quat_rotate)
__device__ __forceinline__ v3 quat_rotate(const MeshXf& t, v3 p)
{
    const float x = t.qx, y = t.qy, z = t.qz, w = t.qw;
    const float x2 = x * 2.0f, y2 = y * 2.0f, z2 = z * 2.0f;
    const float xx = x * x2, yy = y * y2, zz = z * z2;
    const float xy = x * y2, xz = x * z2, yz = y * z2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    v3 r;
    r.x = ((1.0f - (yy + zz)) * p.x + (xy - wz) * p.y) + (xz + wy) * p.z;
    r.y = ((xy + wz) * p.x + (1.0f - (xx + zz)) * p.y) + (yz - wx) * p.z;
    r.z = ((xz - wy) * p.x + (yz + wx) * p.y) + (1.0f - (xx + yy)) * p.z;
    return r;
}

// one thread per triangle: 72 B in, 72 B out
__global__ __launch_bounds__(256) void k_transform(const float* __restrict__ local_tris, const uint32_t* __restrict__ tri_mesh,
                                                   const MeshXf* __restrict__ xf, float* __restrict__ world_tris, uint32_t nt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const MeshXf t = xf[tri_mesh[i]];
    const float* s = local_tris + (size_t)i * 18;
    float* d = world_tris + (size_t)i * 18;
#pragma unroll
    for (int k = 0; k < 3; ++k) {                       // PointLocalToWorld (RayTracedMesh.cs:86-89)
        v3 p = rtm::mk(s[3 * k] * t.sx, s[3 * k + 1] * t.sy, s[3 * k + 2] * t.sz);
        v3 r = quat_rotate(t, p);
        d[3 * k] = r.x + t.px; d[3 * k + 1] = r.y + t.py; d[3 * k + 2] = r.z + t.pz;
    }
#pragma unroll
    for (int k = 3; k < 6; ++k) {                       // DirectionLocalToWorld (:91-94)
        v3 r = quat_rotate(t, rtm::mk(s[3 * k], s[3 * k + 1], s[3 * k + 2]));
        d[3 * k] = r.x; d[3 * k + 1] = r.y; d[3 * k + 2] = r.z;
    }
}

// one thread per chunk
__global__ __launch_bounds__(256) void k_chunk_bounds(const float* __restrict__ world_tris, const uint32_t* __restrict__ range,
                                                      float4* __restrict__ chunk_box, uint32_t nm)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nm) return;
    const uint32_t first = range[2 * m], count = range[2 * m + 1];
    float mn[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, mx[3] = { -mn[0], -mn[0], -mn[0] };
    for (uint32_t t = 0; t < count; ++t) {
        const float* p = world_tris + (size_t)(first + t) * 18;
        for (int k = 0; k < 9; ++k) { mn[k % 3] = __builtin_fminf(mn[k % 3], p[k]); mx[k % 3] = __builtin_fmaxf(mx[k % 3], p[k]); }
    }
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        // new Bounds((min + max) / 2, max - min)  ->  MeshInfo: bounds.min / bounds.max = centre -/+ size * 0.5
        const float centre = (mn[a] + mx[a]) / 2.0f, size = mx[a] - mn[a];
        lo[a] = centre - size * 0.5f; hi[a] = centre + size * 0.5f;
    }
    if (count == 0) for (int a = 0; a < 3; ++a) { lo[a] = 0.f; hi[a] = 0.f; }
    chunk_box[2 * m]     = make_float4(lo[0], lo[1], lo[2], 0.f);
    chunk_box[2 * m + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

// one thread per triangle in BVH order
__global__ __launch_bounds__(256) void k_relayout(const float* __restrict__ world_tris, const uint32_t* __restrict__ order,
                                                  const uint32_t* __restrict__ tri_chunk, const uint32_t* __restrict__ tri_rank,
                                                  float4* __restrict__ tri_geo, float4* __restrict__ tri_nrm, uint32_t nl)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nl) return;
    const uint32_t orig = order[i];
    const float* t = world_tris + (size_t)orig * 18;
    const float ex = t[3] - t[0], ey = t[4] - t[1], ez = t[5] - t[2];          // RayTriangle :152-154
    const float fx = t[6] - t[0], fy = t[7] - t[1], fz = t[8] - t[2];
    const float nx = ey * fz - ez * fy, ny = ez * fx - ex * fz, nz = ex * fy - ey * fx;
    if (tri_chunk[orig] == 0xFFFFFFFFu) {
        // a triangle no chunk addresses is never visited by the reference's loops (RayTracing.shader:276-294): NaN records can never be hit
        const float q = __uint_as_float(0x7FC00000u);
        tri_geo[3 * (size_t)i + 0] = tri_geo[3 * (size_t)i + 1] = tri_geo[3 * (size_t)i + 2] = make_float4(q, q, q, q);
    } else {
    tri_geo[3 * (size_t)i + 0] = make_float4(t[0], t[1], t[2], ex);
    tri_geo[3 * (size_t)i + 1] = make_float4(ey, ez, fx, fy);
    tri_geo[3 * (size_t)i + 2] = make_float4(fz, nx, ny, nz);
    }
    tri_nrm[3 * (size_t)i + 0] = make_float4(t[9], t[10], t[11], __uint_as_float(tri_chunk[orig]));
    tri_nrm[3 * (size_t)i + 1] = make_float4(t[12], t[13], t[14], __uint_as_float(tri_rank[orig]));   // tie-break key: visiting rank
    tri_nrm[3 * (size_t)i + 2] = make_float4(t[15], t[16], t[17], 0.f);
}

// Refit one level of the BVH4 (nodes [n0, n1)), one thread per (node, slot); deeper levels must be done already.
// Leaf slots: padded bounds of their triangles (same padding as bvh.cpp pad_box); internal slots: union of the child
// node's four slot boxes.
__global__ __launch_bounds__(256) void k_refit_level(rtbvh::Node4* __restrict__ nodes, uint32_t n0, uint32_t n1,
                                                     const float* __restrict__ world_tris, const uint32_t* __restrict__ order, float G)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = n0 + (tid >> 2), k = tid & 3u;
    if (node >= n1) return;
    rtbvh::Node4& N = nodes[node];
    const uint32_t c = N.child[k];
    if (c == rtbvh::kEmpty) return;
    float mn[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, mx[3] = { -mn[0], -mn[0], -mn[0] };
    if (c & rtbvh::kLeafBit) {
        const uint32_t first = (c & 0x7FFFFFFFu) >> 2, count = (c & 3u) + 1u;
        for (uint32_t j = 0; j < count; ++j) {
            const float* p = world_tris + (size_t)order[first + j] * 18;
            for (int q = 0; q < 9; ++q) { mn[q % 3] = __builtin_fminf(mn[q % 3], p[q]); mx[q % 3] = __builtin_fmaxf(mx[q % 3], p[q]); }
        }
        for (int a = 0; a < 3; ++a) {
            const float m = __builtin_fmaxf(__builtin_fabsf(mn[a]), __builtin_fabsf(mx[a]));
            const float e = 3e-5f * m + 2e-6f * G + 1e-30f;
            mn[a] -= e; mx[a] += e;
        }
    } else {
        const rtbvh::Node4& C = nodes[c];
        for (int s = 0; s < 4; ++s) {
            mn[0] = __builtin_fminf(mn[0], C.minx[s]); mn[1] = __builtin_fminf(mn[1], C.miny[s]); mn[2] = __builtin_fminf(mn[2], C.minz[s]);
            mx[0] = __builtin_fmaxf(mx[0], C.maxx[s]); mx[1] = __builtin_fmaxf(mx[1], C.maxy[s]); mx[2] = __builtin_fmaxf(mx[2], C.maxz[s]);
        }
    }
    N.minx[k] = mn[0]; N.miny[k] = mn[1]; N.minz[k] = mn[2];
    N.maxx[k] = mx[0]; N.maxy[k] = mx[1]; N.maxz[k] = mx[2];
}

// Node4 -> Node4h (bvh.hpp): one thread per node, 128 B in, 128 B out.  Runs after the host build's upload and after every refit.
__global__ __launch_bounds__(256) void k_compact_nodes(const rtbvh::Node4* __restrict__ nodes, rtbvh::Node4h* __restrict__ out, uint32_t nn)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nn) return;
    const rtbvh::Node4 N = nodes[i];
    const float* mins[3] = { N.minx, N.miny, N.minz };
    const float* maxs[3] = { N.maxx, N.maxy, N.maxz };
    float O[3];
    uint16_t hmin[3][4], hmax[3][4];
    const float INF = __builtin_inff();
    for (int a = 0; a < 3; ++a) {
        float lo = INF, hi = -INF;
        for (int k = 0; k < 4; ++k) {
            if (mins[a][k] > -INF && mins[a][k] < INF) lo = __builtin_fminf(lo, mins[a][k]);
            if (maxs[a][k] > -INF && maxs[a][k] < INF) hi = __builtin_fmaxf(hi, maxs[a][k]);
        }
        float o = 0.5f * lo + 0.5f * hi;
        if (!(o > -INF && o < INF)) o = (lo < INF) ? lo : (hi > -INF ? hi : 0.0f);
        O[a] = o;
        for (int k = 0; k < 4; ++k) {
            hmin[a][k] = rtbvh::f16_round_down(rtbvh::f32_below(mins[a][k] - o));
            hmax[a][k] = rtbvh::f16_round_up(rtbvh::f32_above(maxs[a][k] - o));
        }
    }
    rtbvh::Node4h H;
    auto pack4 = [](const uint16_t* h, uint32_t* w) { w[0] = (uint32_t)h[0] | ((uint32_t)h[1] << 16); w[1] = (uint32_t)h[2] | ((uint32_t)h[3] << 16); };
    for (int c = 0; c < 4; ++c) {
        pack4((c & 1) ? hmax[0] : hmin[0], &H.w[4 * c]);
        pack4((c & 2) ? hmax[1] : hmin[1], &H.w[4 * c + 2]);
    }
    pack4(hmin[2], &H.w[16]); pack4(hmax[2], &H.w[18]);
    pack4(hmax[2], &H.w[20]); pack4(hmin[2], &H.w[22]);
    for (int k = 0; k < 4; ++k) H.w[24 + k] = N.child[k];
    H.w[28] = __float_as_uint(O[0]); H.w[29] = __float_as_uint(O[1]); H.w[30] = __float_as_uint(O[2]); H.w[31] = 0u;
    out[i] = H;
}

// ---- costliest-first tile order on the device (LPT scheduling of the persistent waves): a counting sort of the tiles by the
// cost the last launch recorded, 8192 buckets of 3 % width (exponent + 5 mantissa bits of the cost as a float), largest first.
// Three tiny launches on the render stream instead of a device -> host copy, a host sort and a copy back: a camera that moves
// every frame re-measures its tile costs every frame.  The order inside a bucket is whatever the atomics give; the image never
// depends on the order.
constexpr int kCostBuckets = 8192;
__device__ __forceinline__ uint32_t cost_bucket(uint32_t cost)
{
    const uint32_t b = __float_as_uint((float)cost) >> 18;           // <= (158 << 5 | 31) = 5087
    return (uint32_t)(kCostBuckets - 1) - (b < (uint32_t)kCostBuckets ? b : (uint32_t)(kCostBuckets - 1));
}
__global__ __launch_bounds__(256) void k_tile_hist(const uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ hist)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&hist[cost_bucket(cost[i])], 1u);
}
__global__ __launch_bounds__(1024) void k_tile_scan(uint32_t* __restrict__ hist)      // one block: exclusive prefix sum in place
{
    __shared__ uint32_t part[1024];
    constexpr int per = kCostBuckets / 1024;
    uint32_t v[per], sum = 0;
    for (int k = 0; k < per; ++k) { v[k] = hist[threadIdx.x * per + k]; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int k = 0; k < per; ++k) { hist[threadIdx.x * per + k] = run; run += v[k]; }
}
__global__ __launch_bounds__(256) void k_tile_scatter(const uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ order)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[atomicAdd(&offsets[cost_bucket(cost[i])], 1u)] = i;
}

// ---- display step after the path: linear RGBA32F -> sRGB RGBA8 (the reference's final Blit(resultTexture, target) into an
// sRGB back buffer, RayTracingManager.cs:84, ProjectSettings.asset:50).  16 B read + 4 B write per pixel.
__device__ __forceinline__ float linear_to_srgb(float c)
{
    c = rtm::saturate(c);
    return (c <= 0.0031308f) ? 12.92f * c : 1.055f * rtm::pow_(c, 0.41666666f) - 0.055f;
}
__device__ __forceinline__ uint32_t to_unorm8(float v) { return (uint32_t)__builtin_floorf(rtm::saturate(v) * 255.0f + 0.5f); }

__global__ __launch_bounds__(256) void k_display_srgb8(const float4* __restrict__ rgba, uint32_t* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 p = rgba[i];
        out[i] = to_unorm8(linear_to_srgb(p.x)) | (to_unorm8(linear_to_srgb(p.y)) << 8)
               | (to_unorm8(linear_to_srgb(p.z)) << 16) | (to_unorm8(p.w) << 24);
    }
}

} // namespace rtg
