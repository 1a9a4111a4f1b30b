// rt_primary.hpp — per-pixel candidate lists for camera rays (round 4).
//
// 55 % of the rays of the headline workload are camera rays, and with the defocus jitter off (frag :377-378 with DefocusStrength = 0)
// all camera rays of a pixel start at one point and pass through one small footprint: the pixel's focus point plus the divergence
// jitter, a disc of radius DivergeStrength / width in the (camRight, camUp) plane (frag :380-382).  The reference walks its whole chunk
// list for every one of them (RayTracing.shader:276-294); the BVH replaced that by ~9 node steps per ray; for most pixels a handful
// of triangles is all that any ray through the footprint can possibly hit first.  k_primary_lists finds them once per camera / scene:
//
//   1. the four corner rays of the footprint's bounding square (widened: see below) are traced to their closest hit with the tracer's own
//      traversal (float, f16 nodes, LDS stack): these hits only NOMINATE triangles, nothing is concluded from them;
//   2. a triangle T that ALL FOUR corner rays hit with barycentrics >= kBaryMargin and determinant >= kDetMargin is hit by every ray
//      of the footprint (a central projection maps the square's convex hull onto a convex region of T's plane inside T), no farther
//      than the farthest corner hit, and robustly so in the kernels' float arithmetic (the margins dwarf its rounding errors): the
//      closest hit of every camera ray of the pixel OVER ALL TRIANGLES therefore lies within t_max = that distance (+ margin) — which
//      is what k_stream's first traversal of a ray looks for: the reference's chunk cull (RT_INTERSECT_FLAT_CHUNKS) is applied to its
//      answer in SHADE, and a ray whose answer fails it is traced again from the root, without a list.  Without such a T, t_max = infinity;
//   3. every leaf whose (padded) box meets the footprint's frustum — four planes through the camera position — within t_max is a
//      candidate; a triangle in any other leaf cannot be the closest hit of a ray of this pixel: a hit the kernels' arithmetic accepts
//      lies inside its leaf's padded box (that is what the padding is for, bvh.cpp pad_box), and the ray lies inside the frustum.
//
// Up to four candidate leaves become the pixel's list; a camera ray then starts its query with those leaves on its traversal stack
// and no node step at all (rt_stream.hpp), and the closest hit is decided exactly as before — RayTriangle's arithmetic, (dst, visiting
// rank), the chunk filter — over a superset of the triangles the full traversal would have tested before accepting the same hit.  More
// than four candidates, depth of field, or a degenerate footprint: no list, the ray starts at the root as always.  The image cannot
// depend on the lists (tests: with and without them, against the oracle).
//
// Everything here runs in double precision on the float scene data, with the footprint widened by kWiden and an absolute slack that
// covers the float rounding of the kernels' own ray generation; the frustum test is conservative in the same way.
#pragma once
#include "rt_kernels.hpp"

namespace rtp {

constexpr uint32_t kNoList = 0xFFFFFFFEu;       // lists[pixel].x: no list, start at the root (rtk::kNone in .x = certain miss: an empty list)
constexpr int kMaxList = 4;
constexpr double kWiden = 1.05, kBaryMargin = 0.01, kDetMargin = 1.5e-6, kTmaxMargin = 1.0001;

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return { a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }


struct PrimaryArgs {
    rt_params p;
    int row0, nrows, row_stride;        // as FrameArgs: local row ly -> global row row0 + (ly / 8) * row_stride + ly % 8
    uint4* lists;                       // [nrows * width]
    float4* focus;                      // [nrows * width] the pixel's focus point (frag :364-366), which every one of its camera rays needs again
    int stack_cap;                      // LDS stack entries per lane (the BVH's worst case)
    unsigned int* counts;               // [4] pixels with a list bounded by a common triangle, with an unbounded list, certainly missing everything, without a list
};

// RayTriangle (:150-174) in double on the tracer's own operands (A, eAB, eAC, cross(eAB, eAC) of the BVH-order record); d = unit direction
__device__ __forceinline__ bool tri_hit(const float4* __restrict__ tri_geo, uint32_t ti, D3 o, D3 d, double& dst, double& u, double& v, double& det)
{
    const float4 g0 = tri_geo[3 * (size_t)ti], g1 = tri_geo[3 * (size_t)ti + 1], g2 = tri_geo[3 * (size_t)ti + 2];
    const D3 A{ g0.x, g0.y, g0.z }, e1{ g0.w, g1.x, g1.y }, e2{ g1.z, g1.w, g2.x }, n{ g2.y, g2.z, g2.w };
    const D3 ao = o - A, dao = cross(ao, d);
    det = -dot(d, n);
    if (!(det > 0.0)) return false;
    const double inv = 1.0 / det;
    dst = dot(ao, n) * inv; u = dot(e2, dao) * inv; v = -dot(e1, dao) * inv;
    return dst >= 0.0 && u >= 0.0 && v >= 0.0 && 1.0 - u - v >= 0.0;
}

__global__ __launch_bounds__(64) void k_primary_lists(rtk::DeviceScene S, PrimaryArgs A)
{
    // one lane per pixel, 8 x 8 pixels per wave (the corner rays of neighbouring pixels visit the same nodes)
    const int tiles_x = (A.p.width + 7) / 8;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 8 + (threadIdx.x & 7), ly = ty * 8 + (threadIdx.x >> 3);
    if (x >= A.p.width || ly >= A.nrows) return;
    const int y = A.row0 + (ly >> 3) * A.row_stride + (ly & 7);
    const rt_params& p = A.p;
    const float* M = p.camLocalToWorld;
    uint4 out = make_uint4(kNoList, rtk::kNone, rtk::kNone, rtk::kNone);
    int kind = 3;
    // the traversal stack of this lane: column `lane` of stack[entry][lane] in LDS (A.stack_cap entries), as in the tracer
    extern __shared__ uint32_t lds_stack[];
    rtk::TravStack stk; stk.lds = lds_stack + threadIdx.x; stk.cap = A.stack_cap; stk.glb = nullptr; stk.stride = 0;
    rtk::Counters cnt = {};
    rtk::DeviceScene T = S; T.ns = 0;                                    // (spheres are tested for every ray anyway: the lists are about triangles)
    do {
        // the focus point exactly as the kernels compute it (frag :364-366 in float: rt_stream.hpp camera block)
        const float Wf = (float)(uint32_t)p.width;
        const float uvx = ((float)x + 0.5f) / Wf, uvy = ((float)y + 0.5f) / (float)(uint32_t)p.height;
        const float lx = (uvx - 0.5f) * p.viewParams[0], lyv = (uvy - 0.5f) * p.viewParams[1], lz = 1.0f * p.viewParams[2];
        const float fpx = ((M[0] * lx + M[1] * lyv) + M[2] * lz) + M[3] * 1.0f, fpy = ((M[4] * lx + M[5] * lyv) + M[6] * lz) + M[7] * 1.0f,
                    fpz = ((M[8] * lx + M[9] * lyv) + M[10] * lz) + M[11] * 1.0f;
        A.focus[(size_t)ly * A.p.width + x] = make_float4(fpx, fpy, fpz, 0.0f);
        const D3 fp{ fpx, fpy, fpz }, right{ M[0], M[4], M[8] }, up{ M[1], M[5], M[9] };
        const D3 pos{ p.worldSpaceCameraPos[0], p.worldSpaceCameraPos[1], p.worldSpaceCameraPos[2] };
        // footprint: |jitter| <= |DivergeStrength| / width in both basis directions (RandomPointInCircle has radius <= 1; its sqrt / cos / sin
        // are bounded by 1 + a few ulp), widened by kWiden and by the float rounding of jfp = (fp + right * jx) + up * jy and of the normalisation
        const double rho = fabs((double)p.divergeStrength) / (double)Wf;
        const double coord = fmax(fmax(fabs(fp.x), fabs(fp.y)), fabs(fp.z)) + fmax(fmax(fabs(pos.x), fabs(pos.y)), fabs(pos.z));
        const double bl = sqrt(fmin(dot(right, right), dot(up, up)));
        if (!(bl > 1e-3) || !(coord < 1e30) || !(rho < 1e30)) break;                    // degenerate camera basis or non-finite input: no list
        const double r = rho * kWiden + 16.0 * 1.1920929e-7 * coord / bl;
        D3 dir[4], unit[4];                     // corner directions, as they are and at unit length
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const D3 tgt = fp + right * ((c & 1) ? r : -r) + up * ((c & 2) ? r : -r);
            dir[c] = tgt - pos;
            const double len = sqrt(dot(dir[c], dir[c]));
            if (!(len > 1e-30) || !(len < 1e30)) ok = false;
            unit[c] = dir[c] * (1.0 / len);
        }
        if (!ok) break;
        const D3 centre = fp - pos;
        // ---- 1. the corner rays' closest hits nominate triangles (the tracer's own float traversal; no chunk cull)
        int hit[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const rtm::v3 of = rtm::mk(p.worldSpaceCameraPos[0], p.worldSpaceCameraPos[1], p.worldSpaceCameraPos[2]);
            const rtm::v3 df = rtm::normalize(rtm::mk((float)dir[c].x, (float)dir[c].y, (float)dir[c].z));
            const rtk::Hit h = rtk::closest_hit<false, true>(T, RT_INTERSECT_BRUTE, false, of, df, stk, cnt);
            hit[c] = (h.id != rtk::kNone && (h.id & rtk::kTriBit)) ? (int)(h.id & ~rtk::kTriBit) : -1;
        }
        // ---- 2. a triangle every corner ray hits well inside: the one whose farthest corner hit is nearest
        double tmax = 1e300;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (hit[c] < 0) continue;
            bool seen = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) seen = seen || (e < c && hit[e] == hit[c]);
            if (seen) continue;
            double far = 0.0; bool all = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!all) continue;
                double t, u, v, det;
                const D3 dq = unit[q];
                all = tri_hit(S.tri_geo, (uint32_t)hit[c], pos, dq, t, u, v, det)
                      && det >= kDetMargin && u >= kBaryMargin && v >= kBaryMargin && 1.0 - u - v >= kBaryMargin && t > 0.0;
                far = fmax(far, t);
            }
            if (all) tmax = fmin(tmax, far * kTmaxMargin);
        }
        // ---- 3. the leaves whose boxes meet the frustum within tmax
        // side planes through pos, normals pointing out (corners in the order (-,-) (+,-) (+,+) (-,+))
        D3 nrm[4]; double nslack[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ca = q == 0 ? 0 : q == 1 ? 1 : q == 2 ? 3 : 2, cb = q == 0 ? 1 : q == 1 ? 3 : q == 2 ? 2 : 0;
            D3 n = cross(dir[ca], dir[cb]);
            if (dot(n, centre) > 0.0) n = n * -1.0;
            nrm[q] = n;
            nslack[q] = 1e-12 * (fabs(n.x) + fabs(n.y) + fabs(n.z)) * (coord + 1.0);     // double rounding of the plane test, generously
        }
        const rtbvh::Node4* nodes = reinterpret_cast<const rtbvh::Node4*>(S.nodes);
        uint32_t f0 = rtk::kNone, f1 = rtk::kNone, f2 = rtk::kNone, f3 = rtk::kNone; int nf = 0; bool too_many = false;
        int sp = 0; stk.push(sp, 0u);
        while (sp > 0 && !too_many) {
            const uint32_t c = stk.pop(sp);
            const rtbvh::Node4& N = nodes[c];
            for (int k = 0; k < 4; ++k) {
                if (N.child[k] == rtbvh::kEmpty) continue;
                const double bx0 = N.minx[k], bx1 = N.maxx[k], by0 = N.miny[k], by1 = N.maxy[k], bz0 = N.minz[k], bz1 = N.maxz[k];
                if (!(bx0 <= bx1 && by0 <= by1 && bz0 <= bz1)) {        // a NaN box (degenerate input): cannot be reasoned about
                    if (bx0 != bx0 || bx1 != bx1 || by0 != by0 || by1 != by1 || bz0 != bz0 || bz1 != bz1) { too_many = true; break; }
                    continue;                                            // (+inf, -inf): an empty slot's box
                }
                bool outside = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                            // the box corner deepest inside the plane's inner side is still outside
                    const D3 pc{ nrm[q].x > 0.0 ? bx0 : bx1, nrm[q].y > 0.0 ? by0 : by1, nrm[q].z > 0.0 ? bz0 : bz1 };
                    outside = outside || dot(nrm[q], pc - pos) > nslack[q];
                }
                if (!outside && tmax < 1e299) {                          // nearest point of the box to the camera farther than tmax
                    const double dx = fmax(fmax(bx0 - pos.x, 0.0), pos.x - bx1), dy = fmax(fmax(by0 - pos.y, 0.0), pos.y - by1), dz = fmax(fmax(bz0 - pos.z, 0.0), pos.z - bz1);
                    outside = dx * dx + dy * dy + dz * dz > tmax * tmax * (1.0 + 1e-8);
                }
                if (outside) continue;
                if (N.child[k] & rtbvh::kLeafBit) {
                    if (nf >= kMaxList) { too_many = true; break; }
                    const uint32_t leaf = N.child[k];
                    if (nf == 0) f0 = leaf; else if (nf == 1) f1 = leaf; else if (nf == 2) f2 = leaf; else f3 = leaf;
                    ++nf;
                } else {
                    if (sp >= A.stack_cap) { too_many = true; break; }
                    stk.push(sp, N.child[k]);
                }
            }
        }
        if (too_many) break;
        out = make_uint4(f0, f1, f2, f3);
        kind = nf == 0 ? 2 : (tmax < 1e299 ? 0 : 1);
    } while (false);
    A.lists[(size_t)ly * A.p.width + x] = out;
    if (A.counts) {                                                     // one atomic per kind and wave
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long m = rtk::ballot_(kind == k);
            if (m != 0ull && (unsigned)__builtin_ctzll(m) == (threadIdx.x & 63u)) atomicAdd(&A.counts[k], (unsigned int)__popcll(m));
        }
    }
}

} // namespace rtp
