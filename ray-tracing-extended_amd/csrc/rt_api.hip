// rt_api.hip — C-ABI of include/rt.h on top of the HIP kernels (gfx950 only, no CPU fallback).
//
// Host-side counterpart of RayTracingManager.OnRenderImage / InitFrame (RayTracingManager.cs:49-124) below
// the Material.Set* / Graphics.Blit boundary: owns the device copies of the three structured buffers, the
// accumulation target (resultTexture) and the per-frame target (currentFrame).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include "rt_kernels.hpp"
#include "rt_stream.hpp"
namespace rtk { const void* stream_kernel(bool counting, bool philox, bool compact, bool triangles); }   // rt_stream_kernels.hip
#include "rt_geom.hpp"
#include "rt_bvh_gpu.hpp"
#include "rt_primary.hpp"

namespace {

thread_local std::string g_create_error;

#define RT_HIP(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(ctx, -100, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T> struct DevBuf {
    T* p = nullptr; size_t cap = 0, used = 0;       // used: elements the last ensure() asked for (what clone_scene copies)
    hipError_t ensure(size_t n)
    {
        used = n;
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = std::max<size_t>(n, 1);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; used = 0; }
};

} // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    rt_params params{};
    bool have_params = false;
    int row0 = 0, nrows = -1;           // -1: whole image
    int band_first = 0, band_stride = 0; // band_stride > 0: interleaved 8-row bands (overrides row0/nrows)

    std::vector<rt_sphere>   h_spheres;
    std::vector<rt_triangle> h_tris;
    std::vector<rt_meshinfo> h_mesh;
    bool scene_dirty = true;
    size_t n_tris = 0, n_chunks = 0, n_spheres = 0;     // of the scene on the device (set by build_scene / build_scene_local / clone_scene)
    float sphere_mag = 0.f;                             // largest |coordinate| of a sphere surface point of that scene
    // on-device geometry pipeline (rt_upload_local_meshes / rt_set_mesh_transforms)
    bool geom_local = false, xf_dirty = false;
    uint64_t frames_traced = 0, frames_at_world_build = ~0ull;     // device_bvh = -1: a world-space scene that changes again soon is one that moves
    std::vector<rt_triangle>       h_local_tris;
    std::vector<rt_local_chunk>    h_lchunks;
    std::vector<rt_mesh_transform> h_xf;
    std::vector<float>             mesh_radius;     // largest |local coordinate| * sqrt(3) per mesh
    int n_meshes = 0;
    DevBuf<float> d_local_tris;
    DevBuf<uint32_t> d_tri_mesh, d_tri_chunk, d_tri_rank, d_order;
    DevBuf<rtg::MeshXf> d_xf;
    hipEvent_t evg0 = nullptr, evg1 = nullptr;

    DevBuf<float4> d_sph_geom, d_sph_mat, d_nodes, d_nodes_h, d_tri_geo, d_tri_nrm, d_chunk_mat, d_chunk_box;
    DevBuf<float>  d_raw_tris;
    DevBuf<uint32_t> d_raw_range;
    DevBuf<float4> d_frame, d_accum;
    DevBuf<uint32_t> d_display;
    DevBuf<float4> d_batch;            // per-frame outputs of a multi-frame launch
    DevBuf<uint32_t> d_tile_order, d_tile_cost, d_tile_hist;
    bool lpt_active = true;             // the last launch used (or could have used) a costliest-first order: the automatic kernel choice waits for it
    bool tile_order_valid = false, tile_order_stale = false; int tile_order_n = 0;   // stale: usable, re-measured by the next launch
    size_t target_pixels = 0;
    int target_w = 0, target_h = 0, target_row0 = 0, target_rows = 0, target_row_stride = 8;
    unsigned int* d_tile_counter = nullptr;
    unsigned long long* d_counters = nullptr;

    rtbvh::Bvh bvh;                 // host builder's result (nodes stay empty after a device build; order / levels / depth are shared)
    size_t n_nodes = 0;             // BVH4 nodes on the device
    rtgb::Workspace bvh_ws;         // device builder's scratch
    float area_at_build = 0.f;      // sum of internal child-box areas right after the last build (refit quality monitor)
    float area_after_refit = 0.f;   // ... after the last refit: written by an asynchronous copy, so it lives in the context, not on a stack
    int opt_device_bvh = -1;        // 1: build the BVH on the device (Morton order + PLOC + collapse), 0: host binned-SAH builder,
                                    // -1: device for the on-device geometry pipeline (meshes that move), host for world-space uploads
                                    // (a static scene is built once and traced for many frames: the SAH tree is traced 4-16 % faster,
                                    // the device build is 25-140x faster)
    int opt_bvh_radius = -16;       // device builder: PLOC search radius; positive: of the first rounds, doubled once a quarter and again once a
                                    // sixteenth of the clusters is left; negative = that radius in every round (default: 16 throughout — with the
                                    // treelet passes on top, profiles/sweep_devtree_r04.txt: wave-level node steps per ray 10.69 / 13.94 on the
                                    // 100k / 1M workloads against 11.16 / 14.52 for 8 widening and 10.64 / 14.91 for the host's SAH tree)
    int opt_peer_copies = 0;        // 1: rt_multi's device-to-device copies take the peer API (hipMemcpyPeerAsync) even between contexts of ONE device
                                    // — the branch a multi-GPU node takes, runnable on a one-GPU box (tests)
    int opt_bvh_treelet_ratio = 8, opt_bvh_treelet_isolate = 1, opt_bvh_treelet_first = 1;      // (tuning of the passes: scale step, large-box isolation, first item size)
    int opt_bvh_treelets = 6;       // device builder: sweep-SAH passes over the clustering's tree, one wave per treelet of <= 64 items (rt_bvh_gpu.hpp step 3c)
    int opt_bvh_top = 0;            // device builder: > 0 = once the bottom-up rounds have left at most this many clusters, the top of the tree is built by
                                    // the host's binned-SAH split search over their boxes (round 3's default 1024: a few hundred KB and about a
                                    // millisecond of host time); 0 (default since round 4) = everything on the device: the treelet passes reach the root
    int opt_rebuild_percent = 200;  // device pipeline: rebuild instead of refit once the internal area exceeds this share of the build's
    int n_cu = 0;
    int opt_kernel = -1;            // -1: auto (k_trace or k_stream, measured per scene), 0: k_trace, 1: k_stream
    int auto_choice = -1; double auto_ms[2] = { -1.0, -1.0 };
    unsigned variants_launched = 0;     // kernel variants that have run at least once in this context (automatic choice: see launch_frames)
    int opt_shade_threshold = 48;
    int opt_tile_sync = 1;
    int opt_fetch_guide = 4;        // k_stream: groups of tiles_per_fetch items while more than this many groups per wave are left (then smaller)
    int opt_fetch_guide_philox = 1; // ... the same in Philox mode
    int opt_tiles_per_fetch = 16;   // k_stream: items a wave reserves per fetch while the queue is long (guided: fewer near the end).  Fixed groups of
                                    // 2 / 4 / 8: 11.89 / 12.17 / 11.65 Grays/s (the tail grows); guided 4 / 8 / 12 / 16 / 24: 12.35 / 12.55 / 12.60 / 12.61 / 12.60.
                                    // 16 = the sub-tiles of one 8x8 tile in a 2x2x16 launch: groups stay tile-aligned (15.31 against 15.26 at 12)
    int opt_compact_nodes = 1;      // k_trace / k_stream: traverse the f16 form of the nodes (Node4h: 5 loads per visit instead of 7)
    int opt_stream_tile = 4;        // k_stream: log2 of the most frames interleaved in a wave (0: 8x8 pixels x 1 frame, 2: 4x4 x 4, 4: 2x2 x 16)
                                    // measured on the 100k-triangle workload: 10.86 / 11.48 / 11.89 Grays/s; with 4 tiles per fetch 12.17
    int opt_node_min = 10;          // k_stream: 4..10 within 0.5 % of each other on the 100k-triangle workload (+7 % over 1); 6 / 8 / 10 on the
                                    // million-triangle one: 11.72 / 11.83 / 11.90 Grays/s
    int opt_blocks_per_cu = 0;      // 0: occupancy API
    int opt_full_sort = 0;          // 1: sort all four children; 0: nearest first only (measured +1 %)
    int opt_tile_lpt = 1;           // k_trace: dispatch the costliest tiles first, using the costs measured by the previous launch
    int opt_frame_batch = 0;        // k_trace: frames per launch in rt_render (0 = auto, 1 = one launch per frame)
    int opt_tile_w_log2 = 3;        // k_trace: tile width 2^n (n = 3: 8x8 tiles)
    int opt_bvh_reinsert = 0;       // BVH builder: insertion-based optimisation passes
    int opt_bvh_bins = 32, opt_bvh_cost_exp = 100;   // BVH builder: SAH bins per axis; exponent (percent) of the count in the SAH cost model
    int opt_max_leaf = 2;           // BVH: triangles per leaf (measured best on the 100k-triangle workload: 2)
    int opt_bvh_collapse = 0;       // host builder: 0 = greedy collapse of the binary tree to 4-wide nodes (open the largest child), 1 / 2 = cost-driven (bvh.cpp;
                                    // measured -4.4 % / -1.5 % on the headline scene, -1.4 % / +1.5 % on the million-triangle one: profiles/bvh_collapse_r04.txt)
    int opt_bvh_node_cost = 130;    // ... with a node step costing this many percent of a triangle test
    int opt_stream_stack = 0;       // k_stream: stack entries per lane kept in LDS; deeper BVHs spill the rest to global memory.  0 = as many as let the
                                    // instantiation's waves per SIMD be resident: 24 entries + the groups' item tables = 25,600 B per workgroup for the six-wave
                                    // PCG / f16-node kernel (six workgroups per CU; 26 entries make it five: -8 %), 30 = 31,744 B for the five-wave ones
    int opt_lds_stack = 0;          // k_trace: stack entries per lane kept in LDS (0 = the BVH's worst case, nothing spills)
    DevBuf<uint32_t> d_gstack;
    // camera rays' candidate lists (rt_primary.hpp): valid for one (params, rows, scene) combination
    DevBuf<uint4> d_primary; DevBuf<float4> d_focus; DevBuf<unsigned int> d_primary_counts;
    std::string primary_key;                        // what the lists were built for (parameters, rows, scene version)
    unsigned long long scene_version = 0;           // bumped whenever the tree or its boxes change (build, refit, re-padding)
    int opt_primary_lists = 1;                      // 1: camera rays of a static camera start from their pixel's candidate leaves (k_stream); 0: always from the root
    DevBuf<float> d_park;              // k_stream, Philox mode: parked sub-stream sums
    rt_stats stats{};

    // ---- queued submission (rt_submit_frame / rt_wait): frames handed in one by one — the reference's OnRenderImage pattern,
    // RayTracingManager.cs:74-91 — are traced by a worker thread in launches of whatever has queued up while the previous launch ran
    std::thread q_thread; bool q_started = false;
    std::mutex q_mu; std::condition_variable q_cv, q_idle;
    std::deque<int> q_frames; bool q_busy = false, q_stop = false;
    int q_rc = 0; std::string q_err;
    int opt_queue_depth = 64;       // most frames the worker puts into one launch
    int opt_queue_linger_us = 200;  // after the first frame of an idle queue arrives the worker waits this long for more (a host that submits a burst
                                    // of frames gets one launch for it; a host that submits one frame per display refresh pays 0.2 ms)
};

namespace {

int fail(rt_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

inline double now_ms()
{
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
inline float4 f4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

void pack_material(const rt_material& m, float4* out)
{
    out[0] = f4(m.colour); out[1] = f4(m.emissionColour); out[2] = f4(m.specularColour);
    out[3] = make_float4(m.emissionStrength, m.smoothness, m.specularProbability, u2f((uint32_t)m.flag));
}

// True when every camera ray starts exactly at worldSpaceCameraPos (frag :377-378 with defocusStrength = +-0): the jitter is
// (finite * +-0) / width = +-0, the basis vectors times it are +-0 when they are finite, and pos + (+-0) = pos bit for bit unless a
// component of pos is -0 (which a +0 turns into +0) or not finite.  The kernels then skip that arithmetic (camera_ray).
bool camera_origin_is_fixed(const rt_params& p)
{
    if (!(p.defocusStrength == 0.0f) || p.width < 1) return false;
    const int basis[6] = { 0, 4, 8, 1, 5, 9 };                   // camera right and up (columns 0 and 1 of camLocalToWorld)
    for (int k : basis) if (!std::isfinite(p.camLocalToWorld[k])) return false;
    for (int a = 0; a < 3; ++a) {
        const float v = p.worldSpaceCameraPos[a];
        if (!std::isfinite(v) || (v == 0.0f && std::signbit(v))) return false;
    }
    return true;
}

// Largest |coordinate| a camera-ray origin can have: camera position plus the defocus disc (frag :377-378).
float camera_magnitude(const rt_params& p)
{
    float m = 0.f;
    for (int a = 0; a < 3; ++a) m = std::max(m, std::fabs(p.worldSpaceCameraPos[a]));
    float jitter = std::fabs(p.defocusStrength) / std::max(1.0f, (float)p.width);
    float axis = 0.f;
    for (int a = 0; a < 3; ++a)
        axis = std::max(axis, std::fabs(p.camLocalToWorld[4 * a]) + std::fabs(p.camLocalToWorld[4 * a + 1]));
    return m + jitter * axis;
}

// Node4 -> Node4h on the device (after a build's upload and after every refit)
int compact_nodes(rt_ctx* c)
{
    const uint32_t nn = (uint32_t)c->n_nodes;
    RT_HIP(c, c->d_nodes_h.ensure((size_t)nn * 8));
    if (nn) {
        hipLaunchKernelGGL(rtg::k_compact_nodes, dim3((nn + 255) / 256), dim3(256), 0, c->stream,
                           reinterpret_cast<const rtbvh::Node4*>(c->d_nodes.p), reinterpret_cast<rtbvh::Node4h*>(c->d_nodes_h.p), nn);
        RT_HIP(c, hipGetLastError());
    }
    return 0;
}

// Bounce rays start on sphere surfaces too (Trace :327): the largest |coordinate| of any sphere surface point.
float sphere_magnitude(const rt_ctx* c)
{
    float m = 0.f;
    for (const rt_sphere& s : c->h_spheres) {
        const float r = std::fabs(s.radius);
        for (int a = 0; a < 3; ++a) { const float v = std::fabs(s.position[a]) + r; if (v > m && v < 3.0e38f) m = v; }
    }
    return m;
}

rtbvh::Tuning bvh_tuning(const rt_ctx* c)
{
    rtbvh::Tuning t;
    t.bins = c->opt_bvh_bins; t.cost_exp_percent = c->opt_bvh_cost_exp; t.reinsert_passes = c->opt_bvh_reinsert; t.max_leaf = c->opt_max_leaf;
    t.collapse_dp = c->opt_bvh_collapse; t.node_cost_percent = c->opt_bvh_node_cost;
    return t;
}

// BVH over the world-space triangles in d_raw_tris, built on the device; then the tracer's triangle records and the f16 nodes.
// Needs d_tri_chunk (0xFFFFFFFF = in no chunk: never hit) and d_tri_rank uploaded.
int device_build(rt_ctx* c, uint32_t nt, float origin_magnitude)
{
    RT_HIP(c, c->d_nodes.ensure(((size_t)nt + 1) * 8)); RT_HIP(c, c->d_order.ensure(nt));
    RT_HIP(c, c->d_tri_geo.ensure(3 * (size_t)nt)); RT_HIP(c, c->d_tri_nrm.ensure(3 * (size_t)nt));
    RT_HIP(c, hipEventRecord(c->evg0, c->stream));
    rtgb::Result res;
    RT_HIP(c, rtgb::build(c->stream, c->d_raw_tris.p, nt, origin_magnitude, c->opt_bvh_radius, c->bvh_ws, reinterpret_cast<rtbvh::Node4*>(c->d_nodes.p), c->d_order.p, res,
                          (uint32_t)c->opt_bvh_top, bvh_tuning(c), c->opt_bvh_treelets, c->opt_bvh_treelet_ratio, c->opt_bvh_treelet_isolate, c->opt_bvh_treelet_first));
    c->bvh.nodes.clear(); c->bvh.order.clear();
    c->n_nodes = res.n_nodes; c->bvh.levelStart = res.level_start; c->bvh.maxStack = res.max_stack; c->bvh.magnitude = res.magnitude;
    c->bvh.depth = res.levels;
    if (nt) {
        hipLaunchKernelGGL(rtg::k_relayout, dim3((nt + 255) / 256), dim3(256), 0, c->stream,
                           c->d_raw_tris.p, c->d_order.p, c->d_tri_chunk.p, c->d_tri_rank.p, c->d_tri_geo.p, c->d_tri_nrm.p, nt);
        RT_HIP(c, hipGetLastError());
    }
    { int r = compact_nodes(c); if (r) return r; }
    RT_HIP(c, hipEventRecord(c->evg1, c->stream));
    RT_HIP(c, rtgb::internal_area(c->stream, reinterpret_cast<const rtbvh::Node4*>(c->d_nodes.p), (uint32_t)c->n_nodes, c->bvh_ws, c->area_at_build));
    float ms = 0.f;
    RT_HIP(c, hipEventElapsedTime(&ms, c->evg0, c->evg1));
    c->stats.lastBvhBuildMs = ms; c->stats.bvhBuiltOnDevice = 1; c->stats.bvhBuilds++; c->scene_version++;
    return 0;
}

// Re-layout of the uploaded buffers + BVH build.  Edge vectors and their cross product are the operands of
// RayTriangle (RayTracing.shader:152-154) evaluated once here with the same float operations.
int build_scene(rt_ctx* c)
{
    const size_t ns = c->h_spheres.size(), nt = c->h_tris.size(), nm = c->h_mesh.size();
    if (nt > (1u << 28)) return fail(c, -3, "too many triangles (%zu)", nt);
    // triangle -> chunk map; every triangle the shader can reach belongs to exactly the chunk ranges given
    std::vector<uint32_t> chunk_of(nt, 0xFFFFFFFFu);
    for (size_t m = 0; m < nm; ++m) {
        const rt_meshinfo& mi = c->h_mesh[m];
        if ((uint64_t)mi.firstTriangleIndex + mi.numTriangles > nt)
            return fail(c, -4, "meshinfo[%zu] addresses triangles [%u,%u) beyond the %zu uploaded", m,
                        mi.firstTriangleIndex, mi.firstTriangleIndex + mi.numTriangles, nt);
        for (uint32_t i = 0; i < mi.numTriangles; ++i) {
            uint32_t& slot = chunk_of[mi.firstTriangleIndex + i];
            if (slot != 0xFFFFFFFFu)
                return fail(c, -5, "triangle %u is referenced by chunks %u and %zu (overlapping chunk ranges are not supported)",
                            mi.firstTriangleIndex + i, slot, m);
            slot = (uint32_t)m;
        }
    }
    // equal-distance hits: the reference keeps the triangle its loops reach first — chunks in AllMeshInfo order, triangles in
    // order inside the chunk (CalculateRayCollision :276-293).  That visiting rank, not the buffer index, is the tie-break key.
    std::vector<uint32_t> visit_rank(nt, 0xFFFFFFFFu);
    {
        uint32_t r = 0;
        for (size_t m = 0; m < nm; ++m)
            for (uint32_t i = 0; i < c->h_mesh[m].numTriangles; ++i) visit_rank[c->h_mesh[m].firstTriangleIndex + i] = r++;
    }
    // triangles outside every chunk are never visited by the shader: leave them out of the hierarchy
    std::vector<uint32_t> live; live.reserve(nt);
    for (size_t t = 0; t < nt; ++t) if (chunk_of[t] != 0xFFFFFFFFu) live.push_back((uint32_t)t);
    std::vector<float4> sg(ns), sm(4 * ns), cm(4 * nm), cb(2 * nm);
    for (size_t i = 0; i < ns; ++i) {
        const rt_sphere& s = c->h_spheres[i];
        sg[i] = make_float4(s.position[0], s.position[1], s.position[2], s.radius);
        pack_material(s.material, &sm[4 * i]);
    }
    std::vector<uint32_t> range(2 * nm);
    for (size_t m = 0; m < nm; ++m) {
        const rt_meshinfo& mi = c->h_mesh[m];
        pack_material(mi.material, &cm[4 * m]);
        cb[2 * m]     = make_float4(mi.boundsMin[0], mi.boundsMin[1], mi.boundsMin[2], 0.f);
        cb[2 * m + 1] = make_float4(mi.boundsMax[0], mi.boundsMax[1], mi.boundsMax[2], 0.f);
        range[2 * m] = mi.firstTriangleIndex; range[2 * m + 1] = mi.numTriangles;
    }
#define RT_UP(buf, vec, T)                                                                                  \
    RT_HIP(c, buf.ensure(vec.size()));                                                                      \
    if (!vec.empty()) RT_HIP(c, hipMemcpyAsync(buf.p, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    RT_UP(c->d_sph_geom, sg, float4) RT_UP(c->d_sph_mat, sm, float4)
    RT_UP(c->d_chunk_mat, cm, float4) RT_UP(c->d_chunk_box, cb, float4)
    RT_UP(c->d_raw_range, range, uint32_t)
    RT_HIP(c, c->d_raw_tris.ensure(nt * 18));
    if (nt) RT_HIP(c, hipMemcpyAsync(c->d_raw_tris.p, c->h_tris.data(), nt * sizeof(rt_triangle), hipMemcpyHostToDevice, c->stream));
    // headroom: the boxes are padded for ray origins up to twice as far out as the camera and the spheres are now (the term is
    // 2e-6 * G), so a camera that drifts away from the geometry widens the padding (repad_boxes) once per doubling, not per frame
    const float origin_mag = 2.0f * std::max(camera_magnitude(c->params), sphere_magnitude(c));
    c->stats.bvhBuiltOnDevice = 0;
    // device_bvh = -1: the first build of a world-space scene is the host's (a static scene is built once); a scene that is uploaded
    // again within 16 traced frames of its last build is being animated the reference's way — the whole scene re-sent every frame,
    // RayTracedMesh.cs:36-84 — and a 50-600 ms host build per frame would dwarf the trace: those builds go to the device (2.5-5.4 ms)
    const bool moving = c->opt_device_bvh == -1 && c->frames_at_world_build != ~0ull && c->frames_traced - c->frames_at_world_build <= 16;
    c->frames_at_world_build = c->frames_traced;
    if ((c->opt_device_bvh == 1 || moving) && nt > 0) {
        // the device builder takes every uploaded triangle; one that belongs to no chunk gets NaN records (k_relayout) and can never be hit
        RT_UP(c->d_tri_chunk, chunk_of, uint32_t) RT_UP(c->d_tri_rank, visit_rank, uint32_t)
        { int r = device_build(c, (uint32_t)nt, origin_mag); if (r) return r; }
    } else {
        std::vector<float> pos(9 * live.size());
        for (size_t i = 0; i < live.size(); ++i) std::memcpy(&pos[9 * i], c->h_tris[live[i]].posA, 36);
        const double t0 = now_ms();
        rtbvh::build(pos.data(), 9, (uint32_t)live.size(), origin_mag, bvh_tuning(c), c->bvh);
        c->stats.lastBvhBuildMs = now_ms() - t0; c->stats.bvhBuilds++;
        c->n_nodes = c->bvh.nodes.size();
        if (live.empty()) c->bvh.magnitude = origin_mag;        // no tree: only the trigger of repad_boxes looks at it
        const size_t nl = live.size();
        std::vector<float4> geo(3 * nl), nrm(3 * nl);
        for (size_t i = 0; i < nl; ++i) {
            const uint32_t orig = live[c->bvh.order[i]];
            const rt_triangle& t = c->h_tris[orig];
            const float ex = t.posB[0] - t.posA[0], ey = t.posB[1] - t.posA[1], ez = t.posB[2] - t.posA[2];
            const float fx = t.posC[0] - t.posA[0], fy = t.posC[1] - t.posA[1], fz = t.posC[2] - t.posA[2];
            const float nx = ey * fz - ez * fy, ny = ez * fx - ex * fz, nz = ex * fy - ey * fx;
            geo[3 * i + 0] = make_float4(t.posA[0], t.posA[1], t.posA[2], ex);
            geo[3 * i + 1] = make_float4(ey, ez, fx, fy);
            geo[3 * i + 2] = make_float4(fz, nx, ny, nz);
            nrm[3 * i + 0] = make_float4(t.normalA[0], t.normalA[1], t.normalA[2], u2f(chunk_of[orig]));
            nrm[3 * i + 1] = make_float4(t.normalB[0], t.normalB[1], t.normalB[2], u2f(visit_rank[orig]));
            nrm[3 * i + 2] = make_float4(t.normalC[0], t.normalC[1], t.normalC[2], 0.f);
        }
        RT_UP(c->d_tri_geo, geo, float4) RT_UP(c->d_tri_nrm, nrm, float4)
        // BVH order -> uploaded triangle, on the device: what a later widening of the box padding (repad_boxes) refits from
        std::vector<uint32_t> order_raw(nl);
        for (size_t i = 0; i < nl; ++i) order_raw[i] = live[c->bvh.order[i]];
        RT_UP(c->d_order, order_raw, uint32_t)
        RT_HIP(c, c->d_nodes.ensure(c->bvh.nodes.size() * 8));
        if (!c->bvh.nodes.empty())
            RT_HIP(c, hipMemcpyAsync(c->d_nodes.p, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(rtbvh::Node4),
                                     hipMemcpyHostToDevice, c->stream));
        { int r = compact_nodes(c); if (r) return r; }
        RT_HIP(c, hipStreamSynchronize(c->stream));     // host staging vectors die here
        RT_HIP(c, rtgb::internal_area(c->stream, reinterpret_cast<const rtbvh::Node4*>(c->d_nodes.p), (uint32_t)c->n_nodes, c->bvh_ws, c->area_at_build));
    }
#undef RT_UP
    RT_HIP(c, hipStreamSynchronize(c->stream));     // host staging vectors die here

    c->stats.numSpheres = (int)ns; c->stats.numTriangles = (int)nt; c->stats.numMeshChunks = (int)nm;
    c->stats.numBvhNodes = (int)c->n_nodes; c->stats.bvhMaxStack = c->bvh.maxStack; c->stats.bvhInternalArea = c->area_at_build;
    c->n_spheres = ns; c->n_tris = nt; c->n_chunks = nm; c->sphere_mag = sphere_magnitude(c);
    c->scene_dirty = false; c->tile_order_valid = false; c->scene_version++;
    return 0;
}

// World-space uploads: a ray origin (camera, defocus disc, sphere surface) has moved beyond the magnitude G the boxes were padded
// for (pad = 3e-5 |coord| + 2e-6 G, bvh.cpp pad_box).  The tree stays: its boxes are refitted bottom-up from the uploaded triangles
// with the new G (k_refit_level writes the same padded leaf boxes a build would) and the f16 form is derived again — a few small
// launches instead of a host rebuild and a re-upload of every record.
int repad_boxes(rt_ctx* c, float G)
{
    if (c->n_nodes && c->bvh.levelStart.size() >= 2) {
        for (int L = (int)c->bvh.levelStart.size() - 2; L >= 0; --L) {
            const uint32_t n0 = c->bvh.levelStart[L], n1 = c->bvh.levelStart[L + 1];
            if (n1 > n0)
                hipLaunchKernelGGL(rtg::k_refit_level, dim3(((n1 - n0) * 4 + 255) / 256), dim3(256), 0, c->stream,
                                   reinterpret_cast<rtbvh::Node4*>(c->d_nodes.p), n0, n1, c->d_raw_tris.p, c->d_order.p, G);
        }
        RT_HIP(c, hipGetLastError());
        { int r = compact_nodes(c); if (r) return r; }
    }
    c->bvh.magnitude = G;
    c->stats.bvhRepads++; c->scene_version++;
    return 0;
}

// ---- on-device geometry pipeline -------------------------------------------------------------------------------
// Conservative bound of |coordinate| over camera-ray origins and the transformed meshes (box padding, bvh.cpp pad_box).
float local_scene_magnitude(const rt_ctx* c)
{
    float G = std::max(camera_magnitude(c->params), sphere_magnitude(c));
    for (int m = 0; m < c->n_meshes; ++m) {
        const rt_mesh_transform& t = c->h_xf[m];
        float p = std::max(std::fabs(t.position[0]), std::max(std::fabs(t.position[1]), std::fabs(t.position[2])));
        float sc = std::max(std::fabs(t.lossyScale[0]), std::max(std::fabs(t.lossyScale[1]), std::fabs(t.lossyScale[2])));
        G = std::max(G, p + 1.01f * sc * c->mesh_radius[m]);
    }
    return G;
}

int upload_spheres_and_materials(rt_ctx* c)
{
    const size_t ns = c->h_spheres.size();
    std::vector<float4> sg(ns), sm(4 * ns);
    for (size_t i = 0; i < ns; ++i) {
        const rt_sphere& s = c->h_spheres[i];
        sg[i] = make_float4(s.position[0], s.position[1], s.position[2], s.radius);
        pack_material(s.material, &sm[4 * i]);
    }
    RT_HIP(c, c->d_sph_geom.ensure(ns)); RT_HIP(c, c->d_sph_mat.ensure(4 * ns));
    if (ns) {
        RT_HIP(c, hipMemcpyAsync(c->d_sph_geom.p, sg.data(), ns * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync(c->d_sph_mat.p, sm.data(), 4 * ns * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->stats.numSpheres = (int)ns;
    return 0;
}

// transform + chunk bounds (+ re-layout and refit when the BVH topology already exists)
int run_geometry_kernels(rt_ctx* c, bool have_bvh)
{
    const uint32_t nt = (uint32_t)c->h_local_tris.size(), nm = (uint32_t)c->h_lchunks.size();
    std::vector<rtg::MeshXf> xf(c->n_meshes);
    for (int m = 0; m < c->n_meshes; ++m) {
        const rt_mesh_transform& t = c->h_xf[m];
        xf[m] = { t.position[0], t.position[1], t.position[2], t.rotation[0], t.rotation[1], t.rotation[2], t.rotation[3],
                  t.lossyScale[0], t.lossyScale[1], t.lossyScale[2] };
    }
    RT_HIP(c, c->d_xf.ensure(xf.size()));
    if (!xf.empty()) RT_HIP(c, hipMemcpyAsync(c->d_xf.p, xf.data(), xf.size() * sizeof(rtg::MeshXf), hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipEventRecord(c->evg0, c->stream));
    if (nt) {
        hipLaunchKernelGGL(rtg::k_transform, dim3((nt + 255) / 256), dim3(256), 0, c->stream,
                           c->d_local_tris.p, c->d_tri_mesh.p, c->d_xf.p, c->d_raw_tris.p, nt);
        hipLaunchKernelGGL(rtg::k_chunk_bounds, dim3((nm + 255) / 256), dim3(256), 0, c->stream,
                           c->d_raw_tris.p, c->d_raw_range.p, c->d_chunk_box.p, nm);
    }
    if (have_bvh && nt) {
        hipLaunchKernelGGL(rtg::k_relayout, dim3((nt + 255) / 256), dim3(256), 0, c->stream,
                           c->d_raw_tris.p, c->d_order.p, c->d_tri_chunk.p, c->d_tri_rank.p, c->d_tri_geo.p, c->d_tri_nrm.p, nt);
        const float G = std::max(c->bvh.magnitude, local_scene_magnitude(c));
        c->bvh.magnitude = G;
        for (int L = (int)c->bvh.levelStart.size() - 2; L >= 0; --L) {
            const uint32_t n0 = c->bvh.levelStart[L], n1 = c->bvh.levelStart[L + 1];
            if (n1 > n0)
                hipLaunchKernelGGL(rtg::k_refit_level, dim3(((n1 - n0) * 4 + 255) / 256), dim3(256), 0, c->stream,
                                   reinterpret_cast<rtbvh::Node4*>(c->d_nodes.p), n0, n1, c->d_raw_tris.p, c->d_order.p, G);
        }
        { int r = compact_nodes(c); if (r) return r; }
    }
    RT_HIP(c, hipGetLastError());
    // a refit keeps the topology: meshes that moved apart leave boxes that overlap more and more.  The refitted tree's internal area is
    // summed in the same pass (one more small kernel and a 4-byte copy before the one synchronisation, inside lastGeometryMs); once it
    // has grown past the threshold the tree is rebuilt on the device (cheaper than one frame) instead of refitted.
    const bool check_area = have_bvh && nt && c->opt_device_bvh != 0 && c->opt_rebuild_percent > 0 && c->area_at_build > 0.f;
    if (check_area) RT_HIP(c, rtgb::internal_area_async(c->stream, reinterpret_cast<const rtbvh::Node4*>(c->d_nodes.p), (uint32_t)c->n_nodes, c->bvh_ws, &c->area_after_refit));
    RT_HIP(c, hipEventRecord(c->evg1, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    RT_HIP(c, hipEventElapsedTime(&ms, c->evg0, c->evg1));
    c->stats.lastGeometryMs = ms; c->scene_version++;
    if (check_area) {
        const float area = c->area_after_refit;
        c->stats.refitAreaRatio = area / c->area_at_build;
        if (area > c->area_at_build * (float)c->opt_rebuild_percent / 100.0f) {
            int r = device_build(c, nt, local_scene_magnitude(c)); if (r) return r;
            c->stats.numBvhNodes = (int)c->n_nodes; c->stats.bvhMaxStack = c->bvh.maxStack;
            c->stats.bvhRebuilds++; c->tile_order_stale = true;
        }
    }
    return 0;
}

int build_scene_local(rt_ctx* c)
{
    const size_t nt = c->h_local_tris.size(), nm = c->h_lchunks.size();
    if ((int)c->h_xf.size() != c->n_meshes) return fail(c, -2, "rt_set_mesh_transforms has not been called for the %d meshes", c->n_meshes);
    { int r = upload_spheres_and_materials(c); if (r) return r; }
    std::vector<uint32_t> tri_mesh(nt), tri_chunk(nt), tri_rank(nt), range(2 * nm);
    std::vector<float4> cm(4 * nm);
    uint32_t rank = 0;
    for (size_t m = 0; m < nm; ++m) {
        const rt_local_chunk& ch = c->h_lchunks[m];
        for (uint32_t i = 0; i < ch.numTriangles; ++i) {
            tri_mesh[ch.firstTriangleIndex + i] = ch.meshIndex; tri_chunk[ch.firstTriangleIndex + i] = (uint32_t)m;
            tri_rank[ch.firstTriangleIndex + i] = rank++;         // the reference's visiting order (chunk list order, then in-chunk order)
        }
        range[2 * m] = ch.firstTriangleIndex; range[2 * m + 1] = ch.numTriangles;
        pack_material(ch.material, &cm[4 * m]);
    }
#define RT_UP(buf, vec, T)                                                                                  \
    RT_HIP(c, buf.ensure(vec.size()));                                                                      \
    if (!vec.empty()) RT_HIP(c, hipMemcpyAsync(buf.p, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    RT_UP(c->d_tri_mesh, tri_mesh, uint32_t) RT_UP(c->d_tri_chunk, tri_chunk, uint32_t) RT_UP(c->d_tri_rank, tri_rank, uint32_t)
    RT_UP(c->d_raw_range, range, uint32_t) RT_UP(c->d_chunk_mat, cm, float4)
#undef RT_UP
    RT_HIP(c, c->d_local_tris.ensure(nt * 18)); RT_HIP(c, c->d_raw_tris.ensure(nt * 18));
    RT_HIP(c, c->d_chunk_box.ensure(2 * nm)); RT_HIP(c, c->d_tri_geo.ensure(3 * nt)); RT_HIP(c, c->d_tri_nrm.ensure(3 * nt));
    if (nt) RT_HIP(c, hipMemcpyAsync(c->d_local_tris.p, c->h_local_tris.data(), nt * sizeof(rt_triangle), hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    { int r = run_geometry_kernels(c, false); if (r) return r; }
    c->stats.bvhBuiltOnDevice = 0;
    if (c->opt_device_bvh != 0 && nt > 0) {
        // topology on the device from the world triangles the transform kernel just wrote: nothing goes back to the host
        { int r = device_build(c, (uint32_t)nt, local_scene_magnitude(c)); if (r) return r; }
    } else {
        // world positions (device) -> host, for the one-off topology build
        std::vector<rt_triangle> world(nt);
        if (nt) RT_HIP(c, hipMemcpy(world.data(), c->d_raw_tris.p, nt * sizeof(rt_triangle), hipMemcpyDeviceToHost));
        const double t0 = now_ms();
        rtbvh::build(nt ? world[0].posA : nullptr, 18, (uint32_t)nt, local_scene_magnitude(c), bvh_tuning(c), c->bvh);
        c->stats.lastBvhBuildMs = now_ms() - t0; c->stats.bvhBuilds++;
        c->n_nodes = c->bvh.nodes.size();
        RT_HIP(c, c->d_nodes.ensure(c->bvh.nodes.size() * 8)); RT_HIP(c, c->d_order.ensure(nt));
        if (!c->bvh.nodes.empty()) {
            RT_HIP(c, hipMemcpyAsync(c->d_nodes.p, c->bvh.nodes.data(), c->bvh.nodes.size() * sizeof(rtbvh::Node4), hipMemcpyHostToDevice, c->stream));
            RT_HIP(c, hipMemcpyAsync(c->d_order.p, c->bvh.order.data(), nt * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            { int r = compact_nodes(c); if (r) return r; }
            hipLaunchKernelGGL(rtg::k_relayout, dim3(((uint32_t)nt + 255) / 256), dim3(256), 0, c->stream,
                               c->d_raw_tris.p, c->d_order.p, c->d_tri_chunk.p, c->d_tri_rank.p, c->d_tri_geo.p, c->d_tri_nrm.p, (uint32_t)nt);
            RT_HIP(c, hipGetLastError());
        }
        RT_HIP(c, rtgb::internal_area(c->stream, reinterpret_cast<const rtbvh::Node4*>(c->d_nodes.p), (uint32_t)c->n_nodes, c->bvh_ws, c->area_at_build));
    }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->stats.numTriangles = (int)nt; c->stats.numMeshChunks = (int)nm;
    c->stats.numBvhNodes = (int)c->n_nodes; c->stats.bvhMaxStack = c->bvh.maxStack; c->stats.bvhInternalArea = c->area_at_build;
    c->n_spheres = c->h_spheres.size(); c->n_tris = nt; c->n_chunks = nm; c->sphere_mag = sphere_magnitude(c);
    c->scene_dirty = false; c->xf_dirty = false; c->tile_order_valid = false; c->scene_version++;
    return 0;
}

// Give `dst` the scene `src` has built (world-space uploads): every device buffer is copied device to device — over xGMI when the
// contexts sit on different GPUs — and the host-side description of the tree comes along; dst builds nothing and holds no host copy of
// the triangles.  rt_multi uses it so that N contexts cost one BVH build and one host -> device upload per scene change.
template <class T> int clone_buf(rt_ctx* dst, DevBuf<T>& d, const rt_ctx* src, const DevBuf<T>& s)
{
    RT_HIP(dst, d.ensure(s.used));
    if (!s.used) return 0;
    if (dst->device == src->device && !dst->opt_peer_copies) RT_HIP(dst, hipMemcpyAsync(d.p, s.p, s.used * sizeof(T), hipMemcpyDeviceToDevice, dst->stream));
    else RT_HIP(dst, hipMemcpyPeerAsync(d.p, dst->device, s.p, src->device, s.used * sizeof(T), dst->stream));
    return 0;
}
int clone_scene(rt_ctx* dst, rt_ctx* src)
{
    if (src->scene_dirty || src->geom_local) return fail(dst, -2, "clone_scene: the source context has no built world-space scene");
    RT_HIP(dst, hipSetDevice(src->device));
    RT_HIP(dst, hipStreamSynchronize(src->stream));
    RT_HIP(dst, hipSetDevice(dst->device));
#define RT_CLONE(B) { int r_ = clone_buf(dst, dst->B, src, src->B); if (r_) return r_; }
    RT_CLONE(d_sph_geom) RT_CLONE(d_sph_mat) RT_CLONE(d_nodes) RT_CLONE(d_nodes_h) RT_CLONE(d_tri_geo) RT_CLONE(d_tri_nrm)
    RT_CLONE(d_chunk_mat) RT_CLONE(d_chunk_box) RT_CLONE(d_raw_tris) RT_CLONE(d_raw_range) RT_CLONE(d_order)
#undef RT_CLONE
    RT_HIP(dst, hipStreamSynchronize(dst->stream));
    dst->h_spheres.clear(); dst->h_tris.clear(); dst->h_mesh.clear(); dst->geom_local = false;
    dst->bvh.nodes.clear(); dst->bvh.order.clear();
    dst->bvh.levelStart = src->bvh.levelStart; dst->bvh.maxStack = src->bvh.maxStack; dst->bvh.depth = src->bvh.depth; dst->bvh.magnitude = src->bvh.magnitude;
    dst->n_nodes = src->n_nodes; dst->area_at_build = src->area_at_build;
    dst->n_spheres = src->n_spheres; dst->n_tris = src->n_tris; dst->n_chunks = src->n_chunks; dst->sphere_mag = src->sphere_mag;
    dst->stats.numSpheres = src->stats.numSpheres; dst->stats.numTriangles = src->stats.numTriangles; dst->stats.numMeshChunks = src->stats.numMeshChunks;
    dst->stats.numBvhNodes = src->stats.numBvhNodes; dst->stats.bvhMaxStack = src->stats.bvhMaxStack; dst->stats.bvhInternalArea = src->stats.bvhInternalArea;
    dst->stats.bvhBuiltOnDevice = src->stats.bvhBuiltOnDevice; dst->stats.lastBvhBuildMs = 0.0;
    dst->scene_dirty = false; dst->tile_order_valid = false; dst->auto_choice = -1; dst->scene_version++;
    return 0;
}

int ensure_targets(rt_ctx* c)
{
    const int W = c->params.width, H = c->params.height;
    int r0 = c->nrows < 0 ? 0 : c->row0, nr = c->nrows < 0 ? H : c->nrows, rstride = 8;
    if (c->band_stride > 0) {
        // bands band_first, band_first + band_stride, ... of 8 rows each; the image's last band may be partial
        r0 = c->band_first * 8; rstride = c->band_stride * 8; nr = 0;
        for (int y = r0; y < H; y += rstride) nr += std::min(8, H - y);
        if (r0 > H) r0 = H;
    } else if (r0 < 0 || nr < 0 || r0 + nr > H) return fail(c, -6, "row strip [%d,%d) outside image height %d", r0, r0 + nr, H);
    if (W == c->target_w && H == c->target_h && r0 == c->target_row0 && nr == c->target_rows && rstride == c->target_row_stride) return 0;
    const size_t px = (size_t)W * nr;
    RT_HIP(c, c->d_frame.ensure(px));
    RT_HIP(c, c->d_accum.ensure(px));
    // a re-created render texture starts cleared (ShaderHelper.CreateRenderTexture, ShaderHelper.cs:186-205)
    if (px) {
        RT_HIP(c, hipMemsetAsync(c->d_frame.p, 0, px * sizeof(float4), c->stream));
        RT_HIP(c, hipMemsetAsync(c->d_accum.p, 0, px * sizeof(float4), c->stream));
    }
    c->target_pixels = px; c->target_w = W; c->target_h = H; c->target_row0 = r0; c->target_rows = nr; c->target_row_stride = rstride; c->tile_order_valid = false;
    c->stats.numRenderedFrames = 0; c->stats.totalKernelMs = 0;
    return 0;
}

enum class Variant { Fast, Counting, Flat };

// f(std::bool_constant<a>, std::bool_constant<b>, std::bool_constant<c>) for run-time a, b, c
template <class Fn> const void* dispatch3(bool a, bool b, bool c3, Fn f)
{
    auto lvl2 = [&](auto A) {
        auto lvl3 = [&](auto B) { return c3 ? f(A, B, std::true_type{}) : f(A, B, std::false_type{}); };
        return b ? lvl3(std::true_type{}) : lvl3(std::false_type{});
    };
    return a ? lvl2(std::true_type{}) : lvl2(std::false_type{});
}

// One kernel choice for all n_frames (kernel: 0 k_trace, 1 k_stream).
int launch_frames_k(rt_ctx* c, int first_frame, int n_frames, Variant var, int kernel)
{
    c->frames_traced += (uint64_t)std::max(n_frames, 0);
    if (!c) return -1;
    if (!c->have_params) return fail(c, -2, "rt_set_params has not been called");
    if (n_frames < 0) return fail(c, -2, "n_frames < 0");
    RT_HIP(c, hipSetDevice(c->device));
    if (c->geom_local) {
        if (c->scene_dirty) { int r = build_scene_local(c); if (r) return r; }
        else if (c->xf_dirty || local_scene_magnitude(c) > c->bvh.magnitude) { int r = run_geometry_kernels(c, true); if (r) return r; c->xf_dirty = false; }
    } else {
        if (c->scene_dirty) { int r = build_scene(c); if (r) return r; }
        else {
            const float om = std::max(camera_magnitude(c->params), c->sphere_mag);
            if (om > c->bvh.magnitude) { int r = repad_boxes(c, 2.0f * om); if (r) return r; }     // widen the box padding, keep the tree
        }
    }
    { int r = ensure_targets(c); if (r) return r; }
    if (c->target_pixels == 0 || n_frames == 0) return 0;

    rtk::DeviceScene S{};
    S.sph_geom = c->d_sph_geom.p; S.sph_mat = c->d_sph_mat.p; S.nodes = c->d_nodes.p; S.nodes_h = c->d_nodes_h.p;
    S.tri_geo = c->d_tri_geo.p; S.tri_nrm = c->d_tri_nrm.p; S.chunk_mat = c->d_chunk_mat.p; S.chunk_box = c->d_chunk_box.p;
    S.raw_tris = c->d_raw_tris.p; S.raw_chunk_range = c->d_raw_range.p;
    S.ns = (int)c->n_spheres; S.nn = (int)c->n_nodes;
    // the kernels address nodes and BVH-order triangles with 32-bit byte offsets (128 B and 48 B records)
    if (c->n_nodes >= ((size_t)1 << 25) || c->n_tris >= ((size_t)1 << 32) / 48)
        return fail(c, -7, "scene too large for 32-bit record offsets (%zu BVH nodes)", c->n_nodes);
    S.nt = (int)c->n_tris;
    S.nm = (int)c->n_chunks;

    rtk::FrameArgs F{};
    F.p = c->params;
    F.row0 = c->target_row0; F.nrows = c->target_rows; F.row_stride = c->target_row_stride;
    F.tile_w_log2 = 3;
    F.tiles_x = (c->target_w + 7) / 8; F.tiles_y = (c->target_rows + 7) / 8;
    if (kernel == 0 && c->opt_tile_w_log2 != 3) {        // k_trace only: other tile shapes (same 64 pixels per wave)
        F.tile_w_log2 = c->opt_tile_w_log2;
        const int tw = 1 << F.tile_w_log2, th = 64 >> F.tile_w_log2;
        F.tiles_x = (c->target_w + tw - 1) / tw; F.tiles_y = (c->target_rows + th - 1) / th;
    }
    // the counter-based mode is k_stream's Philox instantiation whatever kernel was asked for (its estimator spreads a pixel's samples
    // over the lanes of a wave); NumRaysPerPixel < 1 draws nothing in either mode and goes to k_trace
    const bool philox = c->params.rngMode == RT_RNG_PHILOX && c->params.numRaysPerPixel >= 1;
    if (philox && var == Variant::Flat) return fail(c, -2, "the flat validation kernel implements the PCG stream only");
    if (philox) kernel = 1;
    if (philox && (c->target_w > 65535 || c->target_rows > 65535)) return fail(c, -7, "the Philox mode addresses at most 65535 x 65535 pixels per context");
    if (philox && (c->params.numRaysPerPixel > 65000 || c->params.maxBounceCount > 32000))
        return fail(c, -7, "the Philox mode takes at most 65000 rays per pixel per frame and 32000 bounces (sample and bounce share one signed 32-bit register)");
    const bool stream = kernel == 1 && var != Variant::Flat && c->params.numRaysPerPixel >= 1        // PCG or Philox instantiation
                        && c->target_w <= 65535 && c->target_rows <= 65535;                          // (16-bit pixel coordinates in k_stream's item tables)
    F.stack_cap = std::max(1, c->bvh.maxStack) + (stream ? 3 : 0);    // the branch-free push writes up to 3 slots past the top
    const bool tile_kernel = !stream && var != Variant::Flat;   // k_trace, PCG or Philox
    if (tile_kernel && c->opt_lds_stack > 0) F.stack_cap = std::min(F.stack_cap, c->opt_lds_stack);
    if (tile_kernel) F.stack_cap = std::min(F.stack_cap, 64);        // a very deep tree spills past 64 entries instead of overflowing the LDS
    // k_stream: at most opt_stream_stack entries per lane in LDS (30 = five workgroups per CU); a deeper worst case spills
    const bool six_waves = stream && !philox && var == Variant::Fast && c->opt_compact_nodes != 0 && c->n_nodes > 0;       // rt_stream.hpp stream_waves()
    const int stream_stack = c->opt_stream_stack > 0 ? c->opt_stream_stack : (six_waves ? 24 : 30);
    const bool stream_spill = stream && F.stack_cap > stream_stack;
    if (stream_spill) F.stack_cap = stream_stack;
    F.full_sort = c->opt_full_sort;
    F.fixed_origin = camera_origin_is_fixed(c->params) ? 1 : 0;
    F.out_frame = c->d_frame.p; F.accum = c->d_accum.p;
    F.tile_counter = c->d_tile_counter; F.counters = c->d_counters;

    const size_t lds = var == Variant::Flat ? 0
                              : (size_t)F.stack_cap * 64 * sizeof(uint32_t) * rtk::kWavesPerBlock
                                + (stream ? (size_t)rtk::kGroupMax * sizeof(uint2) * rtk::kWavesPerBlock : 0);     // k_stream: + the groups' item tables
    if (lds > 160 * 1024) return fail(c, -7, "BVH needs a %d-entry traversal stack: exceeds the 160 KiB LDS", F.stack_cap);
    const bool counting = var == Variant::Counting;
    const bool compact = c->opt_compact_nodes != 0;            // k_trace / k_stream; the flat twin reads neither
    const void* fn = var == Variant::Flat ? (const void*)rtk::k_trace<false, true>
                   : stream ? rtk::stream_kernel(counting, philox, compact, c->n_nodes > 0)       // instantiated in rt_stream_kernels.hip
                   : c->n_nodes == 0      // spheres only: the instantiation compiled for six waves per SIMD
                            ? dispatch3(counting, false, false, [](auto C, auto, auto) { return (const void*)rtk::k_trace<decltype(C)::value, false, false, 6>; })
                            : dispatch3(counting, false, compact, [](auto C, auto, auto H) { return (const void*)rtk::k_trace<decltype(C)::value, false, decltype(H)::value>; });
    if (lds > 64 * 1024) RT_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    RT_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, rtk::kBlock, lds));
    if (per_cu < 1) return fail(c, -7, "kernel does not fit a CU (LDS %zu B)", lds);
    const int ntiles = F.tiles_x * F.tiles_y;
    // work items of a multi-frame launch = tiles x frames: a thin strip (one rank of eight: 4,080 tiles) still fills every resident wave
    // Philox mode: S = 16 / 4 / 1 sample lanes per pixel (the estimator's sub-streams, include/rt.h RT_RNG_PHILOX)
    const int sample_lanes_log2 = !philox ? 0 : c->params.numRaysPerPixel >= 16 ? 4 : c->params.numRaysPerPixel >= 4 ? 2 : 0;
    const bool stream_sync = stream && (c->opt_tile_sync || philox);       // k_stream taking whole work items (the Philox instantiation always does)
    const size_t frames_in_queue = ((stream_sync && c->opt_frame_batch != 1) ? (size_t)std::max(1, std::min(n_frames, 64)) : 1) << sample_lanes_log2;
    const int want = (int)std::min<size_t>(((size_t)ntiles * frames_in_queue + rtk::kWavesPerBlock - 1) / rtk::kWavesPerBlock, (size_t)1 << 20);
    if (c->opt_blocks_per_cu > 0) per_cu = std::min(per_cu, c->opt_blocks_per_cu);
    const int grid = std::max(1, std::min(want, per_cu * c->n_cu));
    rtk::StreamArgs A{};
    A.shade_threshold = std::max(1, std::min(64, c->opt_shade_threshold));
    A.total_pixels = (unsigned int)ntiles * 64u;
    A.tile_sync = stream_sync ? 1 : 0;
    A.sample_lanes_log2 = sample_lanes_log2;
    A.tiles_per_fetch = std::max(1, std::min(rtk::kGroupMax, c->opt_tiles_per_fetch));
    A.guide_div = 1;
    A.node_min = std::max(1, std::min(64, c->opt_node_min));
    const unsigned int gstack_stride = (unsigned int)grid * rtk::kBlock;
    if (philox) {      // where the lanes park their sub-stream sums until the wave's group of items is done: [wave][item of the group][3][64]
        RT_HIP(c, c->d_park.ensure((size_t)grid * rtk::kWavesPerBlock * (size_t)A.tiles_per_fetch * 192));
        F.park = c->d_park.p;
    }
    if (stream_spill) {
        RT_HIP(c, c->d_gstack.ensure((size_t)(c->bvh.maxStack + 3 - F.stack_cap) * gstack_stride));
        F.gstack = c->d_gstack.p; F.gstack_stride = gstack_stride;
    }
    if (tile_kernel && c->bvh.maxStack > F.stack_cap) {
        RT_HIP(c, c->d_gstack.ensure((size_t)(c->bvh.maxStack - F.stack_cap) * gstack_stride));
        F.gstack = c->d_gstack.p; F.gstack_stride = gstack_stride;
    }

    // k_trace can trace several frames per launch (work items = (frame, tile)): the persistent waves then balance over
    // frames as well — what matters when a rank's strip has about as many tiles as the chip has wave slots.
    int batch = 1;
    const bool stream_tiles = stream_sync;                       // k_stream taking whole tiles: same (frame, tile) items as k_trace
    if ((tile_kernel || stream_tiles) && n_frames > 1 && c->opt_frame_batch != 1) {
        const size_t budget = (size_t)4 << 30;                                  // <= 4 GiB of per-frame outputs (16 frames at 3840x2160)
        const size_t per_frame = c->target_pixels * sizeof(float4);
        batch = (int)std::min<size_t>((size_t)n_frames, std::max<size_t>(1, budget / per_frame));
        if (c->opt_frame_batch > 1) batch = std::min(batch, c->opt_frame_batch);
        batch = std::min(batch, 256);
        if (batch > 1) RT_HIP(c, c->d_batch.ensure((size_t)batch * c->target_pixels));
    }
    // LPT scheduling of the persistent waves: a launch records every tile's cost; the next ones hand tiles out costliest
    // first, so the end of a launch is filled with cheap tiles instead of waiting for a few expensive ones.
    const bool lpt = (tile_kernel || stream_sync) && c->opt_tile_lpt && ntiles > 1;
    c->lpt_active = lpt;
    bool record_costs = false;
    if (lpt) {
        if (c->tile_order_n != ntiles) { c->tile_order_valid = false; c->tile_order_n = ntiles; }
        RT_HIP(c, c->d_tile_cost.ensure(ntiles)); RT_HIP(c, c->d_tile_order.ensure(ntiles));
        if (!c->tile_order_valid || c->tile_order_stale) {
            record_costs = true;
            RT_HIP(c, hipMemsetAsync(c->d_tile_cost.p, 0, (size_t)ntiles * sizeof(uint32_t), c->stream));
        }
        F.tile_order = c->tile_order_valid ? c->d_tile_order.p : nullptr;
        F.tile_cost = record_costs ? c->d_tile_cost.p : nullptr;
    }
    // ---- camera rays' candidate lists: every pixel's camera rays start from <= 4 leaves found once per camera / scene (rt_primary.hpp)
    {
        std::string key((const char*)&c->params, sizeof c->params);
        const unsigned long long geo[6] = { c->scene_version, (unsigned long long)c->target_row0, (unsigned long long)c->target_rows,
                                            (unsigned long long)c->target_row_stride, (unsigned long long)c->target_w, (unsigned long long)c->target_h };
        key.append((const char*)geo, sizeof geo);
        const bool eligible = stream && c->opt_primary_lists && c->n_nodes > 0 && F.fixed_origin && c->target_pixels > 0 && c->bvh.maxStack <= 160;
        if (eligible) {                 // (the build takes well under a millisecond at 1080p: a camera that moves every frame pays it every frame and still gains)
            if (key != c->primary_key) {
                RT_HIP(c, c->d_primary.ensure(c->target_pixels)); RT_HIP(c, c->d_focus.ensure(c->target_pixels)); RT_HIP(c, c->d_primary_counts.ensure(4));
                RT_HIP(c, hipMemsetAsync(c->d_primary_counts.p, 0, 4 * sizeof(unsigned int), c->stream));
                rtp::PrimaryArgs PA{};
                PA.p = c->params; PA.row0 = c->target_row0; PA.nrows = c->target_rows; PA.row_stride = c->target_row_stride;
                PA.lists = c->d_primary.p; PA.focus = c->d_focus.p; PA.counts = c->d_primary_counts.p;
                const int tiles = ((c->target_w + 7) / 8) * ((c->target_rows + 7) / 8);
                PA.stack_cap = std::max(1, c->bvh.maxStack);                       // (the whole worst case in LDS: at most 160 entries x 64 lanes x 4 B = 40 KB per wave)
                const size_t plds = (size_t)PA.stack_cap * 64 * sizeof(uint32_t);
                RT_HIP(c, hipEventRecord(c->evg0, c->stream));
                hipLaunchKernelGGL(rtp::k_primary_lists, dim3(tiles), dim3(64), plds, c->stream, S, PA);
                RT_HIP(c, hipGetLastError());
                RT_HIP(c, hipEventRecord(c->evg1, c->stream));
                unsigned int h[4];
                RT_HIP(c, hipMemcpyAsync(h, c->d_primary_counts.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
                RT_HIP(c, hipStreamSynchronize(c->stream));
                float ms = 0.f;
                RT_HIP(c, hipEventElapsedTime(&ms, c->evg0, c->evg1));
                for (int k = 0; k < 4; ++k) c->stats.primaryLists[k] = h[k];
                c->stats.lastPrimaryListsMs = ms; c->stats.primaryListBuilds++;
                c->primary_key = key;
            }
            F.primary = c->d_primary.p; F.focus = c->d_focus.p;
        }
    }
    RT_HIP(c, hipMemsetAsync(c->d_counters, 0, rtk::kNumCounters * sizeof(unsigned long long), c->stream));
    RT_HIP(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < n_frames; ) {
        int nb = batch > 1 ? std::min(batch, n_frames - i) : 1;
        // k_stream, frame-interleaved sub-tiles: a wave = (4x4 or 2x2 pixels) x (4 or 16 frames); launches take whole frame
        // groups, the remainder of the render goes out as 8x8 x 1 items
        A.n16 = A.n4 = 0; A.n1 = nb;
        if (stream_tiles) {
            // as many groups of 16 frames as fit, then groups of 4, the rest one by one — all in this one launch
            int rem = nb;
            if (!philox && c->opt_stream_tile >= 4) { A.n16 = rem / 16; rem -= A.n16 * 16; }      // (Philox: sample lanes, not frames, fill a wave)
            if (!philox && c->opt_stream_tile >= 2) { A.n4 = rem / 4; rem -= A.n4 * 4; }
            A.n1 = rem;
            // groups shrink towards the end of the launch (k_stream: guided self-scheduling): up to tiles_per_fetch items per fetch
            // while more than guide x (waves of the launch) x that many items are left
            A.tiles_per_fetch = std::max(1, std::min(rtk::kGroupMax, c->opt_tiles_per_fetch));
            // (Philox: the units of a group are small — a few samples — and handed out dynamically, so big groups balance by themselves and
            // only the launch's last round of groups needs to shrink: a single 1080p frame 19.2 -> 17.5 ms with guide 1; PCG units are whole
            // pixels and need the finer tail: 19.9 -> 47.9 ms with it)
            A.guide_div = std::max(1, grid * rtk::kWavesPerBlock * (philox ? std::max(1, c->opt_fetch_guide_philox) : std::max(1, c->opt_fetch_guide)));
        }
        F.frame = first_frame + i;
        F.frames_in_launch = nb; F.frame_stride = (unsigned int)c->target_pixels;
        F.out_frame = nb > 1 ? c->d_batch.p : c->d_frame.p;
        RT_HIP(c, hipMemsetAsync(c->d_tile_counter, 0, sizeof(unsigned int), c->stream));
        {
            void* args[3] = { (void*)&S, (void*)&F, (void*)&A };    // k_trace takes (S, F) only
            RT_HIP(c, hipLaunchKernel(fn, dim3(grid), dim3(rtk::kBlock), args, lds, c->stream));
        }
        RT_HIP(c, hipGetLastError());
        if (nb > 1) {
            const int ag = (int)std::min<size_t>((c->target_pixels + 255) / 256, (size_t)c->n_cu * 8);
            hipLaunchKernelGGL(rtk::k_accumulate<>, dim3(ag), dim3(256), 0, c->stream, c->d_batch.p, c->d_accum.p, c->d_frame.p,
                               c->target_pixels, F.frame_stride, F.frame, nb);
            RT_HIP(c, hipGetLastError());
        }
        i += nb;
    }
    c->stats.lastFramesPerLaunch = batch;
    c->stats.lastKernel = var == Variant::Flat ? 4 : stream ? 1 : 0;
    c->stats.lastFramesInterleaved = stream ? (philox ? 1 : A.n16 ? 16 : A.n4 ? 4 : 1) : 1;
    c->stats.lastSampleLanes = philox ? 1 << sample_lanes_log2 : 1;
    RT_HIP(c, hipEventRecord(c->ev1, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    RT_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->stats.lastKernelMs = ms; c->stats.totalKernelMs += ms;
    if (record_costs) {
        // costliest tiles first for the next launches: counting sort on the device, no host round trip
        RT_HIP(c, c->d_tile_hist.ensure(rtg::kCostBuckets));
        RT_HIP(c, hipMemsetAsync(c->d_tile_hist.p, 0, rtg::kCostBuckets * sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(rtg::k_tile_hist, dim3((ntiles + 255) / 256), dim3(256), 0, c->stream, c->d_tile_cost.p, (uint32_t)ntiles, c->d_tile_hist.p);
        hipLaunchKernelGGL(rtg::k_tile_scan, dim3(1), dim3(1024), 0, c->stream, c->d_tile_hist.p);
        hipLaunchKernelGGL(rtg::k_tile_scatter, dim3((ntiles + 255) / 256), dim3(256), 0, c->stream, c->d_tile_cost.p, (uint32_t)ntiles,
                           c->d_tile_hist.p, c->d_tile_order.p);
        RT_HIP(c, hipGetLastError());
        c->tile_order_valid = true; c->tile_order_stale = false;
    }
    c->stats.numRenderedFrames += n_frames;
    {
        unsigned long long h[rtk::kNumCounters];
        RT_HIP(c, hipMemcpy(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
        c->stats.rays = h[0];                        // counted by every variant
        if (var == Variant::Counting) {
            c->stats.sphereTests = h[1]; c->stats.nodeVisits = h[2]; c->stats.triTests = h[3]; c->stats.hits = h[4];
            for (int k = 0; k < 5; ++k) { c->stats.phaseLanes[k] = h[5 + k]; c->stats.phaseExecs[k] = h[10 + k]; }
            for (int k = 0; k < rtk::kNumRegions; ++k) c->stats.regionExecs[k] = h[15 + k];
        } else {
            c->stats.sphereTests = c->stats.nodeVisits = c->stats.triTests = c->stats.hits = 0;
            for (int k = 0; k < 5; ++k) c->stats.phaseLanes[k] = c->stats.phaseExecs[k] = 0;
            for (int k = 0; k < rtk::kNumRegions; ++k) c->stats.regionExecs[k] = 0;
        }
    }
    return 0;
}

// Kernel choice "auto" (option kernel = -1, the default): k_trace and k_stream (resumable traversal, stragglers deferred)
// produce the same bits, and which one is faster depends on the scene (measured: k_stream +8 % on the 100k-triangle
// workload, -5 % on the 1M-triangle one, -13 % on spheres only).  So the first frames after a scene / camera change are
// used as the measurement: one frame records the tile costs, one is timed with k_trace, one with k_stream (all three are
// ordinary frames of the render — nothing is traced twice); the faster kernel takes the rest.
int launch_frames(rt_ctx* c, int first_frame, int n_frames, Variant var)
{
    if (!c) return -1;
    const bool eligible = c->opt_kernel < 0 && var != Variant::Flat && c->have_params
                          && c->params.numRaysPerPixel >= 1 && c->params.rngMode != RT_RNG_PHILOX;   // (Philox: always k_stream)
    if (!eligible) {
        const int kernel = c->opt_kernel < 0 ? 0 : c->opt_kernel;
        return launch_frames_k(c, first_frame, n_frames, var, kernel);
    }
    // (with the costliest-first order switched off, or a single tile, there is no order to wait for)
    auto order_ready = [&]() { return c->tile_order_valid || !c->lpt_active; };
    // Decided: single frames go to the kernel that won the single-frame timing; launches of 4 frames or more over a BVH go to
    // k_stream, whose frame-interleaved items (2x2 pixels x 16 frames per wave) have no counterpart in k_trace (measured on both
    // triangle workloads: +13 % / +10 % over its own 8x8 x 1 items, which is what the single-frame timing compares)
    auto decided = [&](int frames) { return (c->stats.numBvhNodes > 0 && frames >= 4 && c->opt_tile_sync) ? 1 : c->auto_choice; };
    if (c->auto_choice >= 0 && !c->scene_dirty && order_ready())
        return launch_frames_k(c, first_frame, n_frames, var, decided(n_frames));

    if (n_frames == 0) {            // scene / geometry update only (rt_multi's build on its first context)
        const bool was_dirty = c->scene_dirty;
        const int r = launch_frames_k(c, first_frame, 0, var, c->auto_choice >= 0 ? c->auto_choice : 0);
        if (was_dirty) { c->auto_choice = -1; c->auto_ms[0] = c->auto_ms[1] = -1.0; }     // a new scene is measured again, as on the contexts that receive it
        return r;
    }
    rt_stats sum{}; bool any = false;
    auto add = [&]() {
        const rt_stats& s = c->stats;
        sum.rays += s.rays; sum.sphereTests += s.sphereTests; sum.nodeVisits += s.nodeVisits; sum.triTests += s.triTests; sum.hits += s.hits;
        for (int k = 0; k < 5; ++k) { sum.phaseLanes[k] += s.phaseLanes[k]; sum.phaseExecs[k] += s.phaseExecs[k]; }
        for (int k = 0; k < rtk::kNumRegions; ++k) sum.regionExecs[k] += s.regionExecs[k];
        sum.lastKernelMs += s.lastKernelMs; any = true;
    };
    int done = 0;
    while (done < n_frames) {
        int kernel, count = 1;
        if (c->scene_dirty || !order_ready()) { kernel = 0; c->auto_choice = -1; c->auto_ms[0] = c->auto_ms[1] = -1.0; }   // records the tile costs
        else if (c->auto_choice >= 0) { count = n_frames - done; kernel = decided(count); }
        else if (c->auto_ms[0] < 0) kernel = 0;
        else kernel = 1;
        const bool probing = c->auto_choice < 0 && !c->scene_dirty && order_ready();
        int r = launch_frames_k(c, first_frame + done, count, var, kernel);
        if (r) return r;
        add();
        if (c->target_pixels == 0) { done += count; continue; }
        // a kernel variant's very first launch in a context also pays its one-off set-up (code upload, the scratch ring of
        // k_stream): that frame is rendered like any other but not used as the timing
        const int variant = kernel * 4 + (c->params.rngMode == RT_RNG_PHILOX ? 2 : 0) + (c->opt_compact_nodes ? 1 : 0);
        const bool first_use = !((c->variants_launched >> variant) & 1u);
        c->variants_launched |= 1u << variant;
        if (probing) {
            if (!first_use) c->auto_ms[kernel] = c->stats.lastKernelMs;
            if (c->auto_ms[0] >= 0 && c->auto_ms[1] >= 0) c->auto_choice = (c->stats.numBvhNodes > 0 && c->auto_ms[1] < c->auto_ms[0]) ? 1 : 0;
        }
        done += count;
    }
    if (any) {
        c->stats.rays = sum.rays; c->stats.sphereTests = sum.sphereTests; c->stats.nodeVisits = sum.nodeVisits;
        c->stats.triTests = sum.triTests; c->stats.hits = sum.hits;
        for (int k = 0; k < 5; ++k) { c->stats.phaseLanes[k] = sum.phaseLanes[k]; c->stats.phaseExecs[k] = sum.phaseExecs[k]; }
        for (int k = 0; k < rtk::kNumRegions; ++k) c->stats.regionExecs[k] = sum.regionExecs[k];
        c->stats.lastKernelMs = sum.lastKernelMs;
    }
    c->stats.autoKernel = c->auto_choice;
    return 0;
}

int read_target(rt_ctx* c, bool accum, float* dst, size_t n_floats, bool to_device)
{
    if (!c) return -1;
    if (!dst && n_floats) return fail(c, -2, "null destination");
    if (c->have_params) { int r = ensure_targets(c); if (r) return r; }
    if (n_floats != c->target_pixels * 4)
        return fail(c, -2, "expected %zu floats (rows*width*4), got %zu", c->target_pixels * 4, n_floats);
    if (!n_floats) return 0;
    RT_HIP(c, hipSetDevice(c->device));
    const float4* src = accum ? c->d_accum.p : c->d_frame.p;
    RT_HIP(c, hipMemcpyAsync(dst, src, n_floats * sizeof(float), to_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- queued submission ----------------------------------------------------------------------------------------------------------
void queue_worker(rt_ctx* c)
{
    std::unique_lock<std::mutex> lk(c->q_mu);
    for (;;) {
        c->q_cv.wait(lk, [&] { return c->q_stop || !c->q_frames.empty(); });
        if (c->q_frames.empty()) break;                 // (stop is honoured once the queue has drained)
        if (c->opt_queue_linger_us > 0 && !c->q_stop)
            c->q_cv.wait_for(lk, std::chrono::microseconds(c->opt_queue_linger_us), [&] { return c->q_stop || (int)c->q_frames.size() >= c->opt_queue_depth; });
        // the longest run of consecutive frame indices at the head of the queue: one launch (rt_render(first, n))
        const int first = c->q_frames.front(); int n = 0;
        while (!c->q_frames.empty() && c->q_frames.front() == first + n && n < c->opt_queue_depth) { c->q_frames.pop_front(); ++n; }
        c->q_busy = true;
        const bool failed = c->q_rc != 0;
        lk.unlock();
        const int r = failed ? 0 : launch_frames(c, first, n, Variant::Fast);      // after a failure the rest of the queue is dropped
        lk.lock();
        if (r && !c->q_rc) { c->q_rc = r; c->q_err = c->err; }
        if (!failed && !r) c->stats.queuedLaunches++;
        c->q_busy = false;
        c->q_idle.notify_all();
    }
}

// Every entry point but rt_submit_frame starts here: the queue is empty and the worker idle before anything else touches the context
// (a context has one caller thread; the worker is the library's own).  Returns the first error a queued launch met, once.
int settle(rt_ctx* c)
{
    if (!c->q_started) return 0;
    std::unique_lock<std::mutex> lk(c->q_mu);
    c->q_idle.wait(lk, [&] { return c->q_frames.empty() && !c->q_busy; });
    const int r = c->q_rc;
    if (r) { c->err = "queued frame: " + c->q_err; c->q_rc = 0; }
    return r;
}
#define RT_SETTLE(c) do { const int qr_ = settle(c); if (qr_) return qr_; } while (0)

} // namespace

extern "C" {

int rt_abi_version(void) { return 1; }

int rt_sizeof(const char* name)
{
    if (!name) return -1;
    if (!std::strcmp(name, "rt_material")) return (int)sizeof(rt_material);
    if (!std::strcmp(name, "rt_sphere"))   return (int)sizeof(rt_sphere);
    if (!std::strcmp(name, "rt_triangle")) return (int)sizeof(rt_triangle);
    if (!std::strcmp(name, "rt_meshinfo")) return (int)sizeof(rt_meshinfo);
    if (!std::strcmp(name, "rt_params"))   return (int)sizeof(rt_params);
    if (!std::strcmp(name, "rt_stats"))    return (int)sizeof(rt_stats);
    if (!std::strcmp(name, "rt_mesh_transform")) return (int)sizeof(rt_mesh_transform);
    if (!std::strcmp(name, "rt_local_chunk")) return (int)sizeof(rt_local_chunk);
    if (!std::strcmp(name, "rt_multi_info")) return (int)sizeof(rt_multi_info);
    return -1;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

rt_ctx* rt_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(nullptr, -1, "no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "count = 0" : hipGetErrorString(e)); return nullptr; }
    if (device < 0 || device >= n) { fail(nullptr, -1, "device %d out of range [0,%d)", device, n); return nullptr; }
    if ((e = hipSetDevice(device)) != hipSuccess) { fail(nullptr, -1, "hipSetDevice: %s", hipGetErrorString(e)); return nullptr; }
    rt_ctx* c = new rt_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { fail(nullptr, -1, "hipGetDeviceProperties: %s", hipGetErrorString(e)); delete c; return nullptr; }
    c->n_cu = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess
        || (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess
        || (e = hipEventCreate(&c->evg0)) != hipSuccess || (e = hipEventCreate(&c->evg1)) != hipSuccess
        || (e = hipMalloc((void**)&c->d_tile_counter, sizeof(unsigned int))) != hipSuccess
        || (e = hipMalloc((void**)&c->d_counters, rtk::kNumCounters * sizeof(unsigned long long))) != hipSuccess) {
        fail(nullptr, -1, "context setup: %s", hipGetErrorString(e));
        rt_destroy(c);
        return nullptr;
    }
    c->stream = c->own_stream;
    return c;
}

void rt_destroy(rt_ctx* c)
{
    if (!c) return;
    if (c->q_started) {
        { std::lock_guard<std::mutex> lk(c->q_mu); c->q_stop = true; }
        c->q_cv.notify_all();
        c->q_thread.join();
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->d_sph_geom.release(); c->d_sph_mat.release(); c->d_nodes.release(); c->d_nodes_h.release(); c->d_tri_geo.release(); c->d_tri_nrm.release();
    c->d_chunk_mat.release(); c->d_chunk_box.release(); c->d_raw_tris.release(); c->d_raw_range.release();
    c->d_primary.release(); c->d_focus.release(); c->d_primary_counts.release();
    c->d_frame.release(); c->d_accum.release(); c->d_gstack.release(); c->d_park.release(); c->d_display.release(); c->d_batch.release(); c->d_tile_order.release(); c->d_tile_cost.release(); c->d_tile_hist.release();
    if (c->d_tile_counter) (void)hipFree(c->d_tile_counter);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->evg0) (void)hipEventDestroy(c->evg0);
    if (c->evg1) (void)hipEventDestroy(c->evg1);
    c->bvh_ws.release();
    c->d_local_tris.release(); c->d_tri_mesh.release(); c->d_tri_chunk.release(); c->d_tri_rank.release(); c->d_order.release(); c->d_xf.release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int rt_set_stream(rt_ctx* c, void* hip_stream)
{
    if (!c) return -1;
    RT_SETTLE(c);
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return 0;
}

int rt_set_params(rt_ctx* c, const rt_params* p)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (!p) return fail(c, -2, "null params");
    if (p->width < 0 || p->height < 0 || (int64_t)p->width * p->height > (int64_t)1 << 31) return fail(c, -2, "bad target size %dx%d", p->width, p->height);
    if (p->numRaysPerPixel < 0) return fail(c, -2, "numRaysPerPixel < 0");
    if (p->rngMode != RT_RNG_PCG && p->rngMode != RT_RNG_PHILOX) return fail(c, -2, "unknown rngMode %d", p->rngMode);
    if (p->intersectMode != RT_INTERSECT_FLAT_CHUNKS && p->intersectMode != RT_INTERSECT_BRUTE) return fail(c, -2, "unknown intersectMode %d", p->intersectMode);
    // a moved camera keeps the previous frame's tile order and kernel choice as predictors; the next launch re-measures the costs
    if (!c->have_params || std::memcmp(&c->params, p, sizeof *p) != 0) c->tile_order_stale = true;
    if (c->have_params && c->params.rngMode != p->rngMode) c->auto_choice = -1;      // the kernels' relative speed depends on the RNG
    c->params = *p; c->have_params = true;
    return 0;
}

int rt_upload_spheres(rt_ctx* c, const rt_sphere* s, int n)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (n < 0 || (n > 0 && !s)) return fail(c, -2, "bad sphere upload (n=%d)", n);
    c->h_spheres.assign(s, s + n); c->scene_dirty = true;
    return 0;
}
int rt_upload_triangles(rt_ctx* c, const rt_triangle* t, int n)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (n < 0 || (n > 0 && !t)) return fail(c, -2, "bad triangle upload (n=%d)", n);
    c->h_tris.assign(t, t + n); c->scene_dirty = true; c->geom_local = false;
    return 0;
}
int rt_upload_meshinfo(rt_ctx* c, const rt_meshinfo* m, int n)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (n < 0 || (n > 0 && !m)) return fail(c, -2, "bad meshinfo upload (n=%d)", n);
    c->h_mesh.assign(m, m + n); c->scene_dirty = true; c->geom_local = false;
    return 0;
}

int rt_upload_local_meshes(rt_ctx* c, const rt_triangle* tris, int n_tris, const rt_local_chunk* chunks, int n_chunks, int n_meshes)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (n_tris < 0 || n_chunks < 0 || n_meshes < 0 || (n_tris > 0 && !tris) || (n_chunks > 0 && !chunks)) return fail(c, -2, "bad local mesh upload");
    if (n_tris > (1 << 28)) return fail(c, -3, "too many triangles (%d)", n_tris);
    std::vector<uint8_t> seen(n_tris, 0);
    std::vector<float> radius(n_meshes, 0.f);
    for (int m = 0; m < n_chunks; ++m) {
        const rt_local_chunk& ch = chunks[m];
        if ((uint64_t)ch.firstTriangleIndex + ch.numTriangles > (uint64_t)n_tris) return fail(c, -4, "chunk %d addresses triangles beyond the %d uploaded", m, n_tris);
        if (ch.meshIndex >= (uint32_t)n_meshes) return fail(c, -4, "chunk %d refers to mesh %u of %d", m, ch.meshIndex, n_meshes);
        for (uint32_t i = 0; i < ch.numTriangles; ++i) {
            uint8_t& s = seen[ch.firstTriangleIndex + i];
            if (s) return fail(c, -5, "triangle %u is referenced by more than one chunk", ch.firstTriangleIndex + i);
            s = 1;
            const float* p = tris[ch.firstTriangleIndex + i].posA;
            for (int k = 0; k < 9; ++k) radius[ch.meshIndex] = std::max(radius[ch.meshIndex], std::fabs(p[k]));
        }
    }
    for (int t = 0; t < n_tris; ++t) if (!seen[t]) return fail(c, -5, "triangle %d belongs to no chunk", t);
    for (float& r : radius) r *= 1.7320508f;
    c->h_local_tris.assign(tris, tris + n_tris); c->h_lchunks.assign(chunks, chunks + n_chunks);
    c->mesh_radius = radius; c->n_meshes = n_meshes;
    c->geom_local = true; c->scene_dirty = true;
    return 0;
}

int rt_set_mesh_transforms(rt_ctx* c, const rt_mesh_transform* xf, int n_meshes)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (n_meshes < 0 || (n_meshes > 0 && !xf)) return fail(c, -2, "bad transform upload");
    if (c->geom_local && n_meshes != c->n_meshes) return fail(c, -2, "%d transforms for %d meshes", n_meshes, c->n_meshes);
    c->h_xf.assign(xf, xf + n_meshes);
    c->xf_dirty = true;
    return 0;
}

int rt_read_world_geometry(rt_ctx* c, rt_triangle* tris_out, int n_tris, rt_meshinfo* mi_out, int n_chunks)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (!c->geom_local) return fail(c, -2, "no local meshes uploaded");
    if (!c->have_params) return fail(c, -2, "rt_set_params has not been called");
    if (n_tris != (int)c->h_local_tris.size() || n_chunks != (int)c->h_lchunks.size()) return fail(c, -2, "size mismatch");
    RT_HIP(c, hipSetDevice(c->device));
    if (c->scene_dirty) { int r = build_scene_local(c); if (r) return r; }
    else if (c->xf_dirty) { int r = run_geometry_kernels(c, true); if (r) return r; c->xf_dirty = false; }
    if (n_tris) RT_HIP(c, hipMemcpy(tris_out, c->d_raw_tris.p, (size_t)n_tris * sizeof(rt_triangle), hipMemcpyDeviceToHost));
    std::vector<float4> box(2 * (size_t)n_chunks);
    if (n_chunks) RT_HIP(c, hipMemcpy(box.data(), c->d_chunk_box.p, box.size() * sizeof(float4), hipMemcpyDeviceToHost));
    for (int m = 0; m < n_chunks; ++m) {
        const rt_local_chunk& ch = c->h_lchunks[m];
        rt_meshinfo& mi = mi_out[m];
        mi.firstTriangleIndex = ch.firstTriangleIndex; mi.numTriangles = ch.numTriangles; mi.material = ch.material;
        mi.boundsMin[0] = box[2 * m].x; mi.boundsMin[1] = box[2 * m].y; mi.boundsMin[2] = box[2 * m].z;
        mi.boundsMax[0] = box[2 * m + 1].x; mi.boundsMax[1] = box[2 * m + 1].y; mi.boundsMax[2] = box[2 * m + 1].z;
    }
    return 0;
}

int rt_set_option(rt_ctx* c, const char* name, int value)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (!name) return fail(c, -2, "null option name");
    if (!std::strcmp(name, "kernel")) {
        if (value < -1 || value > 1) return fail(c, -2, "kernel must be -1 (auto), 0 (k_trace) or 1 (k_stream)");
        c->opt_kernel = value;
    }
    else if (!std::strcmp(name, "shade_threshold")) { if (value < 1 || value > 64) return fail(c, -2, "shade_threshold must be in [1,64]"); c->opt_shade_threshold = value; }
    else if (!std::strcmp(name, "stream_stack")) { if (value != 0 && (value < 4 || value > 128)) return fail(c, -2, "stream_stack must be 0 (automatic) or in [4,128]"); c->opt_stream_stack = value; }
    else if (!std::strcmp(name, "lds_stack")) { if (value < 0 || value > 64) return fail(c, -2, "lds_stack must be in [0,64]"); c->opt_lds_stack = value; }
    else if (!std::strcmp(name, "bvh_bins")) { if (value < 2 || value > 128) return fail(c, -2, "bvh_bins must be in [2,128]"); if (value != c->opt_bvh_bins) c->scene_dirty = true; c->opt_bvh_bins = value; }
    else if (!std::strcmp(name, "bvh_reinsert")) { if (value < 0 || value > 16) return fail(c, -2, "bvh_reinsert must be in [0,16]"); if (value != c->opt_bvh_reinsert) c->scene_dirty = true; c->opt_bvh_reinsert = value; }
    else if (!std::strcmp(name, "bvh_cost_exp")) { if (value < 10 || value > 300) return fail(c, -2, "bvh_cost_exp must be in [10,300] (percent)"); if (value != c->opt_bvh_cost_exp) c->scene_dirty = true; c->opt_bvh_cost_exp = value; }
    else if (!std::strcmp(name, "max_leaf")) { if (value < 1 || value > rtbvh::kMaxLeaf) return fail(c, -2, "max_leaf must be in [1,4]"); if (value != c->opt_max_leaf) c->scene_dirty = true; c->opt_max_leaf = value; }
    else if (!std::strcmp(name, "bvh_collapse")) { if (value < 0 || value > 2) return fail(c, -2, "bvh_collapse must be 0 (greedy), 1 (cost-driven, leaves formed by the collapse) or 2 (cost-driven over the split search's leaves)"); if (value != c->opt_bvh_collapse) c->scene_dirty = true; c->opt_bvh_collapse = value; }
    else if (!std::strcmp(name, "bvh_node_cost")) { if (value < 1 || value > 10000) return fail(c, -2, "bvh_node_cost must be in [1,10000] (percent of a triangle test)"); if (value != c->opt_bvh_node_cost) c->scene_dirty = true; c->opt_bvh_node_cost = value; }
    else if (!std::strcmp(name, "tile_w_log2")) { if (value < 0 || value > 6) return fail(c, -2, "tile_w_log2 must be in [0,6]"); c->opt_tile_w_log2 = value; }
    else if (!std::strcmp(name, "tile_lpt")) { c->opt_tile_lpt = value ? 1 : 0; c->tile_order_valid = false; }
    else if (!std::strcmp(name, "frame_batch")) { if (value < 0 || value > 1024) return fail(c, -2, "frame_batch must be in [0,1024]"); c->opt_frame_batch = value; }
    else if (!std::strcmp(name, "fetch_guide")) { if (value < 1 || value > 64) return fail(c, -2, "fetch_guide must be in [1,64]"); c->opt_fetch_guide = value; }
    else if (!std::strcmp(name, "fetch_guide_philox")) { if (value < 1 || value > 64) return fail(c, -2, "fetch_guide_philox must be in [1,64]"); c->opt_fetch_guide_philox = value; }
    else if (!std::strcmp(name, "tiles_per_fetch")) { if (value < 1 || value > 64) return fail(c, -2, "tiles_per_fetch must be in [1,64]"); c->opt_tiles_per_fetch = value; }
    else if (!std::strcmp(name, "node_min")) { if (value < 1 || value > 64) return fail(c, -2, "node_min must be in [1,64]"); c->opt_node_min = value; }
    else if (!std::strcmp(name, "tile_sync")) c->opt_tile_sync = value ? 1 : 0;
    else if (!std::strcmp(name, "compact_nodes")) c->opt_compact_nodes = value ? 1 : 0;
    else if (!std::strcmp(name, "device_bvh")) { if (value < -1 || value > 1) return fail(c, -2, "device_bvh must be -1 (automatic), 0 or 1"); if (value != c->opt_device_bvh) c->scene_dirty = true; c->opt_device_bvh = value; }
    else if (!std::strcmp(name, "bvh_radius")) { if (value == 0 || value < -rtgb::kMaxRadius || value > rtgb::kMaxRadius) return fail(c, -2, "bvh_radius must be in [1,64] (negative: the same radius in every round)"); if (value != c->opt_bvh_radius) c->scene_dirty = true; c->opt_bvh_radius = value; }
    else if (!std::strcmp(name, "peer_copies")) { if (value != 0 && value != 1) return fail(c, -2, "peer_copies must be 0 or 1"); c->opt_peer_copies = value; }
    else if (!std::strcmp(name, "bvh_treelet_ratio")) { if (value < 2 || value > 64) return fail(c, -2, "bvh_treelet_ratio must be in [2,64]"); if (value != c->opt_bvh_treelet_ratio) c->scene_dirty = true; c->opt_bvh_treelet_ratio = value; }
    else if (!std::strcmp(name, "bvh_treelet_isolate")) { if (value < 0 || value > 1) return fail(c, -2, "bvh_treelet_isolate must be 0 or 1"); if (value != c->opt_bvh_treelet_isolate) c->scene_dirty = true; c->opt_bvh_treelet_isolate = value; }
    else if (!std::strcmp(name, "bvh_treelet_first")) { if (value < 1 || value > 4096) return fail(c, -2, "bvh_treelet_first must be in [1,4096]"); if (value != c->opt_bvh_treelet_first) c->scene_dirty = true; c->opt_bvh_treelet_first = value; }
    else if (!std::strcmp(name, "bvh_treelets")) { if (value < 0 || value > 16) return fail(c, -2, "bvh_treelets must be in [0,16] (passes)"); if (value != c->opt_bvh_treelets) c->scene_dirty = true; c->opt_bvh_treelets = value; }
    else if (!std::strcmp(name, "bvh_top")) { if (value < 0 || value > (1 << 20)) return fail(c, -2, "bvh_top must be in [0,1048576] (0 = the device's clustering builds the whole tree)"); if (value != c->opt_bvh_top) c->scene_dirty = true; c->opt_bvh_top = value; }
    else if (!std::strcmp(name, "rebuild_percent")) { if (value < 0 || value > 100000) return fail(c, -2, "rebuild_percent must be in [0,100000] (0 = never rebuild)"); c->opt_rebuild_percent = value; }
    else if (!std::strcmp(name, "stream_tile")) { if (value != 0 && value != 2 && value != 4) return fail(c, -2, "stream_tile must be 0 (8x8 pixels x 1 frame), 2 (4x4 x 4 frames) or 4 (2x2 x 16 frames)"); c->opt_stream_tile = value; }
    else if (!std::strcmp(name, "full_sort")) c->opt_full_sort = value ? 1 : 0;
    else if (!std::strcmp(name, "primary_lists")) { if (value != 0 && value != 1) return fail(c, -2, "primary_lists must be 0 or 1"); c->opt_primary_lists = value; }
    else if (!std::strcmp(name, "queue_depth")) { if (value < 1 || value > 256) return fail(c, -2, "queue_depth must be in [1,256]"); c->opt_queue_depth = value; }
    else if (!std::strcmp(name, "queue_linger_us")) { if (value < 0 || value > 1000000) return fail(c, -2, "queue_linger_us must be in [0,1000000]"); c->opt_queue_linger_us = value; }
    else if (!std::strcmp(name, "blocks_per_cu")) { if (value < 0) return fail(c, -2, "blocks_per_cu must be >= 0"); c->opt_blocks_per_cu = value; }
    else return fail(c, -2, "unknown option '%s'", name);
    return 0;
}

int rt_set_rows(rt_ctx* c, int row0, int nrows)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (row0 < 0 || nrows < 0) return fail(c, -2, "bad row strip (%d,%d)", row0, nrows);
    c->row0 = row0; c->nrows = nrows; c->band_stride = 0;
    return 0;
}

int rt_set_bands(rt_ctx* c, int first_band, int band_stride)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (first_band < 0 || band_stride < 1 || first_band >= band_stride) return fail(c, -2, "bad band pattern (%d,%d)", first_band, band_stride);
    c->band_first = first_band; c->band_stride = band_stride;
    return 0;
}

int rt_render_frame(rt_ctx* c, int frame_index) { if (!c) return -1; RT_SETTLE(c); return launch_frames(c, frame_index, 1, Variant::Fast); }
int rt_render(rt_ctx* c, int first_frame, int n_frames) { if (!c) return -1; RT_SETTLE(c); return launch_frames(c, first_frame, n_frames, Variant::Fast); }
int rt_render_counting(rt_ctx* c, int first_frame, int n_frames) { if (!c) return -1; RT_SETTLE(c); return launch_frames(c, first_frame, n_frames, Variant::Counting); }
int rt_render_frame_flat(rt_ctx* c, int frame_index) { if (!c) return -1; RT_SETTLE(c); return launch_frames(c, frame_index, 1, Variant::Flat); }

// Queued submission.  rt_submit_frame returns at once; a worker thread of the library traces the queued frames in launches of whatever
// has queued up while the previous launch ran (consecutive frame indices share a launch: frame-interleaved work items, one launch tail)
// and accumulates them in submission order — the image equals rt_render's over the same frames, bit for bit.
int rt_submit_frame(rt_ctx* c, int frame_index)
{
    if (!c) return -1;
    if (!c->have_params) { RT_SETTLE(c); return fail(c, -2, "rt_set_params has not been called"); }
    {
        std::lock_guard<std::mutex> lk(c->q_mu);
        if (c->q_rc) return c->q_rc;                    // a queued launch failed: rt_wait (or any other call) reports it
        if (!c->q_started) { c->q_thread = std::thread(queue_worker, c); c->q_started = true; }
        c->q_frames.push_back(frame_index);
    }
    c->q_cv.notify_one();
    return 0;
}
int rt_wait(rt_ctx* c) { if (!c) return -1; RT_SETTLE(c); return 0; }

int rt_reset_accum(rt_ctx* c)
{
    if (!c) return -1;
    RT_SETTLE(c);
    RT_HIP(c, hipSetDevice(c->device));
    if (c->target_pixels) {
        RT_HIP(c, hipMemsetAsync(c->d_accum.p, 0, c->target_pixels * sizeof(float4), c->stream));
        RT_HIP(c, hipMemsetAsync(c->d_frame.p, 0, c->target_pixels * sizeof(float4), c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->stats.numRenderedFrames = 0; c->stats.totalKernelMs = 0; c->stats.queuedLaunches = 0;
    return 0;
}

int rt_write_accum(rt_ctx* c, const float* rgba, size_t n_floats, int frames_rendered)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (!rgba && n_floats) return fail(c, -2, "null source");
    if (frames_rendered < 0) return fail(c, -2, "frames_rendered < 0");
    if (!c->have_params) return fail(c, -2, "rt_set_params has not been called");
    RT_HIP(c, hipSetDevice(c->device));
    { int r = ensure_targets(c); if (r) return r; }
    if (n_floats != c->target_pixels * 4) return fail(c, -2, "expected %zu floats (rows*width*4), got %zu", c->target_pixels * 4, n_floats);
    if (n_floats) {
        RT_HIP(c, hipMemcpyAsync(c->d_accum.p, rgba, n_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->stats.numRenderedFrames = frames_rendered;
    return 0;
}

int rt_read_accum(rt_ctx* c, float* rgba, size_t n) { if (!c) return -1; RT_SETTLE(c); return read_target(c, true, rgba, n, false); }
int rt_read_last_frame(rt_ctx* c, float* rgba, size_t n) { if (!c) return -1; RT_SETTLE(c); return read_target(c, false, rgba, n, false); }
int rt_copy_accum_to_device(rt_ctx* c, void* dst, size_t n) { if (!c) return -1; RT_SETTLE(c); return read_target(c, true, (float*)dst, n, true); }

int rt_read_display(rt_ctx* c, uint32_t* rgba8, size_t n_pixels)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (!rgba8 && n_pixels) return fail(c, -2, "null destination");
    if (c->have_params) { int r = ensure_targets(c); if (r) return r; }
    if (n_pixels != c->target_pixels) return fail(c, -2, "expected %zu pixels (rows*width), got %zu", c->target_pixels, n_pixels);
    if (!n_pixels) return 0;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, c->d_display.ensure(n_pixels));
    RT_HIP(c, hipEventRecord(c->evg0, c->stream));
    const int grid = (int)std::min<size_t>((n_pixels + 255) / 256, (size_t)c->n_cu * 8);
    hipLaunchKernelGGL(rtg::k_display_srgb8, dim3(grid), dim3(256), 0, c->stream, c->d_accum.p, c->d_display.p, n_pixels);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipEventRecord(c->evg1, c->stream));
    RT_HIP(c, hipMemcpyAsync(rgba8, c->d_display.p, n_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    RT_HIP(c, hipEventElapsedTime(&ms, c->evg0, c->evg1));
    c->stats.lastDisplayMs = ms;
    return 0;
}

int rt_read_bvh(rt_ctx* c, void* nodes_f32, void* nodes_f16, size_t n_nodes)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (c->scene_dirty) return fail(c, -2, "the scene has not been built yet (render a frame first)");
    if (n_nodes != c->n_nodes) return fail(c, -2, "expected %zu nodes, got %zu", c->n_nodes, n_nodes);
    if (!n_nodes) return 0;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (nodes_f32) RT_HIP(c, hipMemcpy(nodes_f32, c->d_nodes.p, n_nodes * 128, hipMemcpyDeviceToHost));
    if (nodes_f16) RT_HIP(c, hipMemcpy(nodes_f16, c->d_nodes_h.p, n_nodes * 128, hipMemcpyDeviceToHost));
    return 0;
}

int rt_get_stats(rt_ctx* c, rt_stats* out)
{
    if (!c) return -1;
    RT_SETTLE(c);
    if (!out) return fail(c, -2, "null stats");
    *out = c->stats;
    return 0;
}

} // extern "C"

// ---- several devices behind one handle: frames tile across the GPUs of a node -------------------------------------------
// The path shards into independent pixels (seeds use global pixel coordinates, RayTracing.shader:360-362; accumulation is per
// pixel), so every device holds the whole scene, renders the 8-row bands b with b % N == its rank for all frames, and one
// gather at the end of rt_multi_render brings the strips to the first device: N - 1 peer copies (xGMI point-to-point, each over
// its own link on a fully connected node) and a row scatter.  No other exchange exists on the path.
struct rt_multi {
    std::vector<rt_ctx*> ctx;
    std::string err;
    int width = 0, height = 0;
    bool have_params = false;
    DevBuf<float4> d_image, d_staging;          // on the first context's device: the assembled image, the incoming strips
    DevBuf<uint32_t> d_display;                 // ... and its sRGB8 form (rt_multi_read_display)
    std::vector<hipEvent_t> ev_strip;           // per context: its strip has arrived on the first device (recorded on the SOURCE context's stream)
    int max_rows = 0;
    double lastGatherMs = 0, lastSetupMs = 0;
    bool scene_dirty = true;                    // the first context holds an upload the others have not received yet
    std::vector<int> peer;                      // per context: 1 = the first context's device reads its memory directly (peer access on), 0 = staged by the runtime
};

namespace {

int mfail(rt_multi* m, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (m) m->err = buf; else g_create_error = buf;
    return code;
}

// HIP calls of the gather / distribution phases: the failure goes to the rt_multi's own error string (rt_multi_last_error)
#define M_HIP(m, expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return mfail(m, -100, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

template <class Fn> int for_each_ctx(rt_multi* m, const char* what, Fn f)
{
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        const int r = f(m->ctx[i]);
        if (r) return mfail(m, r, "%s on context %zu: %s", what, i, rt_last_error(m->ctx[i]));
    }
    return 0;
}

// rows of rank r (bands r, r + N, ...) from its strip to their places in the image
__global__ __launch_bounds__(256) void k_scatter_bands(const float4* __restrict__ strip, float4* __restrict__ image, int W, int H, int rank, int N)
{
    const size_t n = (size_t)W * 8;
    int local_band = blockIdx.y;
    const int y0 = (rank + local_band * N) * 8;
    if (y0 >= H) return;
    const int rows = min(8, H - y0);
    const float4* src = strip + (size_t)local_band * n;
    float4* dst = image + (size_t)y0 * W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)rows * W; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

} // namespace

extern "C" {

rt_multi* rt_multi_create(const int* devices, int n_devices)
{
    if (n_devices < 1 || !devices) { mfail(nullptr, -1, "rt_multi_create: no devices given"); return nullptr; }
    rt_multi* m = new rt_multi();
    for (int i = 0; i < n_devices; ++i) {
        rt_ctx* c = rt_create(devices[i]);
        if (!c) { const std::string e = g_create_error; rt_multi_destroy(m); g_create_error = e; return nullptr; }
        m->ctx.push_back(c);
    }
    // direct peer copies between the first device and the others where the hardware allows it; a refusal only means that the runtime
    // stages the copy (rt_multi_get_info reports which it is, per context)
    m->peer.assign(n_devices, 1);
    m->ev_strip.assign(n_devices, nullptr);
    for (int i = 1; i < n_devices; ++i) {
        (void)hipSetDevice(m->ctx[i]->device);
        if (hipEventCreateWithFlags(&m->ev_strip[i], hipEventDisableTiming) != hipSuccess) { mfail(nullptr, -1, "rt_multi_create: event for context %d", i); rt_multi_destroy(m); return nullptr; }
    }
    for (int i = 1; i < n_devices; ++i)
        if (m->ctx[i]->device != m->ctx[0]->device) {
            int ok = 1;
            for (int dir = 0; dir < 2; ++dir) {
                const int a = dir ? m->ctx[i]->device : m->ctx[0]->device, b = dir ? m->ctx[0]->device : m->ctx[i]->device;
                int can = 0;
                (void)hipSetDevice(a);
                if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { ok = 0; continue; }
                const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = 0;
            }
            (void)hipGetLastError();
            m->peer[i] = ok;
        }
    return m;
}

void rt_multi_destroy(rt_multi* m)
{
    if (!m) return;
    for (size_t i = 0; i < m->ev_strip.size() && i < m->ctx.size(); ++i)
        if (m->ev_strip[i]) { (void)hipSetDevice(m->ctx[i]->device); (void)hipEventDestroy(m->ev_strip[i]); }
    if (!m->ctx.empty()) { (void)hipSetDevice(m->ctx[0]->device); m->d_image.release(); m->d_staging.release(); m->d_display.release(); }
    for (rt_ctx* c : m->ctx) rt_destroy(c);
    delete m;
}

const char* rt_multi_last_error(const rt_multi* m) { return m ? m->err.c_str() : g_create_error.c_str(); }
int rt_multi_count(const rt_multi* m) { return m ? (int)m->ctx.size() : 0; }
rt_ctx* rt_multi_context(rt_multi* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

int rt_multi_set_params(rt_multi* m, const rt_params* p)
{
    if (!m) return -1;
    if (!p) return mfail(m, -2, "null params");
    const int N = (int)m->ctx.size();
    for (int i = 0; i < N; ++i) {
        int r = rt_set_params(m->ctx[i], p);
        if (!r) r = rt_set_bands(m->ctx[i], i, N);
        if (r) return mfail(m, r, "rt_set_params on context %d: %s", i, rt_last_error(m->ctx[i]));
    }
    m->width = p->width; m->height = p->height; m->have_params = true;
    return 0;
}
// The three buffers go to the FIRST context only: it builds the scene once (re-layout + BVH) at the next rt_multi_render and the other
// contexts receive the built scene device to device (clone_scene) — one build and one host -> device upload per scene change, whatever N.
static int multi_upload(rt_multi* m, int r, const char* what)
{
    if (r) return mfail(m, r, "%s on context 0: %s", what, rt_last_error(m->ctx[0]));
    m->scene_dirty = true;
    return 0;
}
int rt_multi_upload_spheres(rt_multi* m, const rt_sphere* s, int n)
{
    if (!m) return -1;
    // (a few records: every context takes them — with the geometry pipeline each context builds for itself, whichever upload comes first)
    const int r = for_each_ctx(m, "rt_upload_spheres", [&](rt_ctx* c) { return rt_upload_spheres(c, s, n); });
    if (!r && !m->ctx[0]->geom_local) m->scene_dirty = true;
    return r;
}
int rt_multi_upload_triangles(rt_multi* m, const rt_triangle* t, int n) { return m ? multi_upload(m, rt_upload_triangles(m->ctx[0], t, n), "rt_upload_triangles") : -1; }
int rt_multi_upload_meshinfo(rt_multi* m, const rt_meshinfo* mi, int n) { return m ? multi_upload(m, rt_upload_meshinfo(m->ctx[0], mi, n), "rt_upload_meshinfo") : -1; }
// On-device geometry pipeline behind the handle.  Every context receives the local meshes once and the poses (40 B per mesh) per frame and
// transforms, builds and refits ON ITS OWN DEVICE (the device build takes a few milliseconds): no geometry crosses xGMI per frame.
int rt_multi_upload_local_meshes(rt_multi* m, const rt_triangle* tris, int n_tris, const rt_local_chunk* chunks, int n_chunks, int n_meshes)
{
    if (!m) return -1;
    const int r = for_each_ctx(m, "rt_upload_local_meshes", [&](rt_ctx* c) { return rt_upload_local_meshes(c, tris, n_tris, chunks, n_chunks, n_meshes); });
    if (!r) m->scene_dirty = false;             // nothing to fan out: every context holds the upload itself
    return r;
}
int rt_multi_set_mesh_transforms(rt_multi* m, const rt_mesh_transform* xf, int n_meshes)
{
    return m ? for_each_ctx(m, "rt_set_mesh_transforms", [&](rt_ctx* c) { return rt_set_mesh_transforms(c, xf, n_meshes); }) : -1;
}
int rt_multi_set_option(rt_multi* m, const char* name, int value)
{
    if (!m) return -1;
    const int r = for_each_ctx(m, "rt_set_option", [&](rt_ctx* c) { return rt_set_option(c, name, value); });
    if (!r && m->ctx[0]->scene_dirty) m->scene_dirty = true;        // (a builder option: the first context rebuilds, the others receive the new tree)
    return r;
}
int rt_multi_reset_accum(rt_multi* m) { return m ? for_each_ctx(m, "rt_reset_accum", [&](rt_ctx* c) { return rt_reset_accum(c); }) : -1; }

int rt_multi_render(rt_multi* m, int first_frame, int n_frames)
{
    if (!m) return -1;
    if (!m->have_params) return mfail(m, -2, "rt_multi_set_params has not been called");
    const int N = (int)m->ctx.size();
    bool stale = m->scene_dirty;
    for (rt_ctx* c : m->ctx) stale = stale || c->scene_dirty;       // (a builder option set through rt_multi_context(i), a context never filled)
    if (m->ctx[0]->geom_local) {
        // geometry pipeline: every context holds the local meshes and builds / refits on its own device, inside its rt_render below
        for (rt_ctx* c : m->ctx) if (!c->geom_local) return mfail(m, -2, "rt_multi: some contexts hold local meshes and some do not (use the rt_multi_upload_* calls)");
        stale = false; m->scene_dirty = false;
    }
    if (stale) {
        // ---- scene change: the first context builds (one BVH build, one host -> device upload), the others receive the result
        const double t0 = now_ms();
        rt_ctx* root = m->ctx[0];
        { int r = launch_frames(root, first_frame, 0, Variant::Fast); if (r) return mfail(m, r, "scene build on context 0: %s", rt_last_error(root)); }
        std::vector<int> rcs(N, 0);
        {
            std::vector<std::thread> th;
            for (int i = 1; i < N; ++i) th.emplace_back([&, i]() { rcs[i] = clone_scene(m->ctx[i], root); });
            for (std::thread& t : th) t.join();
        }
        for (int i = 1; i < N; ++i) if (rcs[i]) return mfail(m, rcs[i], "scene transfer to context %d: %s", i, rt_last_error(m->ctx[i]));
        m->scene_dirty = false;
        m->lastSetupMs = now_ms() - t0;
    }
    // every device renders its bands for all frames, concurrently: one host thread per context (a context is single-threaded,
    // the contexts are independent)
    std::vector<int> rc(N, 0);
    {
        std::vector<std::thread> th;
        for (int i = 1; i < N; ++i) th.emplace_back([&, i]() { rc[i] = rt_render(m->ctx[i], first_frame, n_frames); });
        rc[0] = rt_render(m->ctx[0], first_frame, n_frames);
        for (std::thread& t : th) t.join();
    }
    for (int i = 0; i < N; ++i) if (rc[i]) return mfail(m, rc[i], "rt_render on context %d: %s", i, rt_last_error(m->ctx[i]));
    // ---- the one gather: strips -> first device, rows to their places
    rt_ctx* root = m->ctx[0];
    const int W = m->width, H = m->height;
    if ((size_t)W * H == 0) return 0;
    M_HIP(m, hipSetDevice(root->device));
    int max_rows = 0;
    for (rt_ctx* c : m->ctx) max_rows = std::max(max_rows, c->target_rows);
    M_HIP(m, m->d_image.ensure((size_t)W * H));
    M_HIP(m, m->d_staging.ensure((size_t)W * max_rows * (size_t)std::max(1, N - 1)));
    M_HIP(m, hipStreamSynchronize(root->stream));           // (the staging area is allocated and idle: the sources may write)
    const double tg0 = now_ms();
    // every strip travels on ITS source context's stream — N - 1 independent transfers, each over its own xGMI link on a fully connected
    // node, in flight together — and the first device's stream waits for the N - 1 arrival events before it scatters the rows
    for (int i = 1; i < N; ++i) {
        rt_ctx* c = m->ctx[i];
        if (c->target_pixels == 0) continue;
        float4* dst = m->d_staging.p + (size_t)(i - 1) * W * max_rows;
        M_HIP(m, hipSetDevice(c->device));
        if (c->device == root->device && !root->opt_peer_copies) M_HIP(m, hipMemcpyAsync(dst, c->d_accum.p, c->target_pixels * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
        else M_HIP(m, hipMemcpyPeerAsync(dst, root->device, c->d_accum.p, c->device, c->target_pixels * sizeof(float4), c->stream));
        M_HIP(m, hipEventRecord(m->ev_strip[i], c->stream));
    }
    M_HIP(m, hipSetDevice(root->device));
    for (int i = 1; i < N; ++i)
        if (m->ctx[i]->target_pixels != 0) M_HIP(m, hipStreamWaitEvent(root->stream, m->ev_strip[i], 0));
    for (int i = 0; i < N; ++i) {
        rt_ctx* c = m->ctx[i];
        if (c->target_pixels == 0) continue;
        const float4* src = i == 0 ? c->d_accum.p : m->d_staging.p + (size_t)(i - 1) * W * max_rows;
        const int bands = (c->target_rows + 7) / 8;
        hipLaunchKernelGGL(k_scatter_bands, dim3(std::max(1, std::min(64, (W * 8 + 255) / 256)), bands), dim3(256), 0, root->stream, src, m->d_image.p, W, H, i, N);
    }
    M_HIP(m, hipGetLastError());
    M_HIP(m, hipStreamSynchronize(root->stream));
    m->lastGatherMs = now_ms() - tg0;           // host wall time from the first copy's submission to the assembled image (the copies run on N - 1 streams)
    return 0;
}

// The display step for the assembled image (rt_read_display's twin): linear -> sRGB8 on the first device.
int rt_multi_read_display(rt_multi* m, uint32_t* rgba8, size_t n_pixels)
{
    if (!m) return -1;
    if (!rgba8 && n_pixels) return mfail(m, -2, "null destination");
    if (n_pixels != (size_t)m->width * m->height) return mfail(m, -2, "expected %zu pixels (height*width), got %zu", (size_t)m->width * m->height, n_pixels);
    if (!n_pixels) return 0;
    if (!m->d_image.p) return mfail(m, -2, "nothing rendered yet");
    rt_ctx* root = m->ctx[0];
    M_HIP(m, hipSetDevice(root->device));
    M_HIP(m, m->d_display.ensure(n_pixels));
    const int grid = (int)std::min<size_t>((n_pixels + 255) / 256, (size_t)root->n_cu * 8);
    hipLaunchKernelGGL(rtg::k_display_srgb8, dim3(grid), dim3(256), 0, root->stream, m->d_image.p, m->d_display.p, n_pixels);
    M_HIP(m, hipGetLastError());
    M_HIP(m, hipMemcpyAsync(rgba8, m->d_display.p, n_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, root->stream));
    M_HIP(m, hipStreamSynchronize(root->stream));
    return 0;
}

// Restore a saved accumulation state (rt_write_accum's twin): the whole image goes in, every context takes the rows of its bands.
int rt_multi_write_accum(rt_multi* m, const float* rgba, size_t n_floats, int frames_rendered)
{
    if (!m) return -1;
    if (!m->have_params) return mfail(m, -2, "rt_multi_set_params has not been called");
    if (!rgba && n_floats) return mfail(m, -2, "null source");
    const int W = m->width, H = m->height, N = (int)m->ctx.size();
    if (n_floats != (size_t)W * H * 4) return mfail(m, -2, "expected %zu floats (height*width*4), got %zu", (size_t)W * H * 4, n_floats);
    std::vector<float> strip;
    for (int i = 0; i < N; ++i) {
        strip.clear();
        for (int y0 = i * 8; y0 < H; y0 += N * 8) {
            const int rows = std::min(8, H - y0);
            strip.insert(strip.end(), rgba + (size_t)y0 * W * 4, rgba + (size_t)(y0 + rows) * W * 4);
        }
        const int r = rt_write_accum(m->ctx[i], strip.data(), strip.size(), frames_rendered);
        if (r) return mfail(m, r, "rt_write_accum on context %d: %s", i, rt_last_error(m->ctx[i]));
    }
    // the assembled image follows, so that rt_multi_read_accum / rt_multi_read_display show the restored state before the next render
    rt_ctx* root = m->ctx[0];
    M_HIP(m, hipSetDevice(root->device));
    M_HIP(m, m->d_image.ensure((size_t)W * H));
    if (n_floats) M_HIP(m, hipMemcpy(m->d_image.p, rgba, n_floats * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int rt_multi_read_accum(rt_multi* m, float* rgba, size_t n_floats)
{
    if (!m) return -1;
    if (!rgba && n_floats) return mfail(m, -2, "null destination");
    if (n_floats != (size_t)m->width * m->height * 4) return mfail(m, -2, "expected %zu floats (height*width*4), got %zu", (size_t)m->width * m->height * 4, n_floats);
    if (!n_floats) return 0;
    if (!m->d_image.p) return mfail(m, -2, "nothing rendered yet");
    rt_ctx* root = m->ctx[0];
    M_HIP(m, hipSetDevice(root->device));
    M_HIP(m, hipMemcpy(rgba, m->d_image.p, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int rt_multi_get_stats(rt_multi* m, rt_stats* out, double* gather_ms)
{
    if (!m) return -1;
    if (!out) return mfail(m, -2, "null stats");
    rt_stats sum = m->ctx[0]->stats;
    for (size_t i = 1; i < m->ctx.size(); ++i) {
        const rt_stats& s = m->ctx[i]->stats;
        sum.rays += s.rays; sum.sphereTests += s.sphereTests; sum.nodeVisits += s.nodeVisits; sum.triTests += s.triTests; sum.hits += s.hits;
        for (int k = 0; k < 5; ++k) { sum.phaseLanes[k] += s.phaseLanes[k]; sum.phaseExecs[k] += s.phaseExecs[k]; }
        for (int k = 0; k < rtk::kNumRegions; ++k) sum.regionExecs[k] += s.regionExecs[k];
        sum.lastKernelMs = std::max(sum.lastKernelMs, s.lastKernelMs); sum.totalKernelMs = std::max(sum.totalKernelMs, s.totalKernelMs);
    }
    *out = sum;
    if (gather_ms) *gather_ms = m->lastGatherMs;
    return 0;
}

int rt_multi_get_info(rt_multi* m, rt_multi_info* out)
{
    if (!m) return -1;
    if (!out) return mfail(m, -2, "null info");
    std::memset(out, 0, sizeof *out);
    out->numContexts = (int32_t)m->ctx.size();
    for (rt_ctx* c : m->ctx) out->bvhBuilds += c->stats.bvhBuilds;
    out->lastSetupMs = m->lastSetupMs; out->lastGatherMs = m->lastGatherMs;
    for (size_t i = 0; i < m->ctx.size() && i < 16; ++i) { out->device[i] = m->ctx[i]->device; out->peerAccess[i] = m->peer[i]; }
    return 0;
}

} // extern "C"
