// rt_host_c.cpp — a plain-C window onto the C++ host, used by the tests (ctypes) to compare its buffers with host.py's.
#include <cstring>
#include <string>

#include "rt_host.hpp"

namespace {
thread_local std::string g_err;
struct Handle { rthost::RayTracingManager mgr; rthost::SceneBuffers buf; bool built = false; };
}

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

void* rth_load_unity(const char* path, int width, int height)
{
    try {
        auto* h = new Handle();
        h->mgr = rthost::LoadUnityScene(path, width, height);
        return h;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

void rth_free(void* h) { delete static_cast<Handle*>(h); }

// counts[0..6] = spheres, triangles, chunks, meshes, serialised chunks, serialised triangles, maxBounceCount
int rth_build(void* hp, int* counts)
{
    auto* h = static_cast<Handle*>(hp);
    try {
        h->buf = h->mgr.BuildBuffers(); h->built = true;
        counts[0] = (int)h->buf.spheres.size(); counts[1] = (int)h->buf.triangles.size(); counts[2] = (int)h->buf.meshInfo.size();
        counts[3] = (int)h->mgr.meshes.size(); counts[4] = h->mgr.serialisedNumMeshChunks; counts[5] = h->mgr.serialisedNumTriangles;
        counts[6] = h->mgr.maxBounceCount;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

int rth_copy(void* hp, rt_params* params, rt_sphere* spheres, rt_triangle* tris, rt_meshinfo* infos)
{
    auto* h = static_cast<Handle*>(hp);
    if (!h->built) { g_err = "rth_build first"; return -1; }
    *params = h->buf.params;
    if (!h->buf.spheres.empty()) std::memcpy(spheres, h->buf.spheres.data(), h->buf.spheres.size() * sizeof(rt_sphere));
    if (!h->buf.triangles.empty()) std::memcpy(tris, h->buf.triangles.data(), h->buf.triangles.size() * sizeof(rt_triangle));
    if (!h->buf.meshInfo.empty()) std::memcpy(infos, h->buf.meshInfo.data(), h->buf.meshInfo.size() * sizeof(rt_meshinfo));
    return 0;
}

// Render `frames` frames of the loaded scene on HIP device `device` through the C++ manager; rgba = width*height*4 floats.
int rth_render(void* hp, int device, int frames, float* rgba)
{
    auto* h = static_cast<Handle*>(hp);
    rt_ctx* ctx = rt_create(device);
    if (!ctx) { g_err = rt_last_error(nullptr); return -1; }
    int rc = 0;
    try {
        std::vector<float> out;
        h->mgr.Start(ctx);
        h->mgr.OnRenderImage(ctx, frames, &out);
        std::memcpy(rgba, out.data(), out.size() * sizeof(float));
    } catch (const std::exception& e) { g_err = e.what(); rc = -1; }
    h->mgr.OnDisable();
    rt_destroy(ctx);
    return rc;
}

// The same scene through an rt_multi of n_contexts contexts (devices[i] each): RayTracingManager::OnRenderImage(rt_multi*, ...).
int rth_render_multi(void* hp, const int* devices, int n_contexts, int frames, float* rgba)
{
    auto* h = static_cast<Handle*>(hp);
    rt_multi* m = rt_multi_create(devices, n_contexts);
    if (!m) { g_err = rt_multi_last_error(nullptr); return -1; }
    int rc = 0;
    try {
        std::vector<float> out;
        rthost::RayTracingManager mgr = h->mgr;
        mgr.OnDisable();                                 // (a copy of the handle's manager: nothing uploaded to this rt_multi yet)
        mgr.Start(m);
        mgr.OnRenderImage(m, frames, &out);
        std::memcpy(rgba, out.data(), out.size() * sizeof(float));
    } catch (const std::exception& e) { g_err = e.what(); rc = -1; }
    rt_multi_destroy(m);
    return rc;
}

// Animated scene through the compiled manager: `steps` times { every mesh is turned by the quaternion q4 (x, y, z, w: rotation = q * rotation)
// and mesh i moves by shift3 * (i + 1); Start; OnRenderImage(frames_per_step) }, through one rt_ctx (n_contexts = 0) or an rt_multi of
// n_contexts contexts, with or without the on-device geometry pipeline.  rgba = the last step's resultTexture.  What the reference does
// every frame (RayTracedMesh.cs:36-84): with device_geometry only the poses travel.
int rth_render_animated(void* hp, const int* devices, int n_contexts, int device_geometry, int steps, int frames_per_step, const float* q4,
                        const float* shift3, float* rgba)
{
    auto* h = static_cast<Handle*>(hp);
    rt_ctx* ctx = nullptr; rt_multi* m = nullptr;
    if (n_contexts > 0) { m = rt_multi_create(devices, n_contexts); if (!m) { g_err = rt_multi_last_error(nullptr); return -1; } }
    else { ctx = rt_create(devices ? devices[0] : 0); if (!ctx) { g_err = rt_last_error(nullptr); return -1; } }
    int rc = 0;
    try {
        rthost::RayTracingManager mgr = h->mgr;
        mgr.OnDisable();
        mgr.deviceGeometry = device_geometry != 0;
        std::vector<float> out;
        for (int s = 0; s < steps; ++s) {
            for (size_t i = 0; i < mgr.meshes.size(); ++i) {
                const rthost::Quaternion q{ q4[0], q4[1], q4[2], q4[3] }, r = mgr.meshes[i].transform.rotation;
                rthost::Quaternion& t = mgr.meshes[i].transform.rotation;         // q * r (UnityEngine Quaternion product)
                t = { q.w * r.x + q.x * r.w + q.y * r.z - q.z * r.y, q.w * r.y + q.y * r.w + q.z * r.x - q.x * r.z,
                      q.w * r.z + q.z * r.w + q.x * r.y - q.y * r.x, q.w * r.w - q.x * r.x - q.y * r.y - q.z * r.z };
                const float k = (float)(i + 1);
                mgr.meshes[i].transform.position.x += shift3[0] * k; mgr.meshes[i].transform.position.y += shift3[1] * k; mgr.meshes[i].transform.position.z += shift3[2] * k;
            }
            if (m) { mgr.Start(m); mgr.OnRenderImage(m, frames_per_step, &out); }
            else { mgr.Start(ctx); mgr.OnRenderImage(ctx, frames_per_step, &out); }
        }
        std::memcpy(rgba, out.data(), out.size() * sizeof(float));
    } catch (const std::exception& e) { g_err = e.what(); rc = -1; }
    if (m) rt_multi_destroy(m);
    if (ctx) rt_destroy(ctx);
    return rc;
}

// ---- MeshSplitter / RayTracedMesh.GetSubMeshes through plain arrays (tests) --------------------------------------------
// verts / normals: n_verts x 3 floats; indices: the index buffer; sub_ranges: n_sub x (indexStart, indexCount).
// mode 0: MeshSplitter::CreateChunks(mesh) (local chunks);  mode 1: a RayTracedMesh without cached chunks and with that mesh as
// meshFilter.sharedMesh, posed by transform10 = position(3) rotation xyzw(4) lossyScale(3): GetSubMeshes() (world chunks);
// mode 2: Split(CreateSubMeshFromTriangles(the mesh's triangles in index order, seed, 0)) with seed = seed3 (the restatement tests
// search the seed vertex the reference must have had).  The chunks stay in a thread-local result until the next call.
namespace { thread_local std::vector<rthost::MeshChunk> g_chunks; }

int rth_split_mesh(const float* verts, const float* normals, int n_verts, const int* indices, int n_indices,
                   const int* sub_ranges, int n_sub, int mode, const float* transform10, const float* seed3, int enforce_limit)
{
    try {
        rthost::Mesh m;
        m.vertices.resize((size_t)n_verts); m.normals.resize((size_t)n_verts);
        for (int i = 0; i < n_verts; ++i) {
            m.vertices[(size_t)i] = { verts[3 * i], verts[3 * i + 1], verts[3 * i + 2] };
            m.normals[(size_t)i] = { normals[3 * i], normals[3 * i + 1], normals[3 * i + 2] };
        }
        m.triangles.assign(indices, indices + n_indices);
        for (int i = 0; i < n_sub; ++i) m.subMeshes.push_back({ sub_ranges[2 * i], sub_ranges[2 * i + 1] });
        if (mode == 0) g_chunks = rthost::MeshSplitter::CreateChunks(m);
        else if (mode == 1) {
            rthost::RayTracedMesh rm;
            rm.sharedMesh = &m; rm.enforceTriangleLimit = enforce_limit != 0;
            rm.transform.position = { transform10[0], transform10[1], transform10[2] };
            rm.transform.rotation = { transform10[3], transform10[4], transform10[5], transform10[6] };
            rm.transform.lossyScale = { transform10[7], transform10[8], transform10[9] };
            g_chunks = rm.GetSubMeshes();
        } else {
            rthost::MeshChunk whole = rthost::MeshSplitter::CreateSubMesh(m, 0, n_indices, 0);
            whole = rthost::MeshSplitter::CreateSubMeshFromTriangles(whole.triangles, { seed3[0], seed3[1], seed3[2] }, 0);
            g_chunks.clear();
            rthost::MeshSplitter::Split(whole, g_chunks);
        }
        return (int)g_chunks.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// tri_counts[n], sub_mesh[n], bounds[n*6] = centre xyz, size xyz; then rth_split_triangles copies all chunks' triangles back to back
void rth_split_info(int* tri_counts, int* sub_mesh, float* bounds)
{
    for (size_t i = 0; i < g_chunks.size(); ++i) {
        const rthost::MeshChunk& c = g_chunks[i];
        tri_counts[i] = (int)c.triangles.size(); sub_mesh[i] = c.subMeshIndex;
        bounds[6 * i] = c.bounds.center.x; bounds[6 * i + 1] = c.bounds.center.y; bounds[6 * i + 2] = c.bounds.center.z;
        bounds[6 * i + 3] = c.bounds.size.x; bounds[6 * i + 4] = c.bounds.size.y; bounds[6 * i + 5] = c.bounds.size.z;
    }
}
void rth_split_triangles(rt_triangle* out)
{
    for (const rthost::MeshChunk& c : g_chunks) {
        if (!c.triangles.empty()) std::memcpy(out, c.triangles.data(), c.triangles.size() * sizeof(rt_triangle));
        out += c.triangles.size();
    }
}

} // extern "C"
