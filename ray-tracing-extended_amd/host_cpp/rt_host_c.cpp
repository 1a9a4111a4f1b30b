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
    rt_destroy(ctx);
    return rc;
}

} // extern "C"
