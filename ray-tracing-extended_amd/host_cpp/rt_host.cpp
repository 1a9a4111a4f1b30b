// rt_host.cpp — C++ host components (see rt_host.hpp).  Arithmetic mirrors host.py operation for operation.
#include "rt_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rthost {

Vector3 operator*(const Quaternion& q, const Vector3& p)
{
    const float x = q.x, y = q.y, z = q.z, w = q.w;
    const float x2 = x * 2.0f, y2 = y * 2.0f, z2 = z * 2.0f;
    const float xx = x * x2, yy = y * y2, zz = z * z2;
    const float xy = x * y2, xz = x * z2, yz = y * z2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    Vector3 r;
    r.x = ((1.0f - (yy + zz)) * p.x + (xy - wz) * p.y) + (xz + wy) * p.z;
    r.y = ((xy + wz) * p.x + (1.0f - (xx + zz)) * p.y) + (yz - wx) * p.z;
    r.z = ((xz - wy) * p.x + (yz + wx) * p.y) + (1.0f - (xx + yy)) * p.z;
    return r;
}

Quaternion operator*(const Quaternion& a, const Quaternion& b)
{
    Quaternion r;
    r.x = ((a.w * b.x + a.x * b.w) + a.y * b.z) - a.z * b.y;
    r.y = ((a.w * b.y + a.y * b.w) + a.z * b.x) - a.x * b.z;
    r.z = ((a.w * b.z + a.z * b.w) + a.x * b.y) - a.y * b.x;
    r.w = ((a.w * b.w - a.x * b.x) - a.y * b.y) - a.z * b.z;
    return r;
}

void Transform::localToWorldMatrix(float m[16]) const
{
    const float x = rotation.x, y = rotation.y, z = rotation.z, w = rotation.w;
    const float r[3][3] = {
        { 1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - z * w), 2.0f * (x * z + y * w) },
        { 2.0f * (x * y + z * w), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - x * w) },
        { 2.0f * (x * z - y * w), 2.0f * (y * z + x * w), 1.0f - 2.0f * (x * x + y * y) } };
    const float s[3] = { lossyScale.x, lossyScale.y, lossyScale.z };
    const float t[3] = { position.x, position.y, position.z };
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) m[4 * i + j] = r[i][j] * s[j];
        m[4 * i + 3] = t[i];
    }
    m[12] = m[13] = m[14] = 0.0f; m[15] = 1.0f;
}

Vector3 Light::worldSpaceLightPos0() const
{
    Vector3 f = rotation * Vector3{ 0, 0, 1 };
    return { -f.x, -f.y, -f.z };
}

// ---- MeshSplitter (Assets/Scripts/Helpers/MeshSplitter.cs) --------------------------------------------------------------
namespace MeshSplitter {
namespace {

struct UBounds {            // UnityEngine.Bounds: centre + extents, float32
    float c[3], e[3];
    UBounds(const float* center, const float* size) { for (int a = 0; a < 3; ++a) { c[a] = center[a]; e[a] = size[a] * 0.5f; } }
    float mn(int a) const { return c[a] - e[a]; }
    float mx(int a) const { return c[a] + e[a]; }
    float size(int a) const { return e[a] * 2.0f; }
    void Encapsulate(const float* p)            // SetMinMax(Vector3.Min(min, p), Vector3.Max(max, p))
    {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(mn(a), p[a]); hi[a] = std::max(mx(a), p[a]); }
        for (int a = 0; a < 3; ++a) { e[a] = (hi[a] - lo[a]) * 0.5f; c[a] = lo[a] + e[a]; }
    }
    bool Contains(const float* p) const
    {
        for (int a = 0; a < 3; ++a) if (!(p[a] >= mn(a) && p[a] <= mx(a))) return false;
        return true;
    }
    Bounds toBounds() const { return { { c[0], c[1], c[2] }, { size(0), size(1), size(2) } }; }
};

UBounds fromBounds(const Bounds& b)
{
    const float c[3] = { b.center.x, b.center.y, b.center.z }, s[3] = { b.size.x, b.size.y, b.size.z };
    return UBounds(c, s);
}

MeshChunk Extract(const std::vector<rt_triangle>& triangles, std::vector<char>& taken, const UBounds& splitBounds, int subMeshIndex)   // :101-124
{
    MeshChunk out; out.subMeshIndex = subMeshIndex;
    const float sz[3] = { splitBounds.size(0), splitBounds.size(1), splitBounds.size(2) };
    UBounds nb(splitBounds.c, sz);
    for (size_t i = 0; i < triangles.size(); ++i) {
        if (taken[i]) continue;
        const rt_triangle& t = triangles[i];
        if (splitBounds.Contains(t.posA) || splitBounds.Contains(t.posB) || splitBounds.Contains(t.posC)) {
            nb.Encapsulate(t.posA); nb.Encapsulate(t.posB); nb.Encapsulate(t.posC);
            out.triangles.push_back(t);
            taken[i] = 1;
        }
    }
    out.bounds = nb.toBounds();
    return out;
}

} // namespace

MeshChunk CreateSubMeshFromTriangles(const std::vector<rt_triangle>& triangles, const Vector3& firstVertex, int subMeshIndex)
{
    const float v0[3] = { firstVertex.x, firstVertex.y, firstVertex.z }, seed[3] = { 0.01f, 0.01f, 0.01f };
    UBounds b(v0, seed);                                                         // new Bounds(verts[indices[0]], Vector3.one * 0.01f)  :39
    for (const rt_triangle& t : triangles) { b.Encapsulate(t.posA); b.Encapsulate(t.posB); b.Encapsulate(t.posC); }
    MeshChunk c; c.triangles = triangles; c.bounds = b.toBounds(); c.subMeshIndex = subMeshIndex;
    return c;
}

MeshChunk CreateSubMesh(const Mesh& mesh, int indexStart, int indexCount, int subMeshIndex)
{
    if (indexStart < 0 || indexCount < 0 || (size_t)indexStart + (size_t)indexCount > mesh.triangles.size())
        throw std::runtime_error("sub-mesh index range outside the index buffer");
    if (indexCount < 3) throw std::runtime_error("sub-mesh without triangles");          // verts[indices[0]] would throw in C# too
    std::vector<rt_triangle> tris((size_t)indexCount / 3);
    auto vert = [&](int i) -> const Vector3& {
        if (i < 0 || (size_t)i >= mesh.vertices.size() || (size_t)i >= mesh.normals.size()) throw std::runtime_error("vertex index outside the mesh");
        return mesh.vertices[(size_t)i];
    };
    for (int i = 0; i + 2 < indexCount; i += 3) {
        float* const f = reinterpret_cast<float*>(&tris[(size_t)i / 3]);          // posA posB posC normalA normalB normalC
        for (int k = 0; k < 3; ++k) {
            const int idx = mesh.triangles[(size_t)indexStart + (size_t)i + (size_t)k];
            const Vector3& p = vert(idx); const Vector3& n = mesh.normals[(size_t)idx];
            f[3 * k] = p.x; f[3 * k + 1] = p.y; f[3 * k + 2] = p.z;
            f[9 + 3 * k] = n.x; f[9 + 3 * k + 1] = n.y; f[9 + 3 * k + 2] = n.z;
        }
    }
    return CreateSubMeshFromTriangles(tris, vert(mesh.triangles[(size_t)indexStart]), subMeshIndex);
}

void Split(const MeshChunk& chunk, std::vector<MeshChunk>& splitChunks, int depth)
{
    if ((int)chunk.triangles.size() > maxTrisPerChunk && depth < maxDepth) {
        const UBounds b = fromBounds(chunk.bounds);
        const float q[3] = { b.size(0) / 4.0f, b.size(1) / 4.0f, b.size(2) / 4.0f };
        std::vector<char> taken(chunk.triangles.size(), 0);
        size_t nTaken = 0;
        for (int x = -1; x <= 1; x += 2)
            for (int y = -1; y <= 1; y += 2)
                for (int z = -1; z <= 1; z += 2) {
                    if (chunk.triangles.size() - nTaken == 0) continue;
                    const float centre[3] = { b.c[0] + q[0] * (float)x, b.c[1] + q[1] * (float)y, b.c[2] + q[2] * (float)z };
                    const float size[3] = { q[0] * 2.0f, q[1] * 2.0f, q[2] * 2.0f };
                    const UBounds splitBounds(centre, size);
                    MeshChunk sub = Extract(chunk.triangles, taken, splitBounds, chunk.subMeshIndex);
                    nTaken += sub.triangles.size();
                    if (!sub.triangles.empty()) Split(sub, splitChunks, depth + 1);
                }
    } else {
        splitChunks.push_back(chunk);
    }
}

std::vector<MeshChunk> CreateChunks(const Mesh& mesh)
{
    std::vector<MeshChunk> subMeshes;
    for (size_t i = 0; i < mesh.subMeshes.size(); ++i)
        subMeshes.push_back(CreateSubMesh(mesh, mesh.subMeshes[i].indexStart, mesh.subMeshes[i].indexCount, (int)i));
    std::vector<MeshChunk> out;
    for (const MeshChunk& sm : subMeshes) Split(sm, out);
    return out;
}

} // namespace MeshSplitter

std::vector<MeshChunk> RayTracedMesh::GetSubMeshes()
{
    const int meshTriangles = mesh ? (int)(mesh->triangles.size() / 3) : triangleCount;     // mesh.triangles.Length / 3  (:19)
    if (enforceTriangleLimit && meshTriangles > RayTracingManager::TriangleLimit)
        throw std::runtime_error("Please use a mesh with fewer than " + std::to_string(RayTracingManager::TriangleLimit) + " triangles");
    // Split mesh into chunks (if result is not already cached)  :24-29
    if (sharedMesh != nullptr && (mesh != sharedMesh || localChunks.empty())) {
        mesh = sharedMesh;
        localChunks = MeshSplitter::CreateChunks(*mesh);
        triangleCount = (int)(mesh->triangles.size() / 3);
    }
    std::vector<MeshChunk> world(localChunks.size());
    const Vector3 pos = transform.position, scale = transform.lossyScale;
    const Quaternion rot = transform.rotation;
    for (size_t c = 0; c < localChunks.size(); ++c) {                            // UpdateWorldChunkFromLocal :56-84
        const MeshChunk& lc = localChunks[c];
        MeshChunk& wc = world[c];
        wc.triangles.resize(lc.triangles.size());
        wc.subMeshIndex = lc.subMeshIndex;
        float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (size_t i = 0; i < lc.triangles.size(); ++i) {
            const float* s = lc.triangles[i].posA;
            float* d = wc.triangles[i].posA;
            for (int k = 0; k < 3; ++k) {                                        // PointLocalToWorld :86-89
                Vector3 r = rot * Vector3{ s[3 * k] * scale.x, s[3 * k + 1] * scale.y, s[3 * k + 2] * scale.z };
                d[3 * k] = r.x + pos.x; d[3 * k + 1] = r.y + pos.y; d[3 * k + 2] = r.z + pos.z;
                for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], d[3 * k + a]); mx[a] = std::max(mx[a], d[3 * k + a]); }
            }
            for (int k = 3; k < 6; ++k) {                                        // DirectionLocalToWorld :91-94
                Vector3 r = rot * Vector3{ s[3 * k], s[3 * k + 1], s[3 * k + 2] };
                d[3 * k] = r.x; d[3 * k + 1] = r.y; d[3 * k + 2] = r.z;
            }
        }
        wc.bounds.center = { (mn[0] + mx[0]) / 2.0f, (mn[1] + mx[1]) / 2.0f, (mn[2] + mx[2]) / 2.0f };   // :82
        wc.bounds.size = { mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2] };
    }
    return world;
}

const RayTracingMaterial& RayTracedMesh::GetMaterial(int subMeshIndex) const
{
    return materials[std::min<size_t>((size_t)std::max(subMeshIndex, 0), materials.size() - 1)];
}

void RayTracingManager::OnValidate()
{
    maxBounceCount = std::max(0, maxBounceCount);
    numRaysPerPixel = std::max(1, numRaysPerPixel);
    environmentSettings.sunFocus = std::max(1.0, environmentSettings.sunFocus);
    environmentSettings.sunIntensity = std::max(0.0, environmentSettings.sunIntensity);
}

void RayTracingManager::UpdateCameraParams(rt_params& p) const
{
    const float deg2rad = 0.017453292f;
    const float half = (float)camera.fieldOfView * 0.5f * deg2rad;
    const float planeHeight = (float)focusDistance * (float)std::tan((double)half) * 2.0f;
    const float planeWidth = planeHeight * (float)camera.aspect;
    p.viewParams[0] = planeWidth; p.viewParams[1] = planeHeight; p.viewParams[2] = (float)focusDistance;
    camera.transform.localToWorldMatrix(p.camLocalToWorld);
    p.worldSpaceCameraPos[0] = camera.transform.position.x; p.worldSpaceCameraPos[1] = camera.transform.position.y;
    p.worldSpaceCameraPos[2] = camera.transform.position.z;
    const Vector3 l = light.worldSpaceLightPos0();
    p.worldSpaceLightPos0[0] = l.x; p.worldSpaceLightPos0[1] = l.y; p.worldSpaceLightPos0[2] = l.z;
}

static double gamma_to_linear(double c)
{
    if (c <= 0.04045) return c / 12.92;
    if (c < 1.0) return std::pow((c + 0.055) / 1.055, 2.4);
    return std::pow(c, 2.2);
}

void RayTracingManager::SetShaderParams(rt_params& p) const
{
    p.maxBounceCount = maxBounceCount; p.numRaysPerPixel = numRaysPerPixel;
    p.defocusStrength = (float)defocusStrength; p.divergeStrength = (float)divergeStrength;
    const EnvironmentSettings& e = environmentSettings;
    p.environmentEnabled = e.enabled ? 1 : 0;
    const double* src[3] = { e.groundColour, e.skyColourHorizon, e.skyColourZenith };
    float* dst[3] = { p.groundColour, p.skyColourHorizon, p.skyColourZenith };
    for (int k = 0; k < 3; ++k) {
        for (int ch = 0; ch < 3; ++ch) dst[k][ch] = (float)(linearColourSpace ? gamma_to_linear(src[k][ch]) : src[k][ch]);
        dst[k][3] = (float)src[k][3];
    }
    p.sunFocus = (float)e.sunFocus; p.sunIntensity = (float)e.sunIntensity;
}

std::vector<rt_sphere> RayTracingManager::CreateSpheres() const
{
    std::vector<rt_sphere> out(spheres.size());
    for (size_t i = 0; i < spheres.size(); ++i) {
        const RayTracedSphere& s = spheres[i];
        out[i].position[0] = s.transform.position.x; out[i].position[1] = s.transform.position.y; out[i].position[2] = s.transform.position.z;
        out[i].radius = s.transform.localScale.x * 0.5f;                         // :178
        out[i].material = s.material;
    }
    return out;
}

void RayTracingManager::CreateMeshes(std::vector<rt_triangle>& tris, std::vector<rt_meshinfo>& infos)
{
    tris.clear(); infos.clear();
    for (RayTracedMesh& mesh : meshes) {
        for (const MeshChunk& chunk : mesh.GetSubMeshes()) {
            rt_meshinfo mi{};
            mi.firstTriangleIndex = (uint32_t)tris.size(); mi.numTriangles = (uint32_t)chunk.triangles.size();
            mi.material = mesh.GetMaterial(chunk.subMeshIndex);
            const Vector3 lo = chunk.bounds.min(), hi = chunk.bounds.max();      // MeshInfo.cs:16-17
            mi.boundsMin[0] = lo.x; mi.boundsMin[1] = lo.y; mi.boundsMin[2] = lo.z;
            mi.boundsMax[0] = hi.x; mi.boundsMax[1] = hi.y; mi.boundsMax[2] = hi.z;
            infos.push_back(mi);
            tris.insert(tris.end(), chunk.triangles.begin(), chunk.triangles.end());
        }
    }
    numMeshChunks = (int)infos.size(); numTriangles = (int)tris.size();
}

SceneBuffers RayTracingManager::BuildBuffers()
{
    SceneBuffers b;
    std::memset(&b.params, 0, sizeof b.params);
    b.params.width = width; b.params.height = height;
    b.params.intersectMode = intersectMode; b.params.rngMode = RT_RNG_PCG;
    UpdateCameraParams(b.params);
    b.spheres = CreateSpheres();
    CreateMeshes(b.triangles, b.meshInfo);
    SetShaderParams(b.params);
    return b;
}

static void check(rt_ctx* ctx, int rc, const char* what)
{
    if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + rt_last_error(ctx));
}
static void mcheck(rt_multi* m, int rc, const char* what)
{
    if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + rt_multi_last_error(m));
}

void RayTracingManager::CreateLocalMeshes(std::vector<rt_triangle>& tris, std::vector<rt_local_chunk>& chunks)
{
    tris.clear(); chunks.clear();
    for (size_t m = 0; m < meshes.size(); ++m) {
        RayTracedMesh& mesh = meshes[m];
        (void)mesh.GetSubMeshes();              // splits the mesh when there is no cached result for it (RayTracedMesh.cs:24-29) and checks the limit
        for (const MeshChunk& chunk : mesh.localChunks) {
            rt_local_chunk c{};
            c.firstTriangleIndex = (uint32_t)tris.size(); c.numTriangles = (uint32_t)chunk.triangles.size(); c.meshIndex = (uint32_t)m;
            c.material = mesh.GetMaterial(chunk.subMeshIndex);
            chunks.push_back(c);
            tris.insert(tris.end(), chunk.triangles.begin(), chunk.triangles.end());
        }
    }
    numMeshChunks = (int)chunks.size(); numTriangles = (int)tris.size();
}

std::vector<rt_mesh_transform> RayTracingManager::CreateTransforms() const
{
    std::vector<rt_mesh_transform> out(meshes.size());
    for (size_t m = 0; m < meshes.size(); ++m) {
        const Transform& t = meshes[m].transform;                                // RayTracedMesh.cs:38-40
        out[m].position[0] = t.position.x; out[m].position[1] = t.position.y; out[m].position[2] = t.position.z;
        out[m].rotation[0] = t.rotation.x; out[m].rotation[1] = t.rotation.y; out[m].rotation[2] = t.rotation.z; out[m].rotation[3] = t.rotation.w;
        out[m].lossyScale[0] = t.lossyScale.x; out[m].lossyScale[1] = t.lossyScale.y; out[m].lossyScale[2] = t.lossyScale.z;
    }
    return out;
}

namespace {
template <class T> bool same_bytes(const std::vector<T>& a, const std::vector<T>& b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}
// the two handle kinds behind one InitFrame
struct CtxApi {
    void params(rt_ctx* c, const rt_params* p) const { check(c, rt_set_params(c, p), "rt_set_params"); }
    void spheres(rt_ctx* c, const rt_sphere* s, int n) const { check(c, rt_upload_spheres(c, s, n), "rt_upload_spheres"); }
    void triangles(rt_ctx* c, const rt_triangle* t, int n) const { check(c, rt_upload_triangles(c, t, n), "rt_upload_triangles"); }
    void meshinfo(rt_ctx* c, const rt_meshinfo* m, int n) const { check(c, rt_upload_meshinfo(c, m, n), "rt_upload_meshinfo"); }
    void local(rt_ctx* c, const rt_triangle* t, int nt, const rt_local_chunk* ch, int nc, int nm) const { check(c, rt_upload_local_meshes(c, t, nt, ch, nc, nm), "rt_upload_local_meshes"); }
    void poses(rt_ctx* c, const rt_mesh_transform* x, int n) const { check(c, rt_set_mesh_transforms(c, x, n), "rt_set_mesh_transforms"); }
};
struct MultiApi {
    void params(rt_multi* m, const rt_params* p) const { mcheck(m, rt_multi_set_params(m, p), "rt_multi_set_params"); }
    void spheres(rt_multi* m, const rt_sphere* s, int n) const { mcheck(m, rt_multi_upload_spheres(m, s, n), "rt_multi_upload_spheres"); }
    void triangles(rt_multi* m, const rt_triangle* t, int n) const { mcheck(m, rt_multi_upload_triangles(m, t, n), "rt_multi_upload_triangles"); }
    void meshinfo(rt_multi* m, const rt_meshinfo* mi, int n) const { mcheck(m, rt_multi_upload_meshinfo(m, mi, n), "rt_multi_upload_meshinfo"); }
    void local(rt_multi* m, const rt_triangle* t, int nt, const rt_local_chunk* ch, int nc, int nm) const { mcheck(m, rt_multi_upload_local_meshes(m, t, nt, ch, nc, nm), "rt_multi_upload_local_meshes"); }
    void poses(rt_multi* m, const rt_mesh_transform* x, int n) const { mcheck(m, rt_multi_set_mesh_transforms(m, x, n), "rt_multi_set_mesh_transforms"); }
};
} // namespace

// InitFrame (RayTracingManager.cs:95-109).  The reference rebuilds and re-uploads its three buffers every frame (its own TODO at
// RayTracedMesh.cs:37); an upload makes the library rebuild its acceleration structure, so only what changed is sent again — and with
// deviceGeometry nothing but the poses is sent per frame.
template <class H, class Api> void RayTracingManager::InitFrameT(H* h, const Api& api)
{
    const bool first = uploaded_to_ != h;
    SceneBuffers b;
    std::memset(&b.params, 0, sizeof b.params);
    b.params.width = width; b.params.height = height;
    b.params.intersectMode = intersectMode; b.params.rngMode = RT_RNG_PCG;
    UpdateCameraParams(b.params);
    b.spheres = CreateSpheres();
    SetShaderParams(b.params);
    api.params(h, &b.params);
    if (first || !same_bytes(b.spheres, sent_.spheres)) api.spheres(h, b.spheres.data(), (int)b.spheres.size());
    if (deviceGeometry) {
        std::vector<rt_triangle> lt; std::vector<rt_local_chunk> lc;
        CreateLocalMeshes(lt, lc);
        if (first || !sent_is_local_ || !same_bytes(lt, sent_local_) || !same_bytes(lc, sent_chunks_)) {
            api.local(h, lt.data(), (int)lt.size(), lc.data(), (int)lc.size(), (int)meshes.size());
            sent_local_.swap(lt); sent_chunks_.swap(lc); sent_is_local_ = true;
        }
        const std::vector<rt_mesh_transform> xf = CreateTransforms();
        api.poses(h, xf.data(), (int)xf.size());
    } else {
        CreateMeshes(b.triangles, b.meshInfo);
        if (first || sent_is_local_ || !same_bytes(b.triangles, sent_.triangles)) api.triangles(h, b.triangles.data(), (int)b.triangles.size());
        if (first || sent_is_local_ || !same_bytes(b.meshInfo, sent_.meshInfo)) api.meshinfo(h, b.meshInfo.data(), (int)b.meshInfo.size());
        sent_.triangles.swap(b.triangles); sent_.meshInfo.swap(b.meshInfo); sent_is_local_ = false;
    }
    sent_.spheres.swap(b.spheres);
    uploaded_to_ = h;
}

void RayTracingManager::InitFrame(rt_ctx* ctx) { InitFrameT(ctx, CtxApi{}); }

void RayTracingManager::Start(rt_ctx* ctx)
{
    numRenderedFrames = 0;
    check(ctx, rt_reset_accum(ctx), "rt_reset_accum");
}

void RayTracingManager::OnRenderImage(rt_ctx* ctx, int frames, std::vector<float>* resultTexture)
{
    InitFrame(ctx);
    check(ctx, rt_render(ctx, numRenderedFrames, frames), "rt_render");
    numRenderedFrames += frames;
    if (resultTexture) {
        resultTexture->resize((size_t)width * height * 4);
        check(ctx, rt_read_accum(ctx, resultTexture->data(), resultTexture->size()), "rt_read_accum");
    }
}

void RayTracingManager::InitFrame(rt_multi* m) { InitFrameT(m, MultiApi{}); }

void RayTracingManager::Start(rt_multi* m)
{
    numRenderedFrames = 0;
    mcheck(m, rt_multi_reset_accum(m), "rt_multi_reset_accum");
}

void RayTracingManager::OnRenderImage(rt_multi* m, int frames, std::vector<float>* resultTexture)
{
    InitFrame(m);
    mcheck(m, rt_multi_render(m, numRenderedFrames, frames), "rt_multi_render");
    numRenderedFrames += frames;
    if (resultTexture) {
        resultTexture->resize((size_t)width * height * 4);
        mcheck(m, rt_multi_read_accum(m, resultTexture->data(), resultTexture->size()), "rt_multi_read_accum");
    }
}

} // namespace rthost
