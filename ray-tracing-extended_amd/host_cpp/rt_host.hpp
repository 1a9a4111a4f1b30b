// rt_host.hpp — C++ host above the C-ABI: the reference's scene-description components, same names and behaviour.
//
//   RayTracingMaterial   Assets/Scripts/Data Types/RayTracingMaterial.cs:4-29       (= rt_material, 64 B)
//   EnvironmentSettings  Assets/Scripts/Data Types/EnvironmentSettings.cs:4-11
//   MeshChunk            Assets/Scripts/Data Types/MeshChunk.cs:6-17
//   MeshSplitter         Assets/Scripts/Helpers/MeshSplitter.cs:8-124               (CreateChunks, CreateSubMesh, Split, Extract)
//   RayTracedSphere      "Assets/Scripts/Render Types/RayTracedSphere.cs":5-7
//   RayTracedMesh        "Assets/Scripts/Render Types/RayTracedMesh.cs":17-99       (GetSubMeshes, GetMaterial)
//   RayTracingManager    Assets/Scripts/RayTracingManager.cs:11-203                  (settings, CreateSpheres, CreateMeshes,
//                                                                                     UpdateCameraParams, SetShaderParams,
//                                                                                     InitFrame, OnRenderImage, OnValidate)
// plus LoadUnityScene(): the reference's .unity files load unchanged (unity_loader.cpp).
// The reference's host is C# on UnityEngine; neither toolchain exists in the build image, so this is the compiled host
// (INTEGRATION.md shows the C# P/Invoke binding for a Unity build).  All marshal arithmetic is float32 in the operation
// order of the C# / UnityEngine expressions; it produces byte-identical buffers to the Python mirror (host.py), tested.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt.h"

namespace rthost {

struct Vector3 { float x = 0, y = 0, z = 0; };
struct Quaternion { float x = 0, y = 0, z = 0, w = 1; };
using RayTracingMaterial = rt_material;
enum MaterialFlag { None = 0, CheckerPattern = 1, InvisibleLight = 2 };       // RayTracingMaterial.cs:6-11

Vector3 operator*(const Quaternion& q, const Vector3& p);                       // UnityEngine Quaternion * Vector3
Quaternion operator*(const Quaternion& a, const Quaternion& b);

struct Transform {                  // what the reference reads: position / rotation / lossyScale / localScale
    Vector3 position; Quaternion rotation; Vector3 lossyScale{1, 1, 1}; Vector3 localScale{1, 1, 1};
    void localToWorldMatrix(float m[16]) const;                                  // Matrix4x4.TRS, row-major
};

struct Bounds {                     // UnityEngine.Bounds as constructed with (center, size)
    Vector3 center, size;
    Vector3 min() const { return { center.x - size.x * 0.5f, center.y - size.y * 0.5f, center.z - size.z * 0.5f }; }
    Vector3 max() const { return { center.x + size.x * 0.5f, center.y + size.y * 0.5f, center.z + size.z * 0.5f }; }
};

struct MeshChunk { std::vector<rt_triangle> triangles; Bounds bounds; int subMeshIndex = 0; };

// What MeshSplitter reads of a UnityEngine.Mesh: vertices, normals, the index buffer and the sub-mesh ranges
// (mesh.vertices / normals / triangles / GetSubMesh(i).indexStart, indexCount — MeshSplitter.cs:15-23).
struct Mesh {
    struct SubMeshDescriptor { int indexStart = 0, indexCount = 0; };
    std::vector<Vector3> vertices, normals;
    std::vector<int> triangles;
    std::vector<SubMeshDescriptor> subMeshes;
};

// MeshSplitter.cs: recursive 8-octant split until <= 48 triangles or depth 6; a triangle goes to the first octant (x, then y,
// then z loop order) that contains one of its vertices.  UnityEngine.Bounds is restated in float32 as centre + extents
// (Encapsulate(p) = SetMinMax(Min(min, p), Max(max, p)), re-deriving centre and extents after every point).
namespace MeshSplitter {
constexpr int maxDepth = 6;             // MeshSplitter.cs:8
constexpr int maxTrisPerChunk = 48;     // MeshSplitter.cs:9
std::vector<MeshChunk> CreateChunks(const Mesh& mesh);                                                       // :11-33
MeshChunk CreateSubMesh(const Mesh& mesh, int indexStart, int indexCount, int subMeshIndex);                 // :35-63
// the same from an explicit triangle list and seed vertex (the 0.01-sized box of :39 sits at the sub-mesh's first vertex)
MeshChunk CreateSubMeshFromTriangles(const std::vector<rt_triangle>& triangles, const Vector3& firstVertex, int subMeshIndex);
void Split(const MeshChunk& chunk, std::vector<MeshChunk>& splitChunks, int depth = 0);                      // :65-99
}

struct EnvironmentSettings {
    bool enabled = false;       // colours are kept as parsed (double): Material.SetColor's sRGB -> linear runs on them
    double groundColour[4] = {0, 0, 0, 0}, skyColourHorizon[4] = {0, 0, 0, 0}, skyColourZenith[4] = {0, 0, 0, 0};
    double sunFocus = 1, sunIntensity = 0;
};

struct Camera { Transform transform; double fieldOfView = 60, aspect = 16.0 / 9.0; };
struct Light { Quaternion rotation; Vector3 worldSpaceLightPos0() const; };     // directional: -forward

struct RayTracedSphere { Transform transform; RayTracingMaterial material{}; };

struct RayTracedMesh {
    Transform transform;
    std::vector<RayTracingMaterial> materials;
    std::vector<MeshChunk> localChunks;         // [SerializeField]: the scenes carry them; empty = not split yet
    const Mesh* mesh = nullptr;                 // [SerializeField] Mesh mesh: the mesh the cached chunks were made from
    const Mesh* sharedMesh = nullptr;           // meshFilter.sharedMesh (null: no MeshFilter — only the serialised chunks exist)
    int triangleCount = 0;
    bool enforceTriangleLimit = true;
    // RayTracedMesh.cs:17-54: throws above 1500 triangles; splits the mesh when there is no cached result for it (:24-29);
    // returns the chunks in world space
    std::vector<MeshChunk> GetSubMeshes();
    const RayTracingMaterial& GetMaterial(int subMeshIndex) const;               // :96-99
};

struct SceneBuffers {               // what InitFrame hands to the device (the reference's three structured buffers)
    rt_params params{};
    std::vector<rt_sphere> spheres;
    std::vector<rt_triangle> triangles;
    std::vector<rt_meshinfo> meshInfo;
};

class RayTracingManager {
public:
    static constexpr int TriangleLimit = 1500;                                   // RayTracingManager.cs:9
    // settings (defaults of RayTracingManager.cs:12-17)
    int maxBounceCount = 4, numRaysPerPixel = 2;
    double defocusStrength = 0, divergeStrength = 0.3, focusDistance = 1;
    EnvironmentSettings environmentSettings;
    // info
    int numRenderedFrames = 0, numMeshChunks = 0, numTriangles = 0;
    // scene
    Camera camera; Light light; int width = 1920, height = 1080;
    std::vector<RayTracedSphere> spheres;
    std::vector<RayTracedMesh> meshes;
    bool linearColourSpace = true;              // ProjectSettings.asset:50 — Material.SetColor converts sRGB -> linear
    int intersectMode = RT_INTERSECT_FLAT_CHUNKS;
    // true: the on-device geometry pipeline (rt_upload_local_meshes / rt_set_mesh_transforms) — the local chunks go to the device once and
    // a frame sends one pose per mesh; false: the reference's way, world-space triangles re-marshalled on the host (sent when they changed)
    bool deviceGeometry = false;
    int serialisedNumMeshChunks = -1, serialisedNumTriangles = -1;   // written into the scene by the reference (:156-157)

    void OnValidate();                                                           // :196-203
    void UpdateCameraParams(rt_params& p) const;                                 // :126-133
    void SetShaderParams(rt_params& p) const;                                    // :111-124
    std::vector<rt_sphere> CreateSpheres() const;                                // :167-187
    void CreateMeshes(std::vector<rt_triangle>& tris, std::vector<rt_meshinfo>& infos);   // :135-164
    SceneBuffers BuildBuffers();                                                 // InitFrame :95-109 without the device
    // the geometry pipeline's inputs: every mesh's local chunks back to back (chunk.meshIndex = index into `meshes`) and one pose per mesh
    void CreateLocalMeshes(std::vector<rt_triangle>& tris, std::vector<rt_local_chunk>& chunks);
    std::vector<rt_mesh_transform> CreateTransforms() const;
    // With a device context: InitFrame + the two blits + frame counter (OnRenderImage :49-93).  Throws on C-ABI errors.
    void InitFrame(rt_ctx* ctx);
    void OnRenderImage(rt_ctx* ctx, int frames, std::vector<float>* resultTexture = nullptr);
    void Start(rt_ctx* ctx);                                                     // :43-46
    // The same through an rt_multi: the frame tiles across the GPUs of the node (interleaved row bands inside the library, one
    // gather at the end of the call); resultTexture is the assembled full image.
    void InitFrame(rt_multi* multi);
    void OnRenderImage(rt_multi* multi, int frames, std::vector<float>* resultTexture = nullptr);
    void Start(rt_multi* multi);
    // :190-194 — the reference releases its buffers here; the library owns them, so this only forgets which context holds them
    // (call it before destroying the context: the next InitFrame uploads again)
    void OnDisable() { uploaded_to_ = nullptr; }
private:
    const void* uploaded_to_ = nullptr;         // the context (or rt_multi) that holds this manager's buffers
    SceneBuffers sent_;                         // ... and what they hold (world-space path): only what changed is sent again
    std::vector<rt_triangle> sent_local_; std::vector<rt_local_chunk> sent_chunks_; bool sent_is_local_ = false;
    template <class H, class Api> void InitFrameT(H* h, const Api& api);
};

// Loads a reference scene (Unity YAML).  Throws std::runtime_error with a message on malformed input.
RayTracingManager LoadUnityScene(const std::string& path, int width = 1920, int height = 1080);

} // namespace rthost
