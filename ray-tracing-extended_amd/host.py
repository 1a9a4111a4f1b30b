"""Host-side mirror of the reference's scene-description surface, above the C-ABI.

Same names, field meaning and error behaviour as the reference's C# components, so that a scene described for
the reference describes the same scene here:

    RayTracingMaterial   Assets/Scripts/Data Types/RayTracingMaterial.cs:4-29
    EnvironmentSettings  Assets/Scripts/Data Types/EnvironmentSettings.cs:4-11
    MeshChunk            Assets/Scripts/Data Types/MeshChunk.cs:6-17
    RayTracedSphere      Assets/Scripts/Render Types/RayTracedSphere.cs:5-7
    RayTracedMesh        Assets/Scripts/Render Types/RayTracedMesh.cs:17-99   (GetSubMeshes, GetMaterial)
    RayTracingManager    Assets/Scripts/RayTracingManager.cs:11-203           (settings, CreateSpheres, CreateMeshes,
                                                                                UpdateCameraParams, SetShaderParams,
                                                                                OnRenderImage, OnValidate)

All marshal arithmetic is float32 (numpy scalars), in the operation order of the C# / UnityEngine expressions it
restates (UnityEngine itself is closed source: Quaternion*Vector3, Matrix4x4.TRS and the sRGB->linear conversion of
Material.SetColor follow Unity's published formulas — "unpinned").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from ._cabi import LOCAL_CHUNK, MATERIAL, MESH_TRANSFORM, MESHINFO, PARAMS, SPHERE, TRIANGLE, RT_INTERSECT_FLAT_CHUNKS

f32 = np.float32


# ---- UnityEngine value types (float32) ----------------------------------------------------------------------
def _v3(v) -> np.ndarray:
    return np.asarray(v, dtype=np.float32).reshape(3)


def quat_rotate(q: Sequence[float], p: np.ndarray) -> np.ndarray:
    """UnityEngine `Quaternion * Vector3` (q = x,y,z,w); p may be (3,) or (N,3).  float32 throughout."""
    x, y, z, w = (f32(c) for c in q)
    p = np.asarray(p, dtype=np.float32)
    two = f32(2)
    x2, y2, z2 = x * two, y * two, z * two
    xx, yy, zz = x * x2, y * y2, z * z2
    xy, xz, yz = x * y2, x * z2, y * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    one = f32(1)
    px, py, pz = p[..., 0], p[..., 1], p[..., 2]
    rx = (one - (yy + zz)) * px + (xy - wz) * py + (xz + wy) * pz
    ry = (xy + wz) * px + (one - (xx + zz)) * py + (yz - wx) * pz
    rz = (xz - wy) * px + (yz + wx) * py + (one - (xx + yy)) * pz
    return np.stack([rx, ry, rz], axis=-1).astype(np.float32)


def quat_mul(a: Sequence[float], b: Sequence[float]) -> np.ndarray:
    ax, ay, az, aw = (f32(c) for c in a)
    bx, by, bz, bw = (f32(c) for c in b)
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], dtype=np.float32)


@dataclass
class Transform:
    """World-space pose as the reference reads it: transform.position / rotation / lossyScale / localScale."""
    position: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    rotation: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1], np.float32))   # x, y, z, w
    lossyScale: np.ndarray = field(default_factory=lambda: np.ones(3, np.float32))
    localScale: Optional[np.ndarray] = None

    def __post_init__(self):
        self.position = _v3(self.position)
        self.rotation = np.asarray(self.rotation, np.float32).reshape(4)
        self.lossyScale = _v3(self.lossyScale)
        self.localScale = self.lossyScale.copy() if self.localScale is None else _v3(self.localScale)

    @property
    def localToWorldMatrix(self) -> np.ndarray:
        """Matrix4x4.TRS(position, rotation, lossyScale), row-major 4x4 float32."""
        x, y, z, w = (f32(c) for c in self.rotation)
        two, one = f32(2), f32(1)
        r = np.array([[one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
                      [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
                      [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)]], dtype=np.float32)
        m = np.zeros((4, 4), np.float32)
        m[:3, :3] = r * self.lossyScale[None, :]
        m[:3, 3] = self.position
        m[3, 3] = 1
        return m


# ---- data types ---------------------------------------------------------------------------------------------------
class MaterialFlag:
    NONE = 0
    CheckerPattern = 1
    InvisibleLight = 2


@dataclass
class RayTracingMaterial:
    colour: Sequence[float] = (1, 1, 1, 1)
    emissionColour: Sequence[float] = (1, 1, 1, 1)
    specularColour: Sequence[float] = (1, 1, 1, 1)
    emissionStrength: float = 0.0
    smoothness: float = 0.0
    specularProbability: float = 1.0
    flag: int = MaterialFlag.NONE

    def SetDefaultValues(self):                      # RayTracingMaterial.cs:21-28
        self.colour = (1, 1, 1, 1); self.emissionColour = (1, 1, 1, 1); self.emissionStrength = 0.0
        self.specularColour = (1, 1, 1, 1); self.smoothness = 0.0; self.specularProbability = 1.0

    def pack(self) -> np.ndarray:
        m = np.zeros((), MATERIAL)
        for k in ("colour", "emissionColour", "specularColour"):
            c = list(getattr(self, k))
            m[k] = (c + [1.0])[:4] if len(c) == 3 else c
        m["emissionStrength"] = self.emissionStrength
        m["smoothness"] = self.smoothness
        m["specularProbability"] = self.specularProbability
        m["flag"] = int(self.flag)
        return m


@dataclass
class EnvironmentSettings:
    enabled: bool = False
    groundColour: Sequence[float] = (0, 0, 0, 0)
    skyColourHorizon: Sequence[float] = (0, 0, 0, 0)
    skyColourZenith: Sequence[float] = (0, 0, 0, 0)
    sunFocus: float = 1.0
    sunIntensity: float = 0.0


@dataclass
class Bounds:
    center: np.ndarray
    size: np.ndarray

    @property
    def min(self):
        return (self.center - self.size * f32(0.5)).astype(np.float32)

    @property
    def max(self):
        return (self.center + self.size * f32(0.5)).astype(np.float32)


@dataclass
class MeshChunk:
    triangles: np.ndarray            # TRIANGLE[n]
    bounds: Bounds
    subMeshIndex: int = 0


@dataclass
class Camera:
    transform: Transform
    fieldOfView: float = 60.0
    aspect: float = 16.0 / 9.0


@dataclass
class Light:
    """Directional light: _WorldSpaceLightPos0.xyz = -forward."""
    rotation: Sequence[float] = (0, 0, 0, 1)

    @property
    def worldSpaceLightPos0(self) -> np.ndarray:
        fwd = quat_rotate(self.rotation, np.array([0, 0, 1], np.float32))
        return (-fwd).astype(np.float32)


# ---- components ---------------------------------------------------------------------------------------------------
@dataclass
class RayTracedSphere:
    transform: Transform
    material: RayTracingMaterial = field(default_factory=RayTracingMaterial)


class Mesh:
    """What MeshSplitter reads of a UnityEngine.Mesh (MeshSplitter.cs:15-23): vertices, normals, the index buffer
    (`triangles`) and the sub-mesh ranges [(indexStart, indexCount)]."""

    def __init__(self, vertices, normals, triangles, subMeshes=None):
        self.vertices = np.asarray(vertices, np.float32).reshape(-1, 3)
        self.normals = np.asarray(normals, np.float32).reshape(-1, 3)
        self.triangles = np.asarray(triangles, np.int32).reshape(-1)
        self.subMeshes = [(0, len(self.triangles))] if subMeshes is None else [(int(a), int(b)) for a, b in subMeshes]


class RayTracedMesh:
    def __init__(self, transform: Transform, materials: List[RayTracingMaterial], localChunks: Optional[List[MeshChunk]] = None,
                 triangleCount: Optional[int] = None, enforceTriangleLimit: bool = True, sharedMesh: Optional[Mesh] = None):
        self.transform = transform
        self.materials = materials
        self.localChunks = localChunks              # [SerializeField]: the scenes carry them; None = not split yet
        self.mesh: Optional[Mesh] = None            # [SerializeField] Mesh mesh: what the cached chunks were made from
        self.sharedMesh = sharedMesh                # meshFilter.sharedMesh (None: no MeshFilter — only serialised chunks)
        if triangleCount is None:
            triangleCount = sum(len(c.triangles) for c in localChunks) if localChunks else (len(sharedMesh.triangles) // 3 if sharedMesh else 0)
        self.triangleCount = triangleCount
        self.enforceTriangleLimit = enforceTriangleLimit
        self.worldChunks: Optional[List[MeshChunk]] = None

    def GetSubMeshes(self) -> List[MeshChunk]:
        """RayTracedMesh.cs:17-54 — chunks in world space (every triangle re-transformed on the host)."""
        meshTriangles = len(self.mesh.triangles) // 3 if self.mesh is not None else self.triangleCount      # :19
        if self.enforceTriangleLimit and meshTriangles > RayTracingManager.TriangleLimit:
            raise Exception(f"Please use a mesh with fewer than {RayTracingManager.TriangleLimit} triangles")
        # Split mesh into chunks (if result is not already cached)  :24-29
        if self.sharedMesh is not None and (self.mesh is not self.sharedMesh or not self.localChunks):
            self.mesh = self.sharedMesh
            self.localChunks = MeshSplitter.CreateChunks(self.mesh)
            self.triangleCount = len(self.mesh.triangles) // 3
        pos, rot, scale = self.transform.position, self.transform.rotation, self.transform.lossyScale
        self.worldChunks = [self._UpdateWorldChunkFromLocal(c, pos, rot, scale) for c in self.localChunks]
        return self.worldChunks

    @staticmethod
    def _UpdateWorldChunkFromLocal(local: MeshChunk, pos, rot, scale) -> MeshChunk:
        """RayTracedMesh.cs:56-84 — rot * Scale(p, scale) + pos; normals rotated only; tight world AABB."""
        lt = local.triangles
        wt = np.zeros(len(lt), TRIANGLE)
        pts = []
        for k in ("posA", "posB", "posC"):
            w = (quat_rotate(rot, lt[k] * scale[None, :]) + pos[None, :]).astype(np.float32)     # PointLocalToWorld :86-89
            wt[k] = w
            pts.append(w)
        for k in ("normalA", "normalB", "normalC"):
            wt[k] = quat_rotate(rot, lt[k])                                                      # DirectionLocalToWorld :91-94
        allp = np.concatenate(pts, axis=0)
        bmin, bmax = allp.min(axis=0), allp.max(axis=0)
        bounds = Bounds(((bmin + bmax) / f32(2)).astype(np.float32), (bmax - bmin).astype(np.float32))   # :82
        return MeshChunk(wt, bounds, local.subMeshIndex)

    def GetMaterial(self, subMeshIndex: int) -> RayTracingMaterial:                                  # :96-99
        return self.materials[min(subMeshIndex, len(self.materials) - 1)]


def gamma_to_linear(c: float) -> float:
    """sRGB -> linear as Unity applies in Material.SetColor when the project is in Linear colour space."""
    c = float(c)
    if c <= 0.04045:
        return c / 12.92
    if c < 1.0:
        return ((c + 0.055) / 1.055) ** 2.4
    return c ** 2.2


class RayTracingManager:
    """Frame driver / marshaller — RayTracingManager.cs.  `backend` is a Tracer (the HIP C-ABI context)."""
    TriangleLimit = 1500                                                      # RayTracingManager.cs:9

    def __init__(self, camera: Camera, light: Optional[Light] = None, width: int = 1920, height: int = 1080,
                 backend=None, linearColourSpace: bool = True):
        # settings, defaults of RayTracingManager.cs:12-17
        self.maxBounceCount = 4
        self.numRaysPerPixel = 2
        self.defocusStrength = 0.0
        self.divergeStrength = 0.3
        self.focusDistance = 1.0
        self.environmentSettings = EnvironmentSettings()
        # info
        self.numRenderedFrames = 0
        self.numMeshChunks = 0
        self.numTriangles = 0
        # scene
        self.camera = camera
        self.light = light or Light()
        self.width, self.height = int(width), int(height)
        self.spheres: List[RayTracedSphere] = []
        self.meshes: List[RayTracedMesh] = []
        self.linearColourSpace = linearColourSpace        # ProjectSettings.asset:50 (m_ActiveColorSpace: 1)
        self.intersectMode = RT_INTERSECT_FLAT_CHUNKS
        self.backend = backend
        self.deviceGeometry = False                       # True: transform / bounds / BVH refit on the GPU (rt_upload_local_meshes)
        self._dirty = True

    # -- RayTracingManager.cs:196-203
    def OnValidate(self):
        self.maxBounceCount = max(0, self.maxBounceCount)
        self.numRaysPerPixel = max(1, self.numRaysPerPixel)
        self.environmentSettings.sunFocus = max(1, self.environmentSettings.sunFocus)
        self.environmentSettings.sunIntensity = max(0, self.environmentSettings.sunIntensity)

    # -- RayTracingManager.cs:126-133
    def UpdateCameraParams(self, params: np.ndarray):
        deg2rad = f32(0.017453292)
        half = f32(self.camera.fieldOfView) * f32(0.5) * deg2rad
        planeHeight = f32(self.focusDistance) * f32(math.tan(float(half))) * f32(2)
        planeWidth = planeHeight * f32(self.camera.aspect)
        params["viewParams"] = (planeWidth, planeHeight, f32(self.focusDistance))
        params["camLocalToWorld"] = self.camera.transform.localToWorldMatrix.reshape(16)
        params["worldSpaceCameraPos"] = self.camera.transform.position
        params["worldSpaceLightPos0"] = self.light.worldSpaceLightPos0

    # -- RayTracingManager.cs:111-124
    def SetShaderParams(self, params: np.ndarray):
        params["maxBounceCount"] = self.maxBounceCount
        params["numRaysPerPixel"] = self.numRaysPerPixel
        params["defocusStrength"] = self.defocusStrength
        params["divergeStrength"] = self.divergeStrength
        env = self.environmentSettings
        params["environmentEnabled"] = 1 if env.enabled else 0
        for key, col in (("groundColour", env.groundColour), ("skyColourHorizon", env.skyColourHorizon),
                         ("skyColourZenith", env.skyColourZenith)):
            c = list(col) + [1.0] * (4 - len(col))
            if self.linearColourSpace:                       # Material.SetColor converts sRGB -> linear (alpha untouched)
                c = [gamma_to_linear(c[0]), gamma_to_linear(c[1]), gamma_to_linear(c[2]), c[3]]
            params[key] = c
        params["sunFocus"] = env.sunFocus
        params["sunIntensity"] = env.sunIntensity

    # -- RayTracingManager.cs:167-187
    def CreateSpheres(self) -> np.ndarray:
        out = np.zeros(len(self.spheres), SPHERE)
        for i, s in enumerate(self.spheres):
            out[i]["position"] = s.transform.position
            out[i]["radius"] = f32(s.transform.localScale[0]) * f32(0.5)
            out[i]["material"] = s.material.pack()
        return out

    # -- RayTracingManager.cs:135-164
    def CreateMeshes(self):
        tris, infos = [], []
        count = 0
        for mesh in self.meshes:
            for chunk in mesh.GetSubMeshes():
                mi = np.zeros((), MESHINFO)
                mi["firstTriangleIndex"] = count
                mi["numTriangles"] = len(chunk.triangles)
                mi["material"] = mesh.GetMaterial(chunk.subMeshIndex).pack()
                mi["boundsMin"] = chunk.bounds.min                                # MeshInfo.cs:16-17
                mi["boundsMax"] = chunk.bounds.max
                infos.append(mi)
                tris.append(chunk.triangles)
                count += len(chunk.triangles)
        self.numMeshChunks = len(infos)
        self.numTriangles = count
        all_tris = np.concatenate(tris) if tris else np.zeros(0, TRIANGLE)
        all_info = np.array(infos, dtype=MESHINFO) if infos else np.zeros(0, MESHINFO)
        return all_tris, all_info

    # -- the same scene for the on-device geometry pipeline: local chunks once, one transform per mesh per frame
    def build_local_buffers(self):
        tris, chunks = [], []
        count = 0
        for mi, mesh in enumerate(self.meshes):
            if mesh.enforceTriangleLimit and mesh.triangleCount > RayTracingManager.TriangleLimit:
                raise Exception(f"Please use a mesh with fewer than {RayTracingManager.TriangleLimit} triangles")
            for chunk in mesh.localChunks:
                c = np.zeros((), LOCAL_CHUNK)
                c["firstTriangleIndex"], c["numTriangles"], c["meshIndex"] = count, len(chunk.triangles), mi
                c["material"] = mesh.GetMaterial(chunk.subMeshIndex).pack()
                chunks.append(c)
                tris.append(chunk.triangles)
                count += len(chunk.triangles)
        self.numMeshChunks, self.numTriangles = len(chunks), count
        return (np.concatenate(tris) if tris else np.zeros(0, TRIANGLE),
                np.array(chunks, dtype=LOCAL_CHUNK) if chunks else np.zeros(0, LOCAL_CHUNK))

    def build_transforms(self) -> np.ndarray:
        xf = np.zeros(len(self.meshes), MESH_TRANSFORM)
        for i, mesh in enumerate(self.meshes):
            xf[i]["position"], xf[i]["rotation"], xf[i]["lossyScale"] = mesh.transform.position, mesh.transform.rotation, mesh.transform.lossyScale
        return xf

    def build_buffers(self):
        """InitFrame (RayTracingManager.cs:95-109) without the device: params + the three structured buffers."""
        params = np.zeros((), PARAMS)
        params["width"], params["height"] = self.width, self.height
        params["intersectMode"] = self.intersectMode
        self.UpdateCameraParams(params)
        spheres = self.CreateSpheres()
        tris, infos = self.CreateMeshes()
        self.SetShaderParams(params)
        return params, spheres, tris, infos

    def InitFrame(self):
        if self.backend is None:
            raise RuntimeError("RayTracingManager has no backend (HIP Tracer); there is no CPU path")
        if self.deviceGeometry:
            params = np.zeros((), PARAMS)
            params["width"], params["height"] = self.width, self.height
            params["intersectMode"] = self.intersectMode
            self.UpdateCameraParams(params)
            self.SetShaderParams(params)
            self.backend.set_params(params)
            if self._dirty:                                  # geometry: once
                self.backend.upload(spheres=self.CreateSpheres())
                self.backend.upload_local_meshes(*self.build_local_buffers(), len(self.meshes))
                self._dirty = False
            self.backend.set_mesh_transforms(self.build_transforms())    # poses: every frame (40 B per mesh)
            return
        params, spheres, tris, infos = self.build_buffers()
        self.backend.set_params(params)
        if self._dirty:
            self.backend.upload(spheres=spheres, triangles=tris, meshinfo=infos)
            self._dirty = False

    def Start(self):                                                              # RayTracingManager.cs:43-46
        self.numRenderedFrames = 0
        if self.backend is not None:
            self.backend.reset_accum()

    def OnRenderImage(self, frames: int = 1) -> np.ndarray:
        """RayTracingManager.cs:49-93 — trace + accumulate `frames` frames, return resultTexture (rows, W, 4)."""
        self.InitFrame()
        self.backend.render(self.numRenderedFrames, frames)
        self.numRenderedFrames += frames
        return self.backend.read_accum()


# ---- MeshSplitter (Assets/Scripts/Helpers/MeshSplitter.cs) ----------------------------------------------------------------
class _UBounds:
    """UnityEngine.Bounds in float32: stored as centre + extents (Bounds.cs of the public UnityCsReference):
    Bounds(c, size): extents = size*0.5; size = extents*2; min/max = centre -/+ extents;
    SetMinMax(mn, mx): extents = (mx-mn)*0.5, centre = mn + extents; Encapsulate(p) = SetMinMax(Min(min,p), Max(max,p));
    Contains(p): p >= centre-extents and p <= centre+extents on every axis (native AABB::IsInside — unpinned)."""
    __slots__ = ("center", "extents")

    def __init__(self, center, size):
        self.center = np.asarray(center, np.float32).copy()
        self.extents = (np.asarray(size, np.float32) * f32(0.5)).astype(np.float32)

    @property
    def size(self):
        return (self.extents * f32(2)).astype(np.float32)

    @property
    def min(self):
        return (self.center - self.extents).astype(np.float32)

    @property
    def max(self):
        return (self.center + self.extents).astype(np.float32)

    def set_min_max(self, mn, mx):
        self.extents = ((mx - mn) * f32(0.5)).astype(np.float32)
        self.center = (mn + self.extents).astype(np.float32)

    def encapsulate_points(self, pts):
        """Encapsulate a sequence of points one after the other (order matters in float32: centre/extents are re-derived
        after every point)."""
        for p in np.asarray(pts, np.float32).reshape(-1, 3):
            self.set_min_max(np.minimum(self.min, p), np.maximum(self.max, p))

    def contains(self, pts):
        pts = np.asarray(pts, np.float32)
        return np.all((pts >= self.min) & (pts <= self.max), axis=-1)


class MeshSplitter:
    """Restatement of MeshSplitter.cs:8-124: recursive 8-octant split until <= 48 triangles or depth 6; a triangle goes
    to the first octant (x, then y, then z loop order) that contains any of its vertices."""
    maxDepth = 6                # MeshSplitter.cs:8
    maxTrisPerChunk = 48        # MeshSplitter.cs:9

    @staticmethod
    def CreateSubMesh(triangles: np.ndarray, subMeshIndex: int, firstVertex=None) -> MeshChunk:
        """:35-63 — bounds start as a box of size 0.01 at the first vertex, then every vertex is encapsulated in order.
        `triangles` is the sub-mesh's triangle list (TRIANGLE[n]) in index-buffer order."""
        v0 = triangles["posA"][0] if firstVertex is None else firstVertex
        b = _UBounds(v0, np.full(3, 0.01, np.float32))
        pts = np.stack([triangles["posA"], triangles["posB"], triangles["posC"]], axis=1).reshape(-1, 3)
        b.encapsulate_points(pts)
        return MeshChunk(triangles.copy(), b, subMeshIndex)

    @staticmethod
    def CreateChunks(mesh) -> list:
        """:11-33 — `mesh` is a Mesh (vertices, normals, index buffer, sub-mesh ranges), or directly the sub-meshes'
        triangle lists [(TRIANGLE[n], subMeshIndex)] in sub-mesh order."""
        if isinstance(mesh, Mesh):
            sub_meshes = []
            for i, (start, count) in enumerate(mesh.subMeshes):
                idx = mesh.triangles[start:start + count].reshape(-1, 3)
                t = np.zeros(len(idx), TRIANGLE)
                for k, (pf, nf) in enumerate((("posA", "normalA"), ("posB", "normalB"), ("posC", "normalC"))):
                    t[pf] = mesh.vertices[idx[:, k]]
                    t[nf] = mesh.normals[idx[:, k]]
                sub_meshes.append((t, i))
        else:
            sub_meshes = mesh
        out = []
        for tris, idx in sub_meshes:
            MeshSplitter.Split(MeshSplitter.CreateSubMesh(tris, idx), out)
        return out

    @staticmethod
    def Split(chunk: MeshChunk, out: list, depth: int = 0):
        """:65-99"""
        tris = chunk.triangles
        if len(tris) > MeshSplitter.maxTrisPerChunk and depth < MeshSplitter.maxDepth:
            b = chunk.bounds
            q = (b.size / f32(4)).astype(np.float32)
            taken = np.zeros(len(tris), bool)
            for x in (-1, 1):
                for y in (-1, 1):
                    for z in (-1, 1):
                        if len(tris) - int(taken.sum()) > 0:
                            off = np.array([q[0] * f32(x), q[1] * f32(y), q[2] * f32(z)], np.float32)
                            split = _UBounds((b.center + off).astype(np.float32), (q * f32(2)).astype(np.float32))
                            sub = MeshSplitter.Extract(tris, taken, split, chunk.subMeshIndex)
                            if len(sub.triangles) > 0:
                                MeshSplitter.Split(sub, out, depth + 1)
        else:
            out.append(chunk)

    @staticmethod
    def Extract(tris: np.ndarray, taken: np.ndarray, split: "_UBounds", subMeshIndex: int) -> MeshChunk:
        """:101-124"""
        inside = split.contains(tris["posA"]) | split.contains(tris["posB"]) | split.contains(tris["posC"])
        pick = inside & ~taken
        new_bounds = _UBounds(split.center, split.size)
        sel = tris[pick]
        pts = np.stack([sel["posA"], sel["posB"], sel["posC"]], axis=1).reshape(-1, 3)
        new_bounds.encapsulate_points(pts)
        taken |= pick
        return MeshChunk(sel.copy(), new_bounds, subMeshIndex)
