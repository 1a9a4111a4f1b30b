"""ctypes window onto the compiled host (host_cpp/: C++ mirror of the reference's C# components + .unity loader)."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_void_p

import numpy as np

from ._cabi import MESHINFO, PARAMS, SPHERE, TRIANGLE, RtError

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_host.so")
_lib = None


def load_host_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(f"{LIB_PATH} not found: run __graft_entry__.build()")
        L = ctypes.CDLL(LIB_PATH)
        L.rth_load_unity.restype = c_void_p
        L.rth_load_unity.argtypes = [c_char_p, c_int, c_int]
        L.rth_last_error.restype = c_char_p
        L.rth_free.argtypes = [c_void_p]
        L.rth_build.argtypes = [c_void_p, POINTER(c_int)]
        L.rth_copy.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.rth_render.argtypes = [c_void_p, c_int, c_int, POINTER(c_float)]
        L.rth_render_multi.argtypes = [c_void_p, POINTER(c_int), c_int, c_int, POINTER(c_float)]
        L.rth_render_animated.argtypes = [c_void_p, POINTER(c_int), c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_float)]
        L.rth_split_mesh.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]
        L.rth_split_info.argtypes = [c_void_p, c_void_p, c_void_p]
        L.rth_split_triangles.argtypes = [c_void_p]
        _lib = L
    return _lib


def cpp_split_mesh(vertices, normals, indices, sub_ranges, mode=0, transform=None, seed=None, enforce_limit=True):
    """The compiled host's MeshSplitter::CreateChunks (mode 0), RayTracedMesh::GetSubMeshes on a mesh without cached chunks
    (mode 1, transform = position(3) + rotation xyzw(4) + lossyScale(3)) or Split from an explicit seed vertex (mode 2).
    Returns [(TRIANGLE[n], centre[3], size[3], subMeshIndex)]."""
    L = load_host_library()
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.int32).reshape(-1)
    sub = np.ascontiguousarray(sub_ranges, np.int32).reshape(-1, 2)
    tf = np.ascontiguousarray(transform if transform is not None else [0, 0, 0, 0, 0, 0, 1, 1, 1, 1], np.float32)
    sd = np.ascontiguousarray(seed if seed is not None else [0, 0, 0], np.float32)
    nc = L.rth_split_mesh(v.ctypes.data_as(c_void_p), n.ctypes.data_as(c_void_p), len(v), idx.ctypes.data_as(c_void_p), len(idx),
                          sub.ctypes.data_as(c_void_p), len(sub), mode, tf.ctypes.data_as(c_void_p), sd.ctypes.data_as(c_void_p),
                          1 if enforce_limit else 0)
    if nc < 0:
        raise RtError(L.rth_last_error().decode())
    counts, subm, bounds = np.zeros(nc, np.int32), np.zeros(nc, np.int32), np.zeros((nc, 6), np.float32)
    L.rth_split_info(counts.ctypes.data_as(c_void_p), subm.ctypes.data_as(c_void_p), bounds.ctypes.data_as(c_void_p))
    tris = np.zeros(int(counts.sum()), TRIANGLE)
    L.rth_split_triangles(tris.ctypes.data_as(c_void_p))
    out, at = [], 0
    for i in range(nc):
        out.append((tris[at:at + counts[i]], bounds[i, :3].copy(), bounds[i, 3:].copy(), int(subm[i])))
        at += counts[i]
    return out


class CppScene:
    """A reference scene loaded and marshalled by the C++ host."""

    def __init__(self, unity_path: str, width: int, height: int):
        self._L = load_host_library()
        self.width, self.height = width, height
        self._h = self._L.rth_load_unity(unity_path.encode(), width, height)
        if not self._h:
            raise RtError("LoadUnityScene: " + self._L.rth_last_error().decode())

    def close(self):
        if self._h:
            self._L.rth_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def build_buffers(self):
        counts = (c_int * 7)()
        if self._L.rth_build(self._h, counts):
            raise RtError("BuildBuffers: " + self._L.rth_last_error().decode())
        self.counts = dict(zip(("spheres", "triangles", "chunks", "meshes", "serialisedChunks", "serialisedTriangles", "maxBounceCount"), counts))
        params = np.zeros((), PARAMS)
        spheres, tris, infos = np.zeros(counts[0], SPHERE), np.zeros(counts[1], TRIANGLE), np.zeros(counts[2], MESHINFO)
        self._L.rth_copy(self._h, params.ctypes.data_as(c_void_p), spheres.ctypes.data_as(c_void_p),
                         tris.ctypes.data_as(c_void_p), infos.ctypes.data_as(c_void_p))
        return params, spheres, tris, infos

    def render(self, frames: int, device: int = 0) -> np.ndarray:
        """RayTracingManager::Start + OnRenderImage(frames) on a fresh rt_ctx; returns resultTexture [H, W, 4]."""
        out = np.empty((self.height, self.width, 4), np.float32)
        if self._L.rth_render(self._h, device, frames, out.ctypes.data_as(POINTER(c_float))):
            raise RtError("OnRenderImage: " + self._L.rth_last_error().decode())
        return out

    def render_multi(self, frames: int, devices) -> np.ndarray:
        """RayTracingManager::OnRenderImage(rt_multi*, frames): the frame tiled over len(devices) contexts; returns [H, W, 4]."""
        out = np.empty((self.height, self.width, 4), np.float32)
        arr = (c_int * len(devices))(*devices)
        if self._L.rth_render_multi(self._h, arr, len(devices), frames, out.ctypes.data_as(POINTER(c_float))):
            raise RtError("OnRenderImage(rt_multi): " + self._L.rth_last_error().decode())
        return out

    def render_animated(self, steps: int, frames_per_step: int, q, shift, devices=(), device_geometry=False, device: int = 0) -> np.ndarray:
        """`steps` pose changes (every mesh is turned by the quaternion q = x, y, z, w and mesh i moves by shift * (i + 1)), each followed by Start +
        OnRenderImage(frames_per_step), through one rt_ctx (devices empty) or an rt_multi; device_geometry = the on-device pipeline (poses only
        per step).  Returns the last image."""
        out = np.empty((self.height, self.width, 4), np.float32)
        devs = list(devices) if devices else [device]
        arr = (c_int * len(devs))(*devs)
        q4 = (c_float * 4)(*[float(v) for v in q])
        sh = (c_float * 3)(*[float(v) for v in shift])
        if self._L.rth_render_animated(self._h, arr, len(devices), 1 if device_geometry else 0, steps, frames_per_step, q4, sh,
                                       out.ctypes.data_as(POINTER(c_float))):
            raise RtError("render_animated: " + self._L.rth_last_error().decode())
        return out
