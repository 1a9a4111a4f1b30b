"""ctypes binding of include/rt.h (the C-ABI of the HIP path tracer) plus the buffer layouts as numpy dtypes.

The layouts are the reference's own GPU-buffer structs (Assets/Scripts/Data Types/RayTracingMaterial.cs:13-19,
Sphere.cs:5-7, Triangle.cs:8-14, MeshInfo.cs:5-9; strides asserted below: 64 / 80 / 72 / 96 bytes).

There is no CPU fallback: if the HIP library has not been built, or no GPU is present, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_LIB") or os.path.join(_HERE, "librt_mi355x.so")   # RTX_LIB: A/B builds of the same ABI

# ---- buffer layouts ------------------------------------------------------------------------------------------
MATERIAL = np.dtype([
    ("colour", "<f4", 4), ("emissionColour", "<f4", 4), ("specularColour", "<f4", 4),
    ("emissionStrength", "<f4"), ("smoothness", "<f4"), ("specularProbability", "<f4"), ("flag", "<i4"),
])
SPHERE = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("material", MATERIAL)])
TRIANGLE = np.dtype([("posA", "<f4", 3), ("posB", "<f4", 3), ("posC", "<f4", 3),
                     ("normalA", "<f4", 3), ("normalB", "<f4", 3), ("normalC", "<f4", 3)])
MESHINFO = np.dtype([("firstTriangleIndex", "<u4"), ("numTriangles", "<u4"), ("material", MATERIAL),
                     ("boundsMin", "<f4", 3), ("boundsMax", "<f4", 3)])
PARAMS = np.dtype([
    ("width", "<i4"), ("height", "<i4"), ("maxBounceCount", "<i4"), ("numRaysPerPixel", "<i4"),
    ("defocusStrength", "<f4"), ("divergeStrength", "<f4"), ("viewParams", "<f4", 3),
    ("camLocalToWorld", "<f4", 16), ("worldSpaceCameraPos", "<f4", 3), ("worldSpaceLightPos0", "<f4", 3),
    ("environmentEnabled", "<i4"), ("groundColour", "<f4", 4), ("skyColourHorizon", "<f4", 4),
    ("skyColourZenith", "<f4", 4), ("sunFocus", "<f4"), ("sunIntensity", "<f4"),
    ("rngMode", "<i4"), ("intersectMode", "<i4"),
])
STATS = np.dtype([
    ("numRenderedFrames", "<i4"), ("numMeshChunks", "<i4"), ("numTriangles", "<i4"), ("numSpheres", "<i4"),
    ("numBvhNodes", "<i4"), ("bvhMaxStack", "<i4"),
    ("rays", "<u8"), ("sphereTests", "<u8"), ("nodeVisits", "<u8"), ("triTests", "<u8"), ("hits", "<u8"),
    ("phaseLanes", "<u8", 5), ("phaseExecs", "<u8", 5),
    ("lastKernelMs", "<f8"), ("totalKernelMs", "<f8"), ("lastGeometryMs", "<f8"), ("lastDisplayMs", "<f8"), ("lastFramesPerLaunch", "<i4"), ("autoKernel", "<i4"),
    ("lastKernel", "<i4"), ("lastFramesInterleaved", "<i4"),
    ("lastBvhBuildMs", "<f8"), ("refitAreaRatio", "<f4"), ("bvhInternalArea", "<f4"), ("bvhBuiltOnDevice", "<i4"), ("bvhBuilds", "<i4"),
    ("bvhRebuilds", "<i4"), ("bvhRepads", "<i4"), ("lastSampleLanes", "<i4"), ("queuedLaunches", "<i4"), ("regionExecs", "<u8", 32),
    ("primaryLists", "<u4", 4), ("primaryListBuilds", "<i4"), ("_reserved", "<i4"), ("lastPrimaryListsMs", "<f8"),
])
MESH_TRANSFORM = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 4), ("lossyScale", "<f4", 3)])
MULTI_INFO = np.dtype([("numContexts", "<i4"), ("bvhBuilds", "<i4"), ("lastSetupMs", "<f8"), ("lastGatherMs", "<f8"),
                       ("device", "<i4", 16), ("peerAccess", "<i4", 16)])
LOCAL_CHUNK = np.dtype([("firstTriangleIndex", "<u4"), ("numTriangles", "<u4"), ("meshIndex", "<u4"), ("_reserved", "<u4"),
                        ("material", MATERIAL)])
assert MATERIAL.itemsize == 64 and SPHERE.itemsize == 80 and TRIANGLE.itemsize == 72 and MESHINFO.itemsize == 96

RT_INTERSECT_FLAT_CHUNKS = 0
RT_INTERSECT_BRUTE = 1

# every symbol include/rt.h declares (tests check that the built library exports each one)
SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_set_params", "rt_upload_spheres",
    "rt_upload_triangles", "rt_upload_meshinfo", "rt_set_rows", "rt_render_frame", "rt_render",
    "rt_render_counting", "rt_render_frame_flat", "rt_reset_accum", "rt_read_accum", "rt_read_last_frame",
    "rt_copy_accum_to_device", "rt_get_stats", "rt_abi_version", "rt_sizeof", "rt_set_option", "rt_set_bands", "rt_upload_local_meshes", "rt_set_mesh_transforms", "rt_read_world_geometry", "rt_read_display",
    "rt_read_bvh", "rt_write_accum", "rt_submit_frame", "rt_wait",
    "rt_multi_create", "rt_multi_destroy", "rt_multi_last_error", "rt_multi_count", "rt_multi_context", "rt_multi_set_params",
    "rt_multi_upload_spheres", "rt_multi_upload_triangles", "rt_multi_upload_meshinfo", "rt_multi_set_option", "rt_multi_reset_accum",
    "rt_multi_render", "rt_multi_read_accum", "rt_multi_get_stats", "rt_multi_get_info",
    "rt_multi_upload_local_meshes", "rt_multi_set_mesh_transforms", "rt_multi_read_display", "rt_multi_write_accum",
]

_lib = None


class RtError(RuntimeError):
    pass


def load_library() -> ctypes.CDLL:
    """Load librt_mi355x.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                      "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.rt_create.restype = c_void_p
    lib.rt_create.argtypes = [c_int]
    lib.rt_destroy.restype = None
    lib.rt_destroy.argtypes = [c_void_p]
    lib.rt_last_error.restype = c_char_p
    lib.rt_last_error.argtypes = [c_void_p]
    lib.rt_set_stream.argtypes = [c_void_p, c_void_p]
    lib.rt_set_params.argtypes = [c_void_p, c_void_p]
    for n in ("rt_upload_spheres", "rt_upload_triangles", "rt_upload_meshinfo"):
        getattr(lib, n).argtypes = [c_void_p, c_void_p, c_int]
    lib.rt_set_rows.argtypes = [c_void_p, c_int, c_int]
    lib.rt_set_option.argtypes = [c_void_p, c_char_p, c_int]
    lib.rt_set_bands.argtypes = [c_void_p, c_int, c_int]
    lib.rt_upload_local_meshes.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int]
    lib.rt_set_mesh_transforms.argtypes = [c_void_p, c_void_p, c_int]
    lib.rt_read_world_geometry.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int]
    lib.rt_render_frame.argtypes = [c_void_p, c_int]
    lib.rt_render_frame_flat.argtypes = [c_void_p, c_int]
    lib.rt_render.argtypes = [c_void_p, c_int, c_int]
    lib.rt_render_counting.argtypes = [c_void_p, c_int, c_int]
    lib.rt_reset_accum.argtypes = [c_void_p]
    lib.rt_submit_frame.argtypes = [c_void_p, c_int]
    lib.rt_wait.argtypes = [c_void_p]
    lib.rt_read_accum.argtypes = [c_void_p, POINTER(c_float), c_size_t]
    lib.rt_read_last_frame.argtypes = [c_void_p, POINTER(c_float), c_size_t]
    lib.rt_copy_accum_to_device.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.rt_get_stats.argtypes = [c_void_p, c_void_p]
    lib.rt_read_display.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.rt_read_bvh.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
    lib.rt_write_accum.argtypes = [c_void_p, POINTER(c_float), c_size_t, c_int]
    lib.rt_abi_version.restype = c_int
    lib.rt_sizeof.argtypes = [c_char_p]
    lib.rt_multi_create.restype = c_void_p
    lib.rt_multi_create.argtypes = [POINTER(c_int), c_int]
    lib.rt_multi_destroy.restype = None
    lib.rt_multi_destroy.argtypes = [c_void_p]
    lib.rt_multi_last_error.restype = c_char_p
    lib.rt_multi_last_error.argtypes = [c_void_p]
    lib.rt_multi_count.argtypes = [c_void_p]
    lib.rt_multi_context.restype = c_void_p
    lib.rt_multi_context.argtypes = [c_void_p, c_int]
    lib.rt_multi_set_params.argtypes = [c_void_p, c_void_p]
    for n in ("rt_multi_upload_spheres", "rt_multi_upload_triangles", "rt_multi_upload_meshinfo"):
        getattr(lib, n).argtypes = [c_void_p, c_void_p, c_int]
    lib.rt_multi_set_option.argtypes = [c_void_p, c_char_p, c_int]
    lib.rt_multi_reset_accum.argtypes = [c_void_p]
    lib.rt_multi_render.argtypes = [c_void_p, c_int, c_int]
    lib.rt_multi_read_accum.argtypes = [c_void_p, POINTER(c_float), c_size_t]
    lib.rt_multi_get_stats.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.rt_multi_get_info.argtypes = [c_void_p, c_void_p]
    lib.rt_multi_upload_local_meshes.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int]
    lib.rt_multi_set_mesh_transforms.argtypes = [c_void_p, c_void_p, c_int]
    lib.rt_multi_read_display.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.rt_multi_write_accum.argtypes = [c_void_p, POINTER(c_float), c_size_t, c_int]
    for n in SYMBOLS:
        f = getattr(lib, n)
        if f.restype is None or n in ("rt_create", "rt_last_error", "rt_destroy", "rt_multi_create", "rt_multi_destroy", "rt_multi_last_error",
                                      "rt_multi_context"):
            continue
        f.restype = c_int
    for name, dt in (("rt_material", MATERIAL), ("rt_sphere", SPHERE), ("rt_triangle", TRIANGLE),
                     ("rt_meshinfo", MESHINFO), ("rt_params", PARAMS), ("rt_stats", STATS),
                     ("rt_mesh_transform", MESH_TRANSFORM), ("rt_local_chunk", LOCAL_CHUNK), ("rt_multi_info", MULTI_INFO)):
        got = lib.rt_sizeof(name.encode())
        if got != dt.itemsize:
            raise RtError(f"ABI mismatch: sizeof({name}) = {got} in the library, {dt.itemsize} in the binding")
    _lib = lib
    return lib


def _as_buffer(arr, dtype):
    a = np.ascontiguousarray(arr, dtype=dtype)
    return a, a.ctypes.data_as(c_void_p), int(a.shape[0]) if a.ndim else 0


class Tracer:
    """One rt_ctx.  Thin, exception-raising wrapper; semantics are those documented in include/rt.h."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = self._lib.rt_create(device)
        if not self._ctx:
            raise RtError("rt_create failed: " + (self._lib.rt_last_error(None) or b"").decode())
        self._params = None

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.rt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): " + (self._lib.rt_last_error(self._ctx) or b"").decode())

    # -- uploads
    def set_params(self, params):
        p = np.ascontiguousarray(params, dtype=PARAMS).reshape(())
        self._params = p.copy()
        self._check(self._lib.rt_set_params(self._ctx, p.ctypes.data_as(c_void_p)), "rt_set_params")

    def upload(self, spheres=None, triangles=None, meshinfo=None):
        if spheres is not None:
            a, ptr, n = _as_buffer(spheres, SPHERE)
            self._check(self._lib.rt_upload_spheres(self._ctx, ptr, n), "rt_upload_spheres")
        if triangles is not None:
            a, ptr, n = _as_buffer(triangles, TRIANGLE)
            self._check(self._lib.rt_upload_triangles(self._ctx, ptr, n), "rt_upload_triangles")
        if meshinfo is not None:
            a, ptr, n = _as_buffer(meshinfo, MESHINFO)
            self._check(self._lib.rt_upload_meshinfo(self._ctx, ptr, n), "rt_upload_meshinfo")

    # -- on-device geometry pipeline
    def upload_local_meshes(self, local_tris, chunks, n_meshes: int):
        t, tp, nt = _as_buffer(local_tris, TRIANGLE)
        ch, cp, nc = _as_buffer(chunks, LOCAL_CHUNK)
        self._local_counts = (nt, nc)
        self._check(self._lib.rt_upload_local_meshes(self._ctx, tp, nt, cp, nc, int(n_meshes)), "rt_upload_local_meshes")

    def set_mesh_transforms(self, transforms):
        x, xp, n = _as_buffer(transforms, MESH_TRANSFORM)
        self._check(self._lib.rt_set_mesh_transforms(self._ctx, xp, n), "rt_set_mesh_transforms")

    def read_world_geometry(self):
        nt, nc = self._local_counts
        tris, infos = np.zeros(nt, TRIANGLE), np.zeros(nc, MESHINFO)
        self._check(self._lib.rt_read_world_geometry(self._ctx, tris.ctypes.data_as(c_void_p), nt,
                                                     infos.ctypes.data_as(c_void_p), nc), "rt_read_world_geometry")
        return tris, infos

    def set_rows(self, row0: int, nrows: int):
        self._rows = (row0, nrows)
        self._check(self._lib.rt_set_rows(self._ctx, row0, nrows), "rt_set_rows")

    def set_bands(self, first_band: int, band_stride: int, height: int = None):
        H = int(self._params["height"]) if height is None else height
        self._rows = (first_band * 8, sum(min(8, H - y) for y in range(first_band * 8, H, band_stride * 8)))
        self._check(self._lib.rt_set_bands(self._ctx, first_band, band_stride), "rt_set_bands")

    def set_option(self, name: str, value: int):
        self._check(self._lib.rt_set_option(self._ctx, name.encode(), int(value)), f"rt_set_option({name})")

    def set_stream(self, stream_ptr):
        self._check(self._lib.rt_set_stream(self._ctx, c_void_p(stream_ptr)), "rt_set_stream")

    # -- rendering
    def render_frame(self, frame: int):
        self._check(self._lib.rt_render_frame(self._ctx, frame), "rt_render_frame")

    def render(self, first_frame: int, n_frames: int):
        self._check(self._lib.rt_render(self._ctx, first_frame, n_frames), "rt_render")

    def render_counting(self, first_frame: int, n_frames: int):
        self._check(self._lib.rt_render_counting(self._ctx, first_frame, n_frames), "rt_render_counting")

    def render_frame_flat(self, frame: int):
        self._check(self._lib.rt_render_frame_flat(self._ctx, frame), "rt_render_frame_flat")

    def reset_accum(self):
        self._check(self._lib.rt_reset_accum(self._ctx), "rt_reset_accum")

    # -- read-back
    def _strip_shape(self):
        W, H = int(self._params["width"]), int(self._params["height"])
        rows = getattr(self, "_rows", (0, H))[1]
        return rows, W

    def read_accum(self) -> np.ndarray:
        rows, W = self._strip_shape()
        out = np.empty((rows, W, 4), np.float32)
        self._check(self._lib.rt_read_accum(self._ctx, out.ctypes.data_as(POINTER(c_float)), out.size), "rt_read_accum")
        return out

    def submit_frame(self, frame_index: int):
        """Queue one frame (returns at once); wait() or any other call makes sure it is in resultTexture."""
        self._check(self._lib.rt_submit_frame(self._ctx, int(frame_index)), "rt_submit_frame")

    def wait(self):
        self._check(self._lib.rt_wait(self._ctx), "rt_wait")

    def write_accum(self, rgba, frames_rendered: int):
        """Restore a saved resultTexture (as read_accum returned it) and the frame counter."""
        a = np.ascontiguousarray(rgba, np.float32)
        self._check(self._lib.rt_write_accum(self._ctx, a.ctypes.data_as(POINTER(c_float)), a.size, int(frames_rendered)), "rt_write_accum")

    def read_last_frame(self) -> np.ndarray:
        rows, W = self._strip_shape()
        out = np.empty((rows, W, 4), np.float32)
        self._check(self._lib.rt_read_last_frame(self._ctx, out.ctypes.data_as(POINTER(c_float)), out.size), "rt_read_last_frame")
        return out

    def read_display(self) -> np.ndarray:
        """resultTexture as sRGB RGBA8, shape (rows, W, 4) uint8, row 0 = bottom."""
        rows, W = self._strip_shape()
        out = np.empty((rows, W), np.uint32)
        self._check(self._lib.rt_read_display(self._ctx, out.ctypes.data_as(c_void_p), out.size), "rt_read_display")
        return out.view(np.uint8).reshape(rows, W, 4)

    def copy_accum_to_device(self, device_ptr: int, n_floats: int):
        self._check(self._lib.rt_copy_accum_to_device(self._ctx, c_void_p(device_ptr), n_floats), "rt_copy_accum_to_device")

    def read_bvh(self):
        """(f32 nodes [n, 32] float32 view, f16 nodes [n, 32] uint32) of the built BVH4 (see rt_read_bvh)."""
        n = self.stats()["numBvhNodes"]
        f32, f16 = np.zeros((n, 32), np.uint32), np.zeros((n, 32), np.uint32)
        self._check(self._lib.rt_read_bvh(self._ctx, f32.ctypes.data_as(c_void_p), f16.ctypes.data_as(c_void_p), n), "rt_read_bvh")
        return f32, f16

    def stats(self) -> dict:
        s = np.zeros((), STATS)
        self._check(self._lib.rt_get_stats(self._ctx, s.ctypes.data_as(c_void_p)), "rt_get_stats")
        return {k: (s[k].item() if s[k].ndim == 0 else s[k].tolist()) for k in STATS.names}


class MultiTracer:
    """One rt_multi: N contexts (one per entry of `devices`; a device may repeat), interleaved 8-row bands, one gather to the
    first device at the end of render()."""

    def __init__(self, devices):
        self._lib = load_library()
        arr = (c_int * len(devices))(*devices)
        self._m = self._lib.rt_multi_create(arr, len(devices))
        if not self._m:
            raise RtError("rt_multi_create failed: " + (self._lib.rt_multi_last_error(None) or b"").decode())
        self._shape = None

    def close(self):
        if getattr(self, "_m", None):
            self._lib.rt_multi_destroy(self._m)
            self._m = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): " + (self._lib.rt_multi_last_error(self._m) or b"").decode())

    def count(self) -> int:
        return self._lib.rt_multi_count(self._m)

    def set_params(self, params):
        p = np.ascontiguousarray(params, dtype=PARAMS).reshape(())
        self._shape = (int(p["height"]), int(p["width"]))
        self._check(self._lib.rt_multi_set_params(self._m, p.ctypes.data_as(c_void_p)), "rt_multi_set_params")

    def upload(self, spheres=None, triangles=None, meshinfo=None):
        for arr, dt, fn in ((spheres, SPHERE, "rt_multi_upload_spheres"), (triangles, TRIANGLE, "rt_multi_upload_triangles"),
                            (meshinfo, MESHINFO, "rt_multi_upload_meshinfo")):
            if arr is not None:
                a, ptr, n = _as_buffer(arr, dt)
                self._check(getattr(self._lib, fn)(self._m, ptr, n), fn)

    def upload_local_meshes(self, local_tris, chunks, n_meshes: int):
        t, tp, nt = _as_buffer(local_tris, TRIANGLE)
        ch, cp, nc = _as_buffer(chunks, LOCAL_CHUNK)
        self._check(self._lib.rt_multi_upload_local_meshes(self._m, tp, nt, cp, nc, int(n_meshes)), "rt_multi_upload_local_meshes")

    def set_mesh_transforms(self, transforms):
        x, xp, n = _as_buffer(transforms, MESH_TRANSFORM)
        self._check(self._lib.rt_multi_set_mesh_transforms(self._m, xp, n), "rt_multi_set_mesh_transforms")

    def read_display(self) -> np.ndarray:
        """the assembled resultTexture as sRGB RGBA8, shape (H, W, 4) uint8, row 0 = bottom"""
        H, W = self._shape
        out = np.empty((H, W), np.uint32)
        self._check(self._lib.rt_multi_read_display(self._m, out.ctypes.data_as(c_void_p), out.size), "rt_multi_read_display")
        return out.view(np.uint8).reshape(H, W, 4)

    def write_accum(self, rgba, frames_rendered: int):
        """Restore a saved resultTexture (the whole image, as read_accum returned it) and the frame counter on every context."""
        a = np.ascontiguousarray(rgba, np.float32)
        self._check(self._lib.rt_multi_write_accum(self._m, a.ctypes.data_as(POINTER(c_float)), a.size, int(frames_rendered)), "rt_multi_write_accum")

    def set_option(self, name: str, value: int):
        self._check(self._lib.rt_multi_set_option(self._m, name.encode(), int(value)), f"rt_multi_set_option({name})")

    def reset_accum(self):
        self._check(self._lib.rt_multi_reset_accum(self._m), "rt_multi_reset_accum")

    def render(self, first_frame: int, n_frames: int):
        self._check(self._lib.rt_multi_render(self._m, first_frame, n_frames), "rt_multi_render")

    def read_accum(self) -> np.ndarray:
        H, W = self._shape
        out = np.empty((H, W, 4), np.float32)
        self._check(self._lib.rt_multi_read_accum(self._m, out.ctypes.data_as(POINTER(c_float)), out.size), "rt_multi_read_accum")
        return out

    def stats(self) -> dict:
        s = np.zeros((), STATS)
        g = ctypes.c_double(0.0)
        self._check(self._lib.rt_multi_get_stats(self._m, s.ctypes.data_as(c_void_p), ctypes.byref(g)), "rt_multi_get_stats")
        d = {k: (s[k].item() if s[k].ndim == 0 else s[k].tolist()) for k in STATS.names}
        d["gatherMs"] = g.value
        return d

    def info(self) -> dict:
        """rt_multi_get_info: contexts, BVH builds summed over them, the last scene change's set-up time, peer access per context"""
        s = np.zeros((), MULTI_INFO)
        self._check(self._lib.rt_multi_get_info(self._m, s.ctypes.data_as(c_void_p)), "rt_multi_get_info")
        n = int(s["numContexts"])
        return {"numContexts": n, "bvhBuilds": int(s["bvhBuilds"]), "lastSetupMs": float(s["lastSetupMs"]), "lastGatherMs": float(s["lastGatherMs"]),
                "device": s["device"][:min(n, 16)].tolist(), "peerAccess": s["peerAccess"][:min(n, 16)].tolist()}
