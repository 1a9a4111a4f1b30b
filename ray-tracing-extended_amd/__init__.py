"""MI355X-native path tracer behind the Ray-Tracing-Extended scene surface.

The directory name carries a hyphen, so import it through `rtx_pkg.load()` (repo root), which registers it as the
module `rtx_amd`.  Only what the hot path needs lives here:

    csrc/      HIP kernels (gfx950), BVH builder and the C-ABI of include/rt.h  -> librt_mi355x.so
    _cabi.py   ctypes binding + buffer layouts
    host.py    mirror of the reference's host components (RayTracingManager, RayTracedSphere, RayTracedMesh ...)
    host_cpp/  the same components + .unity loader as compiled C++ (librt_host.so); host_cpp_binding.py = ctypes window
    scenes.py  synthetic workloads of BASELINE.json configs
    unity_scene.py  loader for the reference's .unity scenes + compact scene interchange format
    distributed.py  row-strip decomposition + frame-end gather
"""
from . import _cabi, distributed, host, host_cpp_binding, imageio, scenes, unity_scene  # noqa: F401
from ._cabi import (LOCAL_CHUNK, MATERIAL, MESH_TRANSFORM, MULTI_INFO, MESHINFO, PARAMS, SPHERE, STATS, TRIANGLE, RT_INTERSECT_BRUTE,  # noqa: F401
                    RT_INTERSECT_FLAT_CHUNKS, MultiTracer, RtError, Tracer, load_library)
from .host import (Camera, EnvironmentSettings, Light, MaterialFlag, Mesh, MeshChunk, MeshSplitter, RayTracedMesh,  # noqa: F401
                   RayTracedSphere, RayTracingManager, RayTracingMaterial, Transform)
