"""Synthetic workloads of BASELINE.json `configs` (frozen tables), built through the host mirror
(host.py) so they exercise the same marshal path a reference scene would.

Every generator returns a RayTracingManager without a backend; `.build_buffers()` gives (params, spheres,
triangles, meshinfo) in the reference's buffer layouts.
"""
from __future__ import annotations

import numpy as np

from ._cabi import TRIANGLE
from .host import (Bounds, Camera, EnvironmentSettings, Light, MaterialFlag, MeshChunk, RayTracedMesh,
                   RayTracedSphere, RayTracingManager, RayTracingMaterial, Transform)

f32 = np.float32


class Pcg:
    """The shader's PCG stream (RayTracing.shader:193-204) in Python integers — used to seed scene tables."""

    def __init__(self, seed: int):
        self.state = seed & 0xFFFFFFFF

    def next_u32(self) -> int:
        self.state = (self.state * 747796405 + 2891336453) & 0xFFFFFFFF
        s = self.state
        r = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
        return ((r >> 22) ^ r) & 0xFFFFFFFF

    def value(self) -> float:
        return float(f32(self.next_u32()) * f32(2.0 ** -32))


BALLS_OUTDOORS_ENV = dict(                      # "Balls Outdoors.unity":493-499
    enabled=True, groundColour=(0.35, 0.3, 0.35, 0), skyColourHorizon=(1, 1, 1, 0),
    skyColourZenith=(0.0788092, 0.36480793, 0.7264151, 0), sunFocus=500.0, sunIntensity=200.0)
BALLS_OUTDOORS_CAMERA = dict(                   # "Balls Outdoors.unity":531,556-558
    position=(1.4000001, 1.5800002, 2.55), rotation=(0.028254312, -0.9670911, 0.12388348, 0.22043003),
    scale=(1.0000004, 1, 1.0000004), fov=53.7)
BALLS_OUTDOORS_LIGHT = (0.058276325, -0.38376775, 0.44279444, 0.8082445)   # "Balls Outdoors.unity":742


def _sphere(pos, diameter, **mat) -> RayTracedSphere:
    return RayTracedSphere(Transform(position=pos, lossyScale=(diameter,) * 3), RayTracingMaterial(**mat))


def config1(width: int = 256, height: int = 256) -> RayTracingManager:
    """configs[0]: 16 random spheres, 256x256, 4 spp, 3 bounces (the CPU-runnable case)."""
    c = BALLS_OUTDOORS_CAMERA
    cam = Camera(Transform(position=c["position"], rotation=c["rotation"], lossyScale=c["scale"]),
                 fieldOfView=c["fov"], aspect=width / height)
    m = RayTracingManager(cam, Light(BALLS_OUTDOORS_LIGHT), width, height)
    m.maxBounceCount, m.numRaysPerPixel = 3, 4
    m.defocusStrength, m.divergeStrength, m.focusDistance = 0.0, 0.5, 1.0
    m.environmentSettings = EnvironmentSettings(**BALLS_OUTDOORS_ENV)
    black = (0, 0, 0, 0)
    # ground: "Balls Outdoors.unity":260,279,359-360 (scale 50 -> radius 25)
    m.spheres.append(_sphere((0, -25, 0), 50.0, colour=(0.384, 0.157, 0.812, 0), emissionColour=black,
                             specularColour=black, specularProbability=0.0))
    rng = Pcg(1)
    for i in range(1, 16):
        x = -4.0 + 8.0 * rng.value()
        z = -4.0 + 8.0 * rng.value()
        r = 0.2 + 0.5 * rng.value()
        col = (rng.value(), rng.value(), rng.value(), 1.0)
        mat = dict(colour=col, emissionColour=black, specularColour=(1, 1, 1, 1), specularProbability=0.0)
        if i % 5 == 0:
            mat.update(emissionColour=col, emissionStrength=4.0)
        if i % 3 == 0:
            mat.update(smoothness=0.9, specularProbability=0.5)
        m.spheres.append(_sphere((x, r, z), 2.0 * r, **mat))
    return m


def config2(width: int = 1920, height: int = 1080) -> RayTracingManager:
    """configs[1]: Cornell-box-style 10 spheres + 2 emissive, env off, 256 spp (4 frames x 64), 8 bounces."""
    cam = Camera(Transform(position=(0, 2, -4.84)), fieldOfView=60.0, aspect=width / height)
    m = RayTracingManager(cam, Light(), width, height)
    m.maxBounceCount, m.numRaysPerPixel = 8, 64
    m.defocusStrength, m.divergeStrength, m.focusDistance = 0.0, 0.3, 1.0
    m.environmentSettings = EnvironmentSettings(enabled=False)
    black, white = (0, 0, 0, 0), (1, 1, 1, 1)
    R = 100.0
    walls = [   # centre, colour, flag, emissionColour (second checker colour)
        ((-3 - R, 2, 0), (0.97, 0, 0, 1), 0, black),            # left, red
        ((3 + R, 2, 0), (0.1, 1, 0.03, 1), 0, black),           # right, green
        ((0, 2, 3 + R), (0.13, 0.39, 0.91, 1), 0, black),       # back, blue
        ((0, -R, 0), white, MaterialFlag.CheckerPattern, (0.2, 0.2, 0.2, 1)),   # floor, checker
        ((0, 4 + R, 0), (0.15, 0.15, 0.15, 1), 0, black),       # ceiling, grey
        ((0, 2, -5 - R), white, 0, black),                      # front (behind the camera)
    ]
    for pos, col, flag, emi in walls:
        m.spheres.append(_sphere(pos, 2 * R, colour=col, emissionColour=emi, specularColour=white,
                                 specularProbability=0.0, flag=flag))
    for x, smooth in zip((-2.1, -0.7, 0.7, 2.1), (0.303, 0.622, 0.883, 1.0)):   # "Reflective Balls.unity" glossy row
        m.spheres.append(_sphere((x, 1.757, 1.0), 1.15, colour=white, emissionColour=black, specularColour=white,
                                 smoothness=smooth, specularProbability=1.0))
    for x in (-1.5, 1.5):                                                        # emissive, (1,0.9997,0.778) x 13
        m.spheres.append(_sphere((x, 3.6, 0.0), 0.8, colour=black, emissionColour=(1, 0.9997, 0.778, 1),
                                 specularColour=white, emissionStrength=13.0, specularProbability=0.0))
    return m


# ---- small procedural triangle meshes (parity cases that do not need the reference's assets) -------------------
def _tri_array(verts, normals) -> np.ndarray:
    verts = np.asarray(verts, np.float32).reshape(-1, 3, 3)
    normals = np.asarray(normals, np.float32).reshape(-1, 3, 3)
    t = np.zeros(len(verts), TRIANGLE)
    t["posA"], t["posB"], t["posC"] = verts[:, 0], verts[:, 1], verts[:, 2]
    t["normalA"], t["normalB"], t["normalC"] = normals[:, 0], normals[:, 1], normals[:, 2]
    return t


def cube_triangles() -> np.ndarray:
    """Unit cube centred at the origin, 12 triangles, outward winding (front faces per RayTriangle's det >= 1e-6)."""
    tris, nrm = [], []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            n = np.zeros(3); n[axis] = sign
            u = np.zeros(3); v = np.zeros(3)
            u[(axis + 1) % 3] = 1.0; v[(axis + 2) % 3] = 1.0
            if sign < 0:
                u, v = v, u
            c = n * 0.5
            p = [c - 0.5 * u - 0.5 * v, c + 0.5 * u - 0.5 * v, c + 0.5 * u + 0.5 * v, c - 0.5 * u + 0.5 * v]
            # choose winding so that cross(B-A, C-A) points along -n ... RayTriangle hits when -dot(dir, cross) >= 1e-6,
            # i.e. when the ray travels against cross(AB, AC): front face normal = cross(AB, AC).
            for a, b, cc in ((0, 1, 2), (0, 2, 3)):
                A, B, C = p[a], p[b], p[cc]
                if np.dot(np.cross(B - A, C - A), n) < 0:
                    B, C = C, B
                tris.append([A, B, C]); nrm.append([n, n, n])
    return _tri_array(tris, nrm)


def uv_sphere_triangles(stacks: int = 8, slices: int = 12) -> np.ndarray:
    """Unit-radius triangulated sphere with smooth normals (exercises normal interpolation)."""
    tris, nrm = [], []

    def pt(i, j):
        th = np.pi * i / stacks
        ph = 2 * np.pi * j / slices
        return np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])

    for i in range(stacks):
        for j in range(slices):
            q = [pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)]
            for a, b, c in ((0, 1, 2), (0, 2, 3)):
                A, B, C = q[a], q[b], q[c]
                cr = np.cross(B - A, C - A)
                if np.linalg.norm(cr) < 1e-9:
                    continue
                if np.dot(cr, A + B + C) < 0:
                    B, C = C, B
                tris.append([A, B, C]); nrm.append([A, B, C])
    return _tri_array(tris, nrm)


def chunked(tris: np.ndarray, max_tris: int = 48, sub_mesh_index: int = 0):
    """Cut a triangle list into consecutive chunks of <= max_tris with the CreateSubMesh-style padded bounds."""
    out = []
    for i in range(0, len(tris), max_tris):
        t = tris[i:i + max_tris]
        pts = np.concatenate([t["posA"], t["posB"], t["posC"]])
        mn, mx = pts.min(0), pts.max(0)
        out.append(MeshChunk(t.copy(), Bounds(((mn + mx) / f32(2)).astype(np.float32), (mx - mn).astype(np.float32)),
                             sub_mesh_index))
    return out


def mesh_test_scene(width: int = 96, height: int = 64, n_objects: int = 6, seed: int = 7) -> RayTracingManager:
    """Small mixed scene: floor quad, emissive quad, rotated cubes and tessellated spheres, two analytic spheres."""
    cam = Camera(Transform(position=(0.3, 2.2, -7.0), rotation=(0.12, 0.0, 0.0, 0.99277)), fieldOfView=50.0,
                 aspect=width / height)
    m = RayTracingManager(cam, Light(BALLS_OUTDOORS_LIGHT), width, height)
    m.maxBounceCount, m.numRaysPerPixel = 4, 4
    m.divergeStrength, m.focusDistance = 0.5, 1.0
    m.environmentSettings = EnvironmentSettings(**BALLS_OUTDOORS_ENV)
    m.environmentSettings.sunIntensity = 10.0
    black, white = (0, 0, 0, 0), (1, 1, 1, 1)
    quad = _tri_array([[[-1, 0, -1], [-1, 0, 1], [1, 0, 1]], [[-1, 0, -1], [1, 0, 1], [1, 0, -1]]],
                      [[[0, 1, 0]] * 3] * 2)
    m.meshes.append(RayTracedMesh(Transform(lossyScale=(8, 1, 8)),
                                  [RayTracingMaterial(colour=white, emissionColour=(0.1, 0.1, 0.4, 1), specularColour=white,
                                                      specularProbability=0.0, flag=MaterialFlag.CheckerPattern)],
                                  chunked(quad)))
    # emissive quad facing down
    m.meshes.append(RayTracedMesh(Transform(position=(0, 5, 0), rotation=(1, 0, 0, 0), lossyScale=(2, 1, 2)),
                                  [RayTracingMaterial(colour=black, emissionColour=white, specularColour=white,
                                                      emissionStrength=6.0, specularProbability=0.0)],
                                  chunked(quad)))
    rng = Pcg(seed)
    cube, ball = cube_triangles(), uv_sphere_triangles()
    for i in range(n_objects):
        ang = rng.value() * np.pi
        q = (0.0, float(np.sin(ang / 2)), 0.0, float(np.cos(ang / 2)))
        s = 0.6 + 0.9 * rng.value()
        pos = (-4.0 + 8.0 * rng.value(), s * 0.5 if i % 2 == 0 else s, -2.0 + 6.0 * rng.value())
        mat = RayTracingMaterial(colour=(rng.value(), rng.value(), rng.value(), 1), emissionColour=black,
                                 specularColour=white, smoothness=0.8 if i % 3 == 0 else 0.0,
                                 specularProbability=0.4 if i % 3 == 0 else 0.0)
        src = cube if i % 2 == 0 else ball
        m.meshes.append(RayTracedMesh(Transform(position=pos, rotation=q, lossyScale=(s, s, s)), [mat], chunked(src, 40)))
    m.spheres.append(_sphere((2.5, 0.7, -3.0), 1.4, colour=white, emissionColour=black, specularColour=white,
                             smoothness=1.0, specularProbability=1.0))
    m.spheres.append(_sphere((-2.0, 0.5, -3.5), 1.0, colour=(0.9, 0.3, 0.2, 1), emissionColour=black,
                             specularColour=white, specularProbability=0.0))
    return m


# ---- configs[2..4]: the reference's Chess scene, instanced -------------------------------------------------------
import os as _os

CONFIGS_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "configs")     # frozen workload tables


def chess_instanced(copies: int, width: int = 1920, height: int = 1080, columns: int = 5, spacing: float = 9.0,
                    dof: bool = False, bounces: int = 8, scene_path: str = None) -> RayTracingManager:
    """The 15 pieces of Assets/Scenes/Chess.unity (5,908 triangles) instanced `copies` times on a grid in front of
    the camera, copy k > 0 yawed about its own board centre by a PCG-seeded angle (seed 3), over one checker floor
    quad, lit by the scene's invisible area light and its sky (sunIntensity 0, Chess.unity:30179-30185).
    copies = 17 -> 100,436 + 4 triangles (configs[2], [3]); copies = 170 -> 1,004,360 + 4 (configs[4])."""
    from .unity_scene import load_scene_npz
    base = load_scene_npz(scene_path or _os.path.join(CONFIGS_DIR, "chess_scene.npz"), width, height)
    pieces = [m for m in base.meshes if m.triangleCount > 2]
    quads = [m for m in base.meshes if m.triangleCount <= 2]
    board = next(q for q in quads if q.materials[0].flag == MaterialFlag.CheckerPattern)
    light = next(q for q in quads if q.materials[0].flag == MaterialFlag.InvisibleLight)
    assert len(pieces) == 15 and sum(p.triangleCount for p in pieces) == 5908

    mgr = RayTracingManager(base.camera, base.light, width, height)
    mgr.maxBounceCount, mgr.numRaysPerPixel = bounces, 64
    mgr.environmentSettings = base.environmentSettings
    mgr.focusDistance = base.focusDistance                       # 3.82 (Chess.unity:30178)
    if dof:
        mgr.defocusStrength, mgr.divergeStrength = base.defocusStrength, base.divergeStrength   # 180, 1 (:30176-30177)
    else:
        mgr.defocusStrength, mgr.divergeStrength = 0.0, base.divergeStrength

    rng = Pcg(3)
    rows = (copies + columns - 1) // columns
    from .host import quat_mul, quat_rotate
    for k in range(copies):
        gx, gz = k % columns - columns // 2, k // columns
        # copy 0 sits where the reference scene is; the grid grows to both sides and away from the camera
        off = np.array([gx * spacing, 0.0, gz * spacing], np.float32)
        yaw = 0.0 if k == 0 else rng.value() * 2.0 * np.pi
        qy = np.array([0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)], np.float32)
        for p in pieces:
            tr = Transform(position=(quat_rotate(qy, p.transform.position) + off).astype(np.float32),
                           rotation=quat_mul(qy, p.transform.rotation), lossyScale=p.transform.lossyScale)
            mgr.meshes.append(RayTracedMesh(tr, p.materials, p.localChunks, triangleCount=p.triangleCount))
    # one floor under the whole grid (the board quad is the unit quad [-0.5,0.5]^2 scaled by 8 in the reference)
    span_x, span_z = columns * spacing + 8.0, rows * spacing + 8.0
    s = float(max(span_x, span_z))
    mgr.meshes.append(RayTracedMesh(Transform(position=(0.0, 0.0, (rows - 1) * spacing * 0.5), rotation=board.transform.rotation,
                                              lossyScale=(s, s, s)), board.materials, board.localChunks, triangleCount=2))
    mgr.meshes.append(RayTracedMesh(light.transform, light.materials, light.localChunks, triangleCount=2))
    return mgr


def workload_table() -> dict:
    """configs/workloads.json: the frozen parameters of BASELINE.json's five configurations."""
    import json
    with open(_os.path.join(CONFIGS_DIR, "workloads.json")) as f:
        return json.load(f)


def _chess_workload(key: str, width: int, height: int) -> RayTracingManager:
    w = workload_table()[key]
    return chess_instanced(w["copies"], width or w["width"], height or w["height"], columns=w["columns"], spacing=w["spacing"],
                           dof=w["dof"], bounces=w["bounces"])


def config3(width: int = 0, height: int = 0) -> RayTracingManager:
    """configs[2]: ~100k triangles, 1920x1080, 1024 spp = 16 frames x 64 rays, 8 bounces, DOF off."""
    return _chess_workload("config3", width, height)


def config4(width: int = 0, height: int = 0) -> RayTracingManager:
    """configs[3]: same scene, 3840x2160, 4096 spp = 64 frames x 64 rays, 12 bounces (8 GPUs, row strips)."""
    return _chess_workload("config4", width, height)


def config5(width: int = 0, height: int = 0) -> RayTracingManager:
    """configs[4]: ~1.0M triangles (170 copies, 17 columns), DOF on (180 / 1 / 3.82), 1024 spp, 8 bounces."""
    return _chess_workload("config5", width, height)


def sphere_table(mgr: RayTracingManager) -> list:
    """The spheres of a manager as plain rows (configs/config1_spheres.json, config2_spheres.json hold the frozen ones)."""
    _, spheres, _, _ = mgr.build_buffers()
    rows = []
    for s in spheres:
        m = s["material"]
        rows.append({"position": [float(v) for v in s["position"]], "radius": float(s["radius"]),
                     "colour": [float(v) for v in m["colour"]], "emissionColour": [float(v) for v in m["emissionColour"]],
                     "specularColour": [float(v) for v in m["specularColour"]], "emissionStrength": float(m["emissionStrength"]),
                     "smoothness": float(m["smoothness"]), "specularProbability": float(m["specularProbability"]),
                     "flag": int(m["flag"])})
    return rows
