"""Loader for the reference's `.unity` scenes (Unity YAML), producing the host mirror's RayTracingManager.

What is read (SURVEY.md §8b "Scene surface"):
  * components are recognised by script GUID (Assets/Scripts/RayTracingManager.cs.meta:2,
    "Assets/Scripts/Render Types/RayTracedSphere.cs.meta":2, "Assets/Scripts/Render Types/RayTracedMesh.cs.meta":2);
  * RayTracingManager fields maxBounceCount, numRaysPerPixel, defocusStrength, divergeStrength, focusDistance,
    environmentSettings{...} (e.g. Assets/Scenes/Chess.unity:30174-30185);
  * RayTracedSphere.material, RayTracedMesh.materials[] and the serialised MeshSplitter output `localChunks[]`
    (triangles, bounds m_Center/m_Extent, subMeshIndex; e.g. "Assets/Scenes/Reflective Balls.unity":356-446);
  * Transform (m_LocalPosition/Rotation/Scale, m_Father), PrefabInstance overrides + m_TransformParent
    (Assets/Scenes/Knight.unity:204-263), Camera "field of view", directional Light rotation.
FBX prefab roots that do not override a transform property keep the importer's values for these Blender exports:
scale 100, rotation -90 deg about X (Assets/Graphics/*.fbx).

Also a compact scene interchange format (`save_scene_npz` / `load_scene_npz`): plain numeric arrays only, so a scene
can be cached or shipped without the YAML.
"""
from __future__ import annotations

import json
import re
from typing import Dict, Optional

import numpy as np

from ._cabi import TRIANGLE
from .host import (Bounds, Camera, EnvironmentSettings, Light, MeshChunk, RayTracedMesh, RayTracedSphere,
                   RayTracingManager, RayTracingMaterial, Transform, quat_mul, quat_rotate)

GUID_MANAGER = "68c390cdf7a860745bbbdeccd7d206a9"
GUID_SPHERE = "52a9ac6d93ef8ff438ff410be33e635a"
GUID_MESH = "da1318d85859d584682b30dbc26ca9f6"

FBX_ROOT_SCALE = (100.0, 100.0, 100.0)
FBX_ROOT_ROTATION = (-0.7071068, 0.0, 0.0, 0.7071067)

_HEADER = re.compile(r"^--- !u!(\d+) &(-?\d+)( stripped)?\s*$", re.M)


def _xyz(d, keys=("x", "y", "z")):
    return np.array([d[k] for k in keys], dtype=np.float32)


def _colour(d):
    return (float(d["r"]), float(d["g"]), float(d["b"]), float(d["a"]))


def _material(d) -> RayTracingMaterial:
    return RayTracingMaterial(colour=_colour(d["colour"]), emissionColour=_colour(d["emissionColour"]),
                              specularColour=_colour(d["specularColour"]), emissionStrength=float(d["emissionStrength"]),
                              smoothness=float(d["smoothness"]), specularProbability=float(d["specularProbability"]),
                              flag=int(d["flag"]))


def _chunks(raw) -> list:
    out = []
    for c in raw or []:
        tl = c["triangles"] or []
        t = np.zeros(len(tl), TRIANGLE)
        for k in TRIANGLE.names:
            t[k] = np.array([[tr[k]["x"], tr[k]["y"], tr[k]["z"]] for tr in tl], dtype=np.float32).reshape(-1, 3)
        centre = _xyz(c["bounds"]["m_Center"])
        extent = _xyz(c["bounds"]["m_Extent"])
        out.append(MeshChunk(t, Bounds(centre, (extent * np.float32(2)).astype(np.float32)), int(c["subMeshIndex"])))
    return out


class _Pose:
    __slots__ = ("pos", "rot", "scale", "parent")

    def __init__(self, pos, rot, scale, parent):
        self.pos, self.rot, self.scale, self.parent = pos, rot, scale, parent


def load_unity_scene(path: str, width: int = 1920, height: int = 1080, backend=None) -> RayTracingManager:
    import yaml
    loader = getattr(yaml, "CSafeLoader", yaml.SafeLoader)
    text = open(path, "r", encoding="utf-8").read()
    parts = _HEADER.split(text)
    docs: Dict[int, tuple] = {}
    order = []
    for i in range(1, len(parts), 4):
        cls, fid, stripped = int(parts[i]), int(parts[i + 1]), bool(parts[i + 2])
        body = yaml.load(parts[i + 3], Loader=loader) or {}
        (_, payload), = body.items() if body else ((None, {}),)
        docs[fid] = (cls, stripped, payload or {})
        order.append(fid)

    # ---- transforms -----------------------------------------------------------------------------------------------
    poses: Dict[int, _Pose] = {}             # transform fileID -> local pose
    go_transform: Dict[int, int] = {}        # game object fileID -> transform fileID
    go_active: Dict[int, bool] = {}
    prefab_pose: Dict[int, _Pose] = {}       # PrefabInstance fileID -> pose of its root
    prefab_active: Dict[int, bool] = {}
    for fid in order:
        cls, stripped, d = docs[fid]
        if cls == 4 and not stripped:
            q = d["m_LocalRotation"]
            poses[fid] = _Pose(_xyz(d["m_LocalPosition"]), np.array([q["x"], q["y"], q["z"], q["w"]], np.float32),
                               _xyz(d["m_LocalScale"]), int(d["m_Father"]["fileID"]))
            go_transform[int(d["m_GameObject"]["fileID"])] = fid
        elif cls == 1 and not stripped:
            go_active[fid] = bool(d.get("m_IsActive", 1))
        elif cls == 1001:
            mod = d["m_Modification"]
            pos = np.zeros(3, np.float32)
            rot = np.array(FBX_ROOT_ROTATION, np.float32)
            scale = np.array(FBX_ROOT_SCALE, np.float32)
            active = True
            for m in mod.get("m_Modifications") or []:
                pp, val = m["propertyPath"], m["value"]
                mm = re.fullmatch(r"m_Local(Position|Rotation|Scale)\.([xyzw])", pp)
                if mm:
                    tgt = {"Position": pos, "Rotation": rot, "Scale": scale}[mm.group(1)]
                    tgt["xyzw".index(mm.group(2))] = np.float32(float(val))
                elif pp == "m_IsActive":
                    active = bool(int(val))
            prefab_pose[fid] = _Pose(pos, rot, scale, int(mod["m_TransformParent"]["fileID"]))
            prefab_active[fid] = active

    def world(pose: Optional[_Pose]) -> Transform:
        """Compose up the m_Father / m_TransformParent chain (rotation and per-axis scale; no shear)."""
        chain = []
        p = pose
        while p is not None:
            chain.append(p)
            par = p.parent
            if par == 0:
                break
            if par in poses:
                p = poses[par]
            elif par in docs and docs[par][0] == 4 and docs[par][1]:          # stripped transform of another prefab
                p = prefab_pose.get(int(docs[par][2]["m_PrefabInstance"]["fileID"]))
            else:
                p = None
        wpos, wrot, wscale = np.zeros(3, np.float32), np.array([0, 0, 0, 1], np.float32), np.ones(3, np.float32)
        for p in reversed(chain):
            wpos = (wpos + quat_rotate(wrot, wscale * p.pos)).astype(np.float32)
            wrot = quat_mul(wrot, p.rot)
            wscale = (wscale * p.scale).astype(np.float32)
        local_scale = chain[0].scale if chain else np.ones(3, np.float32)
        return Transform(position=wpos, rotation=wrot, lossyScale=wscale, localScale=local_scale)

    def transform_of_go(go_id: int):
        """-> (Transform, active) for a scene or prefab-instance game object."""
        cls, stripped, d = docs[go_id]
        if stripped:
            pi = int(d["m_PrefabInstance"]["fileID"])
            return world(prefab_pose[pi]), prefab_active.get(pi, True)
        return world(poses[go_transform[go_id]]), go_active.get(go_id, True)

    # ---- components -------------------------------------------------------------------------------------------------
    manager_doc = None
    camera = None
    light = Light()
    spheres, meshes = [], []
    for fid in order:
        cls, stripped, d = docs[fid]
        if cls == 114 and not stripped:
            guid = (d.get("m_Script") or {}).get("guid")
            go = int(d["m_GameObject"]["fileID"])
            if guid == GUID_MANAGER:
                manager_doc = d
            elif guid == GUID_SPHERE:
                tr, active = transform_of_go(go)
                if active:
                    spheres.append(RayTracedSphere(tr, _material(d["material"])))
            elif guid == GUID_MESH:
                tr, active = transform_of_go(go)
                if active:
                    meshes.append(RayTracedMesh(tr, [_material(m) for m in d["materials"]], _chunks(d.get("localChunks")),
                                                triangleCount=int(d.get("triangleCount", 0)) or None))
        elif cls == 20 and not stripped:
            tr, _ = transform_of_go(int(d["m_GameObject"]["fileID"]))
            camera = Camera(tr, fieldOfView=float(d["field of view"]), aspect=width / height)
        elif cls == 108 and not stripped and int(d.get("m_Type", -1)) == 1:
            tr, _ = transform_of_go(int(d["m_GameObject"]["fileID"]))
            light = Light(tuple(float(c) for c in tr.rotation))
    if manager_doc is None or camera is None:
        raise ValueError(f"{path}: no RayTracingManager / Camera found")

    mgr = RayTracingManager(camera, light, width, height, backend=backend)
    for k in ("maxBounceCount", "numRaysPerPixel"):
        setattr(mgr, k, int(manager_doc[k]))
    for k in ("defocusStrength", "divergeStrength", "focusDistance"):
        setattr(mgr, k, float(manager_doc[k]))
    e = manager_doc["environmentSettings"]
    mgr.environmentSettings = EnvironmentSettings(
        enabled=bool(e["enabled"]), groundColour=_colour(e["groundColour"]), skyColourHorizon=_colour(e["skyColourHorizon"]),
        skyColourZenith=_colour(e["skyColourZenith"]), sunFocus=float(e["sunFocus"]), sunIntensity=float(e["sunIntensity"]))
    mgr.spheres, mgr.meshes = spheres, meshes
    # the reference's own bookkeeping, serialised by CreateMeshes (RayTracingManager.cs:156-157) — used as a KAT
    mgr.serialisedInfo = dict(numMeshChunks=int(manager_doc.get("numMeshChunks", 0)),
                              numTriangles=int(manager_doc.get("numTriangles", 0)))
    return mgr


# ---- compact interchange format -------------------------------------------------------------------------------------
def _mat_row(m: RayTracingMaterial):
    return list(m.colour) + list(m.emissionColour) + list(m.specularColour) + [m.emissionStrength, m.smoothness,
                                                                              m.specularProbability, float(m.flag)]


def _row_mat(r) -> RayTracingMaterial:
    r = [float(x) for x in r]
    return RayTracingMaterial(colour=tuple(r[0:4]), emissionColour=tuple(r[4:8]), specularColour=tuple(r[8:12]),
                              emissionStrength=r[12], smoothness=r[13], specularProbability=r[14], flag=int(r[15]))


def save_scene_npz(mgr: RayTracingManager, path: str):
    e = mgr.environmentSettings
    settings = dict(maxBounceCount=mgr.maxBounceCount, numRaysPerPixel=mgr.numRaysPerPixel,
                    defocusStrength=mgr.defocusStrength, divergeStrength=mgr.divergeStrength,
                    focusDistance=mgr.focusDistance, fieldOfView=mgr.camera.fieldOfView,
                    env=dict(enabled=bool(e.enabled), groundColour=list(e.groundColour), skyColourHorizon=list(e.skyColourHorizon),
                             skyColourZenith=list(e.skyColourZenith), sunFocus=e.sunFocus, sunIntensity=e.sunIntensity),
                    serialisedInfo=getattr(mgr, "serialisedInfo", None))
    arrays = dict(
        settings=np.frombuffer(json.dumps(settings).encode(), dtype=np.uint8),
        cam_pose=np.concatenate([mgr.camera.transform.position, mgr.camera.transform.rotation, mgr.camera.transform.lossyScale]).astype(np.float32),
        light_rot=np.asarray(mgr.light.rotation, np.float32),
        sphere_pose=np.array([np.concatenate([s.transform.position, s.transform.localScale]) for s in mgr.spheres], np.float32).reshape(-1, 6),
        sphere_mat=np.array([_mat_row(s.material) for s in mgr.spheres], np.float64).reshape(-1, 16),
    )
    mesh_pose, mesh_mat_rows, mesh_mat_range, chunk_rows, tris = [], [], [], [], []
    ntri = 0
    for mi, m in enumerate(mgr.meshes):
        mesh_pose.append(np.concatenate([m.transform.position, m.transform.rotation, m.transform.lossyScale]))
        mesh_mat_range.append([len(mesh_mat_rows), len(m.materials), m.triangleCount])
        mesh_mat_rows += [_mat_row(x) for x in m.materials]
        for c in m.localChunks:
            chunk_rows.append([mi, ntri, len(c.triangles), c.subMeshIndex, *c.bounds.center, *c.bounds.size])
            tris.append(c.triangles)
            ntri += len(c.triangles)
    arrays.update(
        mesh_pose=np.array(mesh_pose, np.float32).reshape(-1, 10),
        mesh_mat=np.array(mesh_mat_rows, np.float64).reshape(-1, 16),
        mesh_mat_range=np.array(mesh_mat_range, np.int64).reshape(-1, 3),
        chunk_table=np.array(chunk_rows, np.float64).reshape(-1, 10),
        local_tris=(np.concatenate(tris) if tris else np.zeros(0, TRIANGLE)).view(np.float32).reshape(-1, 18),
    )
    np.savez_compressed(path, **arrays)


def load_scene_npz(path: str, width: int = 1920, height: int = 1080, backend=None) -> RayTracingManager:
    z = np.load(path, allow_pickle=False)
    s = json.loads(bytes(z["settings"]).decode())
    cp = z["cam_pose"]
    cam = Camera(Transform(position=cp[0:3], rotation=cp[3:7], lossyScale=cp[7:10]), fieldOfView=s["fieldOfView"],
                 aspect=width / height)
    mgr = RayTracingManager(cam, Light(tuple(float(c) for c in z["light_rot"])), width, height, backend=backend)
    for k in ("maxBounceCount", "numRaysPerPixel", "defocusStrength", "divergeStrength", "focusDistance"):
        setattr(mgr, k, s[k])
    e = s["env"]
    mgr.environmentSettings = EnvironmentSettings(enabled=e["enabled"], groundColour=tuple(e["groundColour"]),
                                                  skyColourHorizon=tuple(e["skyColourHorizon"]),
                                                  skyColourZenith=tuple(e["skyColourZenith"]), sunFocus=e["sunFocus"],
                                                  sunIntensity=e["sunIntensity"])
    mgr.serialisedInfo = s.get("serialisedInfo")
    for pose, mat in zip(z["sphere_pose"], z["sphere_mat"]):
        mgr.spheres.append(RayTracedSphere(Transform(position=pose[0:3], lossyScale=pose[3:6], localScale=pose[3:6]), _row_mat(mat)))
    tris = np.ascontiguousarray(z["local_tris"]).view(TRIANGLE).reshape(-1)
    chunk_table = z["chunk_table"]
    for mi, (pose, (m0, mn, tc)) in enumerate(zip(z["mesh_pose"], z["mesh_mat_range"])):
        chunks = []
        for row in chunk_table[chunk_table[:, 0] == mi]:
            t0, tn = int(row[1]), int(row[2])
            chunks.append(MeshChunk(tris[t0:t0 + tn].copy(), Bounds(row[4:7].astype(np.float32), row[7:10].astype(np.float32)), int(row[3])))
        mgr.meshes.append(RayTracedMesh(Transform(position=pose[0:3], rotation=pose[3:7], lossyScale=pose[7:10]),
                                        [_row_mat(r) for r in z["mesh_mat"][m0:m0 + mn]], chunks, triangleCount=int(tc)))
    return mgr


# ---- writer: a RayTracingManager as a minimal Unity-YAML scene (the subset the loaders read) ---------------------------
def save_unity_scene(mgr: RayTracingManager, path: str):
    """Writes plain scene objects (no prefab instances): one GameObject + Transform per component, world poses as local
    poses of root objects.  Floats are written with repr() of the float32 value's shortest round-trip decimal."""
    def f(v):
        return repr(float(np.float32(v))) if not isinstance(v, float) else repr(v)

    def v3(n, v):
        return f"  {n}: {{x: {f(v[0])}, y: {f(v[1])}, z: {f(v[2])}}}"

    def col(c):
        c = list(c) + [1.0] * (4 - len(c))
        return "{r: %s, g: %s, b: %s, a: %s}" % tuple(repr(float(x)) for x in c)

    def mat_lines(m, ind):
        p = " " * ind
        return [f"{p}colour: {col(m.colour)}", f"{p}emissionColour: {col(m.emissionColour)}", f"{p}specularColour: {col(m.specularColour)}",
                f"{p}emissionStrength: {repr(float(m.emissionStrength))}", f"{p}smoothness: {repr(float(m.smoothness))}",
                f"{p}specularProbability: {repr(float(m.specularProbability))}", f"{p}flag: {int(m.flag)}"]

    out = ["%YAML 1.1", "%TAG !u! tag:unity3d.com,2011:"]
    next_id = [1000]

    def game_object(name, tr, local_scale=None):
        go, t = next_id[0], next_id[0] + 1
        next_id[0] += 10
        sc = tr.lossyScale if local_scale is None else local_scale
        out.extend([f"--- !u!1 &{go}", "GameObject:", f"  m_Name: {name}", "  m_IsActive: 1",
                    f"--- !u!4 &{t}", "Transform:", f"  m_GameObject: {{fileID: {go}}}",
                    "  m_LocalRotation: {x: %s, y: %s, z: %s, w: %s}" % tuple(f(c) for c in tr.rotation),
                    v3("m_LocalPosition", tr.position), v3("m_LocalScale", sc), "  m_Father: {fileID: 0}"])
        return go

    cam_go = game_object("Main Camera", mgr.camera.transform)
    e = mgr.environmentSettings
    out.extend([f"--- !u!114 &{next_id[0]}", "MonoBehaviour:", f"  m_GameObject: {{fileID: {cam_go}}}",
                f"  m_Script: {{fileID: 11500000, guid: {GUID_MANAGER}, type: 3}}",
                f"  maxBounceCount: {int(mgr.maxBounceCount)}", f"  numRaysPerPixel: {int(mgr.numRaysPerPixel)}",
                f"  defocusStrength: {repr(float(mgr.defocusStrength))}", f"  divergeStrength: {repr(float(mgr.divergeStrength))}",
                f"  focusDistance: {repr(float(mgr.focusDistance))}", "  environmentSettings:",
                f"    enabled: {1 if e.enabled else 0}", f"    groundColour: {col(e.groundColour)}",
                f"    skyColourHorizon: {col(e.skyColourHorizon)}", f"    skyColourZenith: {col(e.skyColourZenith)}",
                f"    sunFocus: {repr(float(e.sunFocus))}", f"    sunIntensity: {repr(float(e.sunIntensity))}",
                f"  numMeshChunks: {sum(len(m.localChunks) for m in mgr.meshes)}",
                f"  numTriangles: {sum(len(c.triangles) for m in mgr.meshes for c in m.localChunks)}"])
    next_id[0] += 10
    out.extend([f"--- !u!20 &{next_id[0]}", "Camera:", f"  m_GameObject: {{fileID: {cam_go}}}", f"  field of view: {repr(float(mgr.camera.fieldOfView))}"])
    next_id[0] += 10
    light_go = game_object("Directional Light", Transform(rotation=mgr.light.rotation))
    out.extend([f"--- !u!108 &{next_id[0]}", "Light:", f"  m_GameObject: {{fileID: {light_go}}}", "  m_Type: 1"])
    next_id[0] += 10
    for i, s in enumerate(mgr.spheres):
        go = game_object(f"Sphere ({i})", s.transform, s.transform.localScale)
        out.extend([f"--- !u!114 &{next_id[0]}", "MonoBehaviour:", f"  m_GameObject: {{fileID: {go}}}",
                    f"  m_Script: {{fileID: 11500000, guid: {GUID_SPHERE}, type: 3}}", "  material:"] + mat_lines(s.material, 4))
        next_id[0] += 10
    for i, m in enumerate(mgr.meshes):
        go = game_object(f"Mesh ({i})", m.transform)
        out.extend([f"--- !u!114 &{next_id[0]}", "MonoBehaviour:", f"  m_GameObject: {{fileID: {go}}}",
                    f"  m_Script: {{fileID: 11500000, guid: {GUID_MESH}, type: 3}}", "  materials:"])
        for mat in m.materials:
            ml = mat_lines(mat, 4)
            out.append("  - " + ml[0].strip())
            out.extend(ml[1:])
        out.extend([f"  triangleCount: {int(m.triangleCount)}", "  localChunks:"])
        for c in m.localChunks:
            out.append("  - triangles:")
            for t in c.triangles:
                first = True
                for k in TRIANGLE.names:
                    v = t[k]
                    out.append(("    - " if first else "      ") + f"{k}: {{x: {f(v[0])}, y: {f(v[1])}, z: {f(v[2])}}}")
                    first = False
            ext = (np.asarray(c.bounds.size, np.float32) * np.float32(0.5)).astype(np.float32)
            out.extend(["    bounds:", "    " + v3("m_Center", c.bounds.center).strip().join(["  ", ""]),
                        "    " + v3("m_Extent", ext).strip().join(["  ", ""]), f"    subMeshIndex: {int(c.subMeshIndex)}"])
        next_id[0] += 10
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
