"""MeshSplitter restatement (host.py) against the reference's own serialised outputs.

Every RayTracedMesh in the .unity scenes carries `localChunks` = the output of MeshSplitter.CreateChunks
(Assets/Scripts/Helpers/MeshSplitter.cs:11-33) for its mesh.  The original index-buffer order is not recoverable, but
chunk membership and order do not depend on it (a triangle goes to the first octant, in loop order, that contains one of
its vertices; inside a chunk the input order is kept), so feeding the concatenated chunk triangles back must reproduce
the serialised chunks — triangles, order and bounds (centre / extent floats) bit for bit — for SOME first vertex (the
0.01-sized seed box of CreateSubMesh, :39, sits at the mesh's first vertex and can widen the root bounds)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def _distinct_meshes(rtx):
    from rtx_amd import unity_scene
    seen, out = set(), []
    for name in ("Reflective_Balls", "Knight", "Suzanne", "Chess", "Thumbnail"):
        m = unity_scene.load_scene_npz(os.path.join(GOLDEN, name + ".npz"))
        for mesh in m.meshes:
            key = (mesh.triangleCount, len(mesh.localChunks),
                   np.concatenate([c.triangles for c in mesh.localChunks]).tobytes()[:4096])
            if key not in seen:
                seen.add(key)
                out.append((name, mesh))
    return out


def _reproduces(MeshSplitter, chunks, idx):
    tris = np.concatenate([c.triangles for c in chunks])
    cands = np.unique(np.concatenate([tris["posA"], tris["posB"], tris["posC"]]), axis=0)
    for v0 in cands:
        out = []
        MeshSplitter.Split(MeshSplitter.CreateSubMesh(tris, idx, firstVertex=v0), out)
        if len(out) != len(chunks):
            continue
        if all(a.triangles.tobytes() == b.triangles.tobytes() and np.array_equal(a.bounds.center, b.bounds.center)
               and np.array_equal(a.bounds.size, b.bounds.size) for a, b in zip(out, chunks)):
            return True
    return False


def test_meshsplitter_reproduces_serialised_meshes(rtx):
    """Knight (456 triangles -> 32 chunks), Suzanne (968 triangles, two sub-meshes -> 6 + 60 chunks) and the cube / quad
    meshes reproduce bit for bit.  The three chess pieces (pawn/king/queen, coordinates ~1e-3 so the 0.01 seed box dominates
    the root bounds) do not from the serialised data alone: Bounds.Encapsulate re-derives centre/extents after every point,
    so the root bounds depend on the lost original vertex order — for them only the invariants are checked."""
    from rtx_amd.host import MeshSplitter
    meshes = _distinct_meshes(rtx)
    assert len(meshes) >= 7
    exact_chunks, order_dependent = 0, []
    for name, mesh in meshes:
        subs = {}
        for c in mesh.localChunks:
            subs.setdefault(c.subMeshIndex, []).append(c)
        for idx, chunks in subs.items():
            tris = np.concatenate([c.triangles for c in chunks])
            assert max(len(c.triangles) for c in chunks) <= MeshSplitter.maxTrisPerChunk          # MeshSplitter.cs:9
            if name in ("Chess", "Thumbnail") and mesh.triangleCount in (358, 582, 466):
                order_dependent.append((name, mesh.triangleCount))
                out = []
                MeshSplitter.Split(MeshSplitter.CreateSubMesh(tris, idx), out)
                # (a vertex on the outer face of the root box can fall outside all eight float-rounded octants: the algorithm
                # itself can drop such triangles; whether it does depends on the unpinned rounding of Bounds.Contains)
                assert 0.95 * len(tris) <= sum(len(c.triangles) for c in out) <= len(tris) and max(len(c.triangles) for c in out) <= 48
                continue
            if name == "Thumbnail":
                continue                                         # same meshes as Knight / Reflective Balls
            assert _reproduces(MeshSplitter, chunks, idx), f"{name}: mesh with {mesh.triangleCount} triangles, sub-mesh {idx}"
            exact_chunks += len(chunks)
    assert exact_chunks >= 100, exact_chunks


def test_meshsplitter_invariants_on_a_procedural_mesh(rtx):
    """<= 48 triangles per leaf unless depth 6 is reached, every triangle exactly once, bounds contain the vertices."""
    from rtx_amd.host import MeshSplitter
    tris = rtx.scenes.uv_sphere_triangles(24, 32)
    assert len(tris) > 1400
    chunks = MeshSplitter.CreateChunks([(tris, 0)])
    assert sum(len(c.triangles) for c in chunks) == len(tris)
    assert max(len(c.triangles) for c in chunks) <= 48
    got = np.sort(np.concatenate([c.triangles for c in chunks]).view(np.uint8).reshape(len(tris), -1), axis=0)
    want = np.sort(tris.view(np.uint8).reshape(len(tris), -1), axis=0)
    assert np.array_equal(got, want)
    for c in chunks:
        pts = np.concatenate([c.triangles["posA"], c.triangles["posB"], c.triangles["posC"]])
        assert np.all(pts >= c.bounds.min - 1e-6) and np.all(pts <= c.bounds.max + 1e-6)
