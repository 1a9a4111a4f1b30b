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


# ---- the compiled host's MeshSplitter (host_cpp/rt_host.cpp) ------------------------------------------------------------
def _mesh_from_triangles(tris):
    """An un-indexed mesh: three vertices per triangle in order (vertex 3i+k = corner k of triangle i)."""
    v = np.stack([tris["posA"], tris["posB"], tris["posC"]], axis=1).reshape(-1, 3)
    n = np.stack([tris["normalA"], tris["normalB"], tris["normalC"]], axis=1).reshape(-1, 3)
    return v, n, np.arange(len(v), dtype=np.int32)


def _same_chunks(cpp, py):
    assert len(cpp) == len(py)
    for (t, c, sz, sub), ch in zip(cpp, py):
        assert t.tobytes() == ch.triangles.tobytes()
        assert np.array_equal(c, np.asarray(ch.bounds.center, np.float32)) and np.array_equal(sz, np.asarray(ch.bounds.size, np.float32))
        assert sub == ch.subMeshIndex


def test_cpp_meshsplitter_equals_python_on_procedural_meshes(rtx):
    """MeshSplitter::CreateChunks (C++) == MeshSplitter.CreateChunks (Python): same chunks, same order, same bytes — an indexed
    mesh with shared vertices and two sub-meshes, and a 1,536-triangle sphere that needs several split levels."""
    from rtx_amd.host import Mesh, MeshSplitter
    from rtx_amd.host_cpp_binding import cpp_split_mesh
    tris = rtx.scenes.uv_sphere_triangles(24, 32)
    v, n, idx = _mesh_from_triangles(tris)
    half = (len(idx) // 6) * 3
    for sub in ([(0, len(idx))], [(0, half), (half, len(idx) - half)]):
        py = MeshSplitter.CreateChunks(Mesh(v, n, idx, sub))
        _same_chunks(cpp_split_mesh(v, n, idx, sub), py)
        assert sum(len(c.triangles) for c in py) == len(tris) and max(len(c.triangles) for c in py) <= 48
    # shared vertices: a cube as 8 vertices + 36 indices, scaled so that it must split (49+ triangles: 5 cubes in one mesh)
    cube = rtx.scenes.cube_triangles()
    allt = np.concatenate([cube.copy() for _ in range(5)])
    for k in range(5):
        for f in ("posA", "posB", "posC"):
            allt[f][12 * k:12 * (k + 1)] += np.float32([1.5 * k, 0.1 * k, -0.7 * k])
    pts = np.stack([allt["posA"], allt["posB"], allt["posC"]], axis=1).reshape(-1, 3)
    uniq, inv = np.unique(pts, axis=0, return_inverse=True)
    nrm = np.zeros_like(uniq); nrm[:, 1] = 1
    py = MeshSplitter.CreateChunks(Mesh(uniq, nrm, inv.astype(np.int32)))
    _same_chunks(cpp_split_mesh(uniq, nrm, inv.astype(np.int32), [(0, len(inv))]), py)
    assert len(py) > 1


def test_cpp_meshsplitter_reproduces_the_references_serialised_chunks(rtx):
    """Knight, both Suzanne sub-meshes, cube and quad: the compiled MeshSplitter reproduces what the reference serialised into
    its scenes bit for bit, from the seed vertex the Python restatement finds (same search as above, run once per mesh)."""
    from rtx_amd.host import MeshSplitter
    from rtx_amd.host_cpp_binding import cpp_split_mesh
    checked = 0
    for name, mesh in _distinct_meshes(rtx):
        if name in ("Chess", "Thumbnail"):
            continue
        subs = {}
        for c in mesh.localChunks:
            subs.setdefault(c.subMeshIndex, []).append(c)
        for idx, chunks in subs.items():
            tris = np.concatenate([c.triangles for c in chunks])
            v, n, ib = _mesh_from_triangles(tris)
            seed = None
            for v0 in np.unique(v, axis=0):
                out = []
                MeshSplitter.Split(MeshSplitter.CreateSubMesh(tris, idx, firstVertex=v0), out)
                if len(out) == len(chunks) and all(a.triangles.tobytes() == b.triangles.tobytes() and np.array_equal(a.bounds.center, b.bounds.center)
                                                   and np.array_equal(a.bounds.size, b.bounds.size) for a, b in zip(out, chunks)):
                    seed = v0
                    break
            assert seed is not None, (name, mesh.triangleCount, idx)
            got = cpp_split_mesh(v, n, ib, [(0, len(ib))], mode=2, seed=seed)
            assert len(got) == len(chunks)
            for (t, c, sz, _), ch in zip(got, chunks):
                assert t.tobytes() == ch.triangles.tobytes()
                assert np.array_equal(c, np.asarray(ch.bounds.center, np.float32)) and np.array_equal(sz, np.asarray(ch.bounds.size, np.float32))
            checked += len(chunks)
    assert checked >= 100, checked


def test_getsubmeshes_splits_a_mesh_that_has_no_cached_chunks(rtx):
    """RayTracedMesh.cs:24-29: a mesh without serialised localChunks is split on first use — in both hosts, with the same world
    chunks (triangles, tight bounds) — and the 1500-triangle limit is the reference's exception in both."""
    from rtx_amd.host import Mesh, RayTracedMesh, RayTracingMaterial, Transform
    from rtx_amd.host_cpp_binding import cpp_split_mesh
    tris = rtx.scenes.uv_sphere_triangles(12, 16)
    v, n, idx = _mesh_from_triangles(tris)
    tf = Transform(position=(1.0, 2.0, -3.0), rotation=(0.1825742, 0.3651484, 0.5477226, 0.7302967), lossyScale=(2.0, 0.5, 1.5))
    rm = RayTracedMesh(tf, [RayTracingMaterial()], None, sharedMesh=Mesh(v, n, idx))
    world = rm.GetSubMeshes()
    assert rm.localChunks and rm.mesh is rm.sharedMesh and rm.triangleCount == len(tris)
    assert rm.GetSubMeshes() is not None and len(rm.localChunks) == len(world)          # cached: no second split
    t10 = np.concatenate([tf.position, tf.rotation, tf.lossyScale]).astype(np.float32)
    _same_chunks(cpp_split_mesh(v, n, idx, [(0, len(idx))], mode=1, transform=t10), world)
    big = rtx.scenes.uv_sphere_triangles(28, 32)
    bv, bn, bi = _mesh_from_triangles(big)
    assert len(big) > 1500
    with pytest.raises(Exception, match="fewer than 1500"):
        # the limit is checked on the cached mesh: first use splits, the second call throws (mesh.triangles.Length / 3, :19)
        r = RayTracedMesh(tf, [RayTracingMaterial()], None, sharedMesh=Mesh(bv, bn, bi), triangleCount=0)
        r.GetSubMeshes(); r.GetSubMeshes()
