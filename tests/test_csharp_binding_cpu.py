"""The C# side is shipped as source (no C# toolchain in this image): these tests parse host_cs/RtNative.cs and check it
structurally against include/rt.h and the loaded library — every exported function has a DllImport, and every struct has the
same fields in the same order at the same offsets (C# LayoutKind.Sequential = natural alignment, like the C compiler) and the
size rt_sizeof() reports."""
import os
import re
import sys

import pytest

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "ray-tracing-extended_amd", "host_cs")
PRIM = {"int": 4, "uint": 4, "float": 4, "double": 8, "ulong": 8, "long": 8, "byte": 1}


def _cs_structs(text):
    """{name: [(field, type, count)]} for every `struct` in the file (fields: `public T x;`, `public T a, b;`, `public fixed T x[N];`)."""
    out = {}
    for m in re.finditer(r"public\s+(?:unsafe\s+)?struct\s+(\w+)[^{]*\{(.*?)\n    \}", text, re.S):
        fields = []
        for line in m.group(2).splitlines():
            line = line.split("//")[0].strip()
            if not line.startswith("public"):
                continue
            f = re.match(r"public\s+fixed\s+(\w+)\s+(\w+)\[(\d+)\];", line)
            if f:
                fields.append((f.group(2), f.group(1), int(f.group(3))))
                continue
            f = re.match(r"public\s+(\w+)\s+([\w\s,]+);", line)
            assert f, line
            for name in f.group(2).split(","):
                fields.append((name.strip(), f.group(1), 1))
        out[m.group(1)] = fields
    return out


def _layout(structs, name):
    """Sequential layout with natural alignment -> ([(field, offset, bytes)], size, alignment)."""
    off, align, rows = 0, 1, []
    for field, typ, count in structs[name]:
        if typ in PRIM:
            size, a = PRIM[typ], PRIM[typ]
        else:
            _, size, a = _layout(structs, typ)
        off = (off + a - 1) // a * a
        rows.append((field, off, size * count))
        off += size * count
        align = max(align, a)
    return rows, (off + align - 1) // align * align, align


def test_every_export_of_rt_h_has_a_dllimport(rtx):
    text = open(os.path.join(CS, "RtNative.cs")).read()
    imported = set(re.findall(r"static\s+extern\s+\w+\s+(rt_\w+)\s*\(", text))
    header = open(os.path.join(ROOT, "include", "rt.h")).read()
    declared = set(re.findall(r"\b(rt_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    assert declared == set(rtx._cabi.SYMBOLS), declared ^ set(rtx._cabi.SYMBOLS)
    assert imported == declared, (sorted(declared - imported), sorted(imported - declared))
    assert text.count("[DllImport(Lib") == len(imported)


def test_csharp_struct_layouts_match_the_c_abi(rtx):
    structs = _cs_structs(open(os.path.join(CS, "RtNative.cs")).read())
    lib = rtx.load_library()
    pairs = {"RtMaterial": ("rt_material", rtx.MATERIAL), "RtSphere": ("rt_sphere", rtx.SPHERE), "RtTriangle": ("rt_triangle", rtx.TRIANGLE),
             "RtMeshInfo": ("rt_meshinfo", rtx.MESHINFO), "RtMeshTransform": ("rt_mesh_transform", rtx.MESH_TRANSFORM),
             "RtLocalChunk": ("rt_local_chunk", rtx.LOCAL_CHUNK), "RtParams": ("rt_params", rtx.PARAMS), "RtStats": ("rt_stats", rtx.STATS),
             "RtMultiInfo": ("rt_multi_info", rtx.MULTI_INFO)}
    assert set(structs) == set(pairs)
    for cs_name, (c_name, dt) in pairs.items():
        rows, size, _ = _layout(structs, cs_name)
        assert size == lib.rt_sizeof(c_name.encode()) == dt.itemsize, (cs_name, size, dt.itemsize)
        assert [r[0] for r in rows] == list(dt.names), (cs_name, [r[0] for r in rows], dt.names)
        for field, off, nbytes in rows:
            assert off == dt.fields[field][1], (cs_name, field, off, dt.fields[field][1])
            assert nbytes == dt.fields[field][0].itemsize, (cs_name, field, nbytes)


def test_c_header_field_order_matches_the_binding(rtx):
    """include/rt.h itself: field names of each typedef struct, in order, equal the numpy layouts the tests and the C# file use."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rt.h")).read(), flags=re.S)
    dts = {"rt_material": rtx.MATERIAL, "rt_sphere": rtx.SPHERE, "rt_triangle": rtx.TRIANGLE, "rt_meshinfo": rtx.MESHINFO,
           "rt_mesh_transform": rtx.MESH_TRANSFORM, "rt_local_chunk": rtx.LOCAL_CHUNK, "rt_params": rtx.PARAMS, "rt_stats": rtx.STATS,
           "rt_multi_info": rtx.MULTI_INFO}
    for name, dt in dts.items():
        body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + ";", header, re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"^(?:const\s+)?\w+\s+", "", decl)            # drop the type
            names += [re.sub(r"\[.*?\]", "", d).strip() for d in decl.split(",")]
        assert names == list(dt.names), (name, names, dt.names)


def test_backend_class_only_calls_what_the_binding_declares():
    """host_cs/RtBackend.cs (the other side of the reference's material-property boundary): every native call exists in RtNative.cs."""
    text = open(os.path.join(CS, "RtBackend.cs")).read()
    native = open(os.path.join(CS, "RtNative.cs")).read()
    available = set(re.findall(r"static\s+(?:extern\s+)?[\w<>\[\]]+\s+(\w+)\s*[<(]", native))
    used = set(re.findall(r"RtNative\.(\w+)", text)) - {"cs"}            # ("RtNative.cs" in comments)
    assert used and used <= available | {"UploadCall"}, used - available
    assert "..." not in text


def test_manager_patch_applies_to_the_reference_and_keeps_its_serialised_surface():
    """host_cs/RayTracingManager.cs.ed: the line edits a maintainer applies to the reference's own RayTracingManager.cs
    (`patch -e`, or tools/apply_ed.py).  An ed script carries line numbers and the new lines only — none of the reference's text.
    Applied to the reference file (read here, on the build machine only), the result keeps the class, every serialised field the .unity
    scenes carry (Chess.unity:30174-30191) and the scene-building code, calls only methods RtBackend has, and no longer touches a
    Material, a ComputeBuffer or ShaderHelper."""
    ref = "/root/reference/Assets/Scripts/RayTracingManager.cs"
    if not os.path.exists(ref):
        pytest.skip("the reference project is not on this machine")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import apply_ed
    script = open(os.path.join(CS, "RayTracingManager.cs.ed")).read().splitlines(True)
    text = "".join(apply_ed.apply_ed(open(ref, newline="").read().replace("\r\n", "\n").splitlines(True), script))
    assert re.search(r"public class RayTracingManager\s*:\s*MonoBehaviour", text) and "public const int TriangleLimit = 1500;" in text
    for field in ("maxBounceCount", "numRaysPerPixel", "defocusStrength", "divergeStrength", "focusDistance", "environmentSettings",
                  "useShaderInSceneView", "rayTracingShader", "accumulateShader", "numRenderedFrames", "numMeshChunks", "numTriangles",
                  "devices", "literalChunkCull", "philox", "deviceGeometry"):
        assert re.search(r"\[SerializeField[^\]]*\]\s*[\w\[\]]+\s+" + field + r"\b", text), field
    assert text.count("{") == text.count("}")
    for gone in ("rayTracingMaterial", "accumulateMaterial", "ComputeBuffer", "ShaderHelper", "RenderTexture.GetTemporary"):
        assert gone not in text, gone
    for kept in ("FindObjectsOfType<RayTracedMesh>()", "FindObjectsOfType<RayTracedSphere>()", "GetSubMeshes()", "OnValidate"):
        assert kept in text, kept
    backend = open(os.path.join(CS, "RtBackend.cs")).read()
    methods = set(re.findall(r"public\s+(?:unsafe\s+)?[\w<>\[\]]+\s+(\w+)\s*[<(]", backend))
    called = set(re.findall(r"backend\??\.(\w+)\(", text))
    assert called and called <= methods, called - methods
    # the script itself holds no line of the reference
    ref_lines = {l.strip() for l in open(ref).read().splitlines() if len(l.strip()) > 12}
    new_lines = [l.strip() for l in script if not re.fullmatch(r"\d+(,\d+)?[acd]\n|\.\n", l)]
    assert not [l for l in new_lines if l in ref_lines]
    # the on-device geometry pipeline is reached from CreateMeshes, before the host-side transform loop
    assert text.index("backend.SetMeshObjects(meshObjects)") < text.index("GetSubMeshes()")


def test_mesh_component_patch_adds_the_local_chunk_accessor():
    """host_cs/RayTracedMesh.cs.ed: one added method, GetLocalChunks(), which RtBackend.SetMeshObjects calls; the component's serialised
    fields and its own methods stay as they are, and the script holds no line of the reference."""
    ref = "/root/reference/Assets/Scripts/Render Types/RayTracedMesh.cs"
    if not os.path.exists(ref):
        pytest.skip("the reference project is not on this machine")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import apply_ed
    script = open(os.path.join(CS, "RayTracedMesh.cs.ed")).read().splitlines(True)
    before = open(ref, newline="").read().replace("\r\n", "\n")
    text = "".join(apply_ed.apply_ed(before.splitlines(True), script))
    assert text.count("{") == text.count("}")
    assert "public MeshChunk[] GetLocalChunks()" in text and text.index("public MeshChunk[] GetSubMeshes()") < text.index("GetLocalChunks()")
    for kept in ("[SerializeField] MeshChunk[] localChunks;", "void UpdateWorldChunkFromLocal(", "public RayTracingMaterial GetMaterial(int subMeshIndex)"):
        assert kept in text, kept
    removed = [l for l in before.splitlines() if l.strip() and l not in text.splitlines()]
    assert not removed, removed
    ref_lines = {l.strip() for l in before.splitlines() if len(l.strip()) > 12}
    new_lines = [l.strip() for l in script if not re.fullmatch(r"\d+(,\d+)?[acd]\n|\.\n", l)]
    assert not [l for l in new_lines if l in ref_lines]
    backend = open(os.path.join(CS, "RtBackend.cs")).read()
    assert ".GetLocalChunks()" in backend
