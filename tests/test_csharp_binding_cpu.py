"""The C# side is shipped as source (no C# toolchain in this image): these tests parse host_cs/RtNative.cs and check it
structurally against include/rt.h and the loaded library — every exported function has a DllImport, and every struct has the
same fields in the same order at the same offsets (C# LayoutKind.Sequential = natural alignment, like the C compiler) and the
size rt_sizeof() reports."""
import os
import re

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
             "RtLocalChunk": ("rt_local_chunk", rtx.LOCAL_CHUNK), "RtParams": ("rt_params", rtx.PARAMS), "RtStats": ("rt_stats", rtx.STATS)}
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
           "rt_mesh_transform": rtx.MESH_TRANSFORM, "rt_local_chunk": rtx.LOCAL_CHUNK, "rt_params": rtx.PARAMS, "rt_stats": rtx.STATS}
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


def test_manager_source_keeps_the_references_serialised_surface():
    """host_cs/RayTracingManager.cs: the class and the serialised field names the .unity scenes carry (Chess.unity:30174-30191),
    and every native call it makes exists in RtNative.cs."""
    text = open(os.path.join(CS, "RayTracingManager.cs")).read()
    assert re.search(r"public class RayTracingManager\s*:\s*MonoBehaviour", text) and "public const int TriangleLimit = 1500;" in text
    for field in ("maxBounceCount", "numRaysPerPixel", "defocusStrength", "divergeStrength", "focusDistance", "environmentSettings",
                  "useShaderInSceneView", "rayTracingShader", "accumulateShader", "numRenderedFrames", "numMeshChunks", "numTriangles"):
        assert re.search(r"\[SerializeField[^\]]*\]\s*\w+\s+" + field + r"\b", text), field
    native = open(os.path.join(CS, "RtNative.cs")).read()
    available = set(re.findall(r"static\s+(?:extern\s+)?[\w<>\[\]]+\s+(\w+)\s*[<(]", native))
    used = set(re.findall(r"RtNative\.(\w+)", text)) - {"cs"}            # ("RtNative.cs" in comments)
    assert used <= available | {"UploadCall"}, used - available
    assert "..." not in text and "// ..." not in text
