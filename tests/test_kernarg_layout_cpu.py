"""k_stream reads its arguments through a struct view of its own kernarg segment (csrc/rt_kernels.hpp fresh_kernargs,
csrc/rt_stream.hpp StreamKernArgs = {DeviceScene, FrameArgs, StreamArgs}).  That view is only right while the compiler lays the
three by-value arguments out like the members of that struct.  This test reads the code object inside the built library (no GPU)
and checks the argument offsets and sizes in the metadata of every k_stream variant against the struct rule."""
import os
import re
import shutil
import struct

import msgpack
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ray-tracing-extended_amd", "librt_mi355x.so")


def code_objects(blob):
    """the gfx950 ELF images of every offload bundle in the library"""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    for m in re.finditer(magic, blob):
        base = m.start()
        n, = struct.unpack_from("<Q", blob, base + len(magic))
        pos = base + len(magic) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, pos)
            ident = blob[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if "gfx950" in ident and size:
                yield blob[base + off:base + off + size]


def kernel_metadata(elf):
    """NT_AMDGPU_METADATA (msgpack) of an AMDGPU ELF: list of kernel descriptions"""
    assert elf[:4] == b"\x7fELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:                                 # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz]
            desc_at = p + 12 + ((namesz + 3) & ~3)
            if ntype == 32 and name.startswith(b"AMDGPU"):
                return msgpack.unpackb(elf[desc_at:desc_at + descsz], raw=False)["amdhsa.kernels"]
            p = desc_at + ((descsz + 3) & ~3)
    return []


def test_k_trace_kernarg_segment_is_laid_out_like_the_struct_view():
    """k_trace reads the per-tile part of its arguments through TraceKernArgs = {DeviceScene, FrameArgs}"""
    if not os.path.exists(LIB):
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.fail("librt_mi355x.so is not built although hipcc is present: run __graft_entry__.build()")
        pytest.skip("library not built and no hipcc to build it with")
    blob = open(LIB, "rb").read()
    seen = 0
    for elf in code_objects(blob):
        for k in kernel_metadata(elf):
            if "k_trace" not in k[".name"]:
                continue
            args = [a for a in k[".args"] if a[".value_kind"] == "by_value"]
            assert len(args) == 2, k[".name"]
            assert args[0][".offset"] == 0 and args[1][".offset"] == (args[0][".size"] + 7) & ~7, (k[".name"], args)
            seen += 1
    assert seen >= 7                                       # the flat twin + COUNT x H + COUNT x the six-wave sphere instantiation


def test_k_stream_kernarg_segment_is_laid_out_like_the_struct_view():
    if not os.path.exists(LIB):
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.fail("librt_mi355x.so is not built although hipcc is present: run __graft_entry__.build()")
        pytest.skip("library not built and no hipcc to build it with")
    blob = open(LIB, "rb").read()
    seen = 0
    for elf in code_objects(blob):
        for k in kernel_metadata(elf):
            if "k_stream" not in k[".name"]:
                continue
            args = [a for a in k[".args"] if a[".value_kind"] == "by_value"]
            assert len(args) == 3, k[".name"]
            pos = 0
            for a in args:                                # struct rule: members in order, each aligned to 8 (static_asserts: alignof <= 8)
                pos = (pos + 7) & ~7
                assert a[".offset"] == pos, (k[".name"], a)
                pos += a[".size"]
            assert k[".sgpr_spill_count"] == 0, (k[".name"], "k_stream spills SGPRs again: see fresh_kernargs")
            seen += 1
    assert seen == 12                                      # COUNT x PHILOX x H, and COUNT x PHILOX without triangles
