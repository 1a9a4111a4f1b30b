#!/usr/bin/env python3
"""A/B builds: tools/build_variant.py <name> [hipcc flags, e.g. -DRT_PROBE_VALU=13]  ->  ab_libs/librt_<name>.so (same ABI as the product
library; selected at run time through RTX_LIB, see tools/ab_libs.sh).  Builds from the working tree with the product's flags;
--drop=<flag> removes one of them first (an `-mllvm` option goes together with its `-mllvm`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g


def build(name, extra):
    out_dir = os.path.join(ROOT, "ab_libs")
    obj_dir = os.path.join(out_dir, "_obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = g.shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs, _ = g.hip_sources()
    cflags = [f for f in g.HIPCC_FLAGS if f != "-shared"]
    for d in [e[len("--drop="):] for e in extra if e.startswith("--drop=")]:
        i = cflags.index(d)
        del cflags[i - 1 if i and cflags[i - 1] == "-mllvm" else i:i + 1]
    extra = [e for e in extra if not e.startswith("--drop=")]
    objs, procs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        tu = g.STREAM_TU_FLAGS if src.endswith("rt_stream_kernels.hip") else []
        procs.append(subprocess.Popen([hipcc, *cflags, *tu, *extra, "-c", src, "-o", obj]))
        objs.append(obj)
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("hipcc failed")
    lib = os.path.join(out_dir, f"librt_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", lib])
    print(lib)


if __name__ == "__main__":
    build(sys.argv[1], sys.argv[2:])
