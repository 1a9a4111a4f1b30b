#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage),
with the flags of the product build.  usage: tools/kernel_resources.py [rt_stream_kernels.hip|rt_api.hip] [-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g


def table(src, extra=()):
    flags = [f for f in g.HIPCC_FLAGS if f != "-shared"]
    if src.endswith("rt_stream_kernels.hip"):
        flags += g.STREAM_TU_FLAGS
    out = subprocess.run([g.shutil.which("hipcc") or "/opt/rocm/bin/hipcc", *flags, *extra, "-Rpass-analysis=kernel-resource-usage",
                          "-c", os.path.join(g.CSRC, src), "-o", "/dev/null"], capture_output=True, text=True)
    if out.returncode:
        sys.stderr.write(out.stderr)
        raise SystemExit(out.returncode)
    rows, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "rt_stream_kernels.hip"
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    for r in table(src, extra):
        print(f"{r['name'][:70]:70s} VGPR {r.get('VGPRs', -1):3d} SGPR {r.get('TotalSGPRs', -1):3d} scratch {r.get('ScratchSize', -1):4d} B "
              f"vspill {r.get('VGPRs Spill', -1):3d} sspill {r.get('SGPRs Spill', -1):3d} occ {r.get('Occupancy', -1)}")
