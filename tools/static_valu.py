#!/usr/bin/env python3
"""Static instruction counts of k_stream's regions, from the assembly of its code object.

k_stream is a persistent loop over a handful of regions (group fetch, SHADE pass with its hit / environment / camera blocks, traversal
burst with node steps, leaf phases and triangle tests).  The counting build of the kernel counts how often a wave executes each of them
(rt_stats.phaseExecs, rt_stats.schedExecs); this tool counts what one execution costs: it compiles the kernels' translation unit to
assembly with -DRT_MARKERS — rt_stream.hpp then emits an assembler comment at every region boundary, nothing else changes — and counts
the VALU / vector-memory / LDS / scalar instructions between the markers (innermost region wins; the blocks behind RT_COLD_PATH, the
IEEE fall-backs that never run on real inputs, are left out).  executions x static counts = the launch's VALU wave-instructions without
a profiler pass; bench.py puts that number next to the rocprofv3 figure (roofline.valu_model).

    python tools/static_valu.py            -> ray-tracing-extended_amd/static_valu.json (stamped with the hash of the kernel sources)

The unmarked compile is assembled as well: its total VALU count must equal the marked one's (the markers are comments in side-effect
asm statements; if they perturbed code generation the tool says so in `marker_drift`)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

CSRC = g.CSRC
OUT = os.path.join(ROOT, "ray-tracing-extended_amd", "static_valu.json")


def region_list():
    """the regions, in rt_stats.regionExecs order: csrc/rt_kernels.hpp RT_REGION_LIST"""
    text = open(os.path.join(CSRC, "rt_kernels.hpp")).read()
    body = re.search(r"#define RT_REGION_LIST\(X\)(.*?)\n(?!\s*X\()", text, re.S).group(1)
    return tuple(re.findall(r"X\((\w+)\)", body))


REGIONS = region_list()
OPTION_REGIONS = ("camera_dof", "camera_focus", "env_sun", "setup_spheres", "node_spill", "refill", "hit_sphere")


def csrc_sha16():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hpp", ".hip", ".cpp", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rt.h"), "rb").read())
    h.update(" ".join(list(g.HIPCC_FLAGS) + list(g.STREAM_TU_FLAGS)).encode())          # (as bench.csrc_sha16)
    return h.hexdigest()[:16]


def assemble(extra):
    hipcc = g.shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc, *flags, *g.STREAM_TU_FLAGS, *extra, "--offload-device-only", "-S",
                               os.path.join(CSRC, "rt_stream_kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def kind(mnemonic):
    if mnemonic.startswith("v_"):
        return "valu"
    if mnemonic.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mnemonic.startswith("ds_"):
        return "lds"
    if mnemonic.startswith("s_"):
        return "salu"
    return "other"


def kernels(text):
    """mangled name -> list of assembly lines of the function body"""
    out = {}
    for m in re.finditer(r"^(_ZN3rtk8k_stream\w+):", text, re.M):
        end = text.index(".Lfunc_end", m.end())
        out[m.group(1)] = text[m.end():end].split("\n")
    return out


DETAIL = None          # set to a dict to collect (region, block label) -> VALU instructions (debugging the attribution)


def count(lines, marked):
    """per region: instructions of the blocks that run on real inputs.  The function body is cut into basic blocks (labels and the
    assembler's `; %bb.N:` comments); a block that holds an RTCOLD / RTRARE comment is cold, and so is everything after it up to the label
    that the branch in front of it jumps to (a slow path may span several blocks: a division's expansion)."""
    blocks, cur = [], {"label": "entry", "lines": []}
    for ln in lines:
        t = ln.strip()
        if not t:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", t) or re.match(r"^; (%bb\.\d+):(.*)$", t)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "lines": [], "in_loop": "Loop" in m.group(2)}
            continue
        cur["lines"].append(t)
    blocks.append(cur)
    # cold blocks
    prev_skip = None
    for i, b in enumerate(blocks):
        if not b.get("cold") and any(t.startswith(("; RTCOLD", "; RTRARE")) for t in b["lines"]):
            b["cold"] = True
            # the skip branch that ends the block in front names the join label: cold up to it when it follows within a few blocks
            ahead = [blocks[j]["label"] for j in range(i + 1, min(i + 8, len(blocks)))]
            if prev_skip in ahead:
                for j in range(i + 1, i + 1 + ahead.index(prev_skip)):
                    blocks[j]["cold"] = True
        code = [t for t in b["lines"] if not t.startswith((";", "."))]
        m = re.match(r"^s_cbranch_exec\w*\s+(\.LBB\d+_\d+)", code[-1]) if code else None
        prev_skip = m.group(1) if m else None
    # option regions — code behind a branch that is uniform for the whole launch (depth of field, spheres, sun, no cached focus points, the
    # stack's spill path, pixel-by-pixel refill): the assembler comment can sit anywhere inside its block, so these are attributed by
    # whole blocks, from the block that holds the `begin` comment to the label the branch in front of that block jumps to
    prev_skip = None
    for i, b in enumerate(blocks):
        for t in b["lines"]:
            m = re.match(r"^; RTMARK begin (\w+)", t)
            if m and m.group(1) in OPTION_REGIONS and marked and "option" not in b:
                b["option"] = m.group(1)
                ahead = [blocks[j]["label"] for j in range(i + 1, min(i + 14, len(blocks)))]
                if prev_skip in ahead:
                    for j in range(i + 1, i + 1 + ahead.index(prev_skip)):
                        blocks[j].setdefault("option", m.group(1))
        code = [t for t in b["lines"] if not t.startswith((";", "."))]
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", code[-1]) if code else None
        prev_skip = m.group(1) if m else None
    regions = {r: {"valu": 0, "vmem": 0, "lds": 0, "salu": 0, "other": 0} for r in REGIONS}
    cold = {"valu": 0, "vmem": 0, "lds": 0, "salu": 0, "other": 0}
    # The region a block starts in follows the CONTROL FLOW, not the layout (the compiler places a region's blocks wherever it likes: the
    # leaf loop's accept path sits in front of the group fetch in one build and behind it in the next).  Successors: every branch target
    # of the block plus the next block unless the block ends in s_branch / s_endpgm; the marker stack at a block's exit is handed to its
    # successors.
    index = {b["label"]: i for i, b in enumerate(blocks)}
    succ = []
    for i, b in enumerate(blocks):
        code = [t for t in b["lines"] if not t.startswith((";", "."))]
        out = [index[m.group(1)] for t in code for m in [re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", t)] if m and m.group(1) in index]
        last = code[-1].split()[0] if code else ""
        if last not in ("s_branch", "s_endpgm", "s_setpc_b64") and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)

    def walk(b, stack, tally):
        stack = list(stack)
        # a block that is entered only by falling through a conditional branch (`; %bb.N:`) and whose first marker opens a region is the
        # body of the `if` that the region starts: the instructions the scheduler placed in front of the comment are the region's too
        marks = [re.match(r"^; RTMARK (begin|end) (\w+)", t) for t in b["lines"] if t.startswith("; RTMARK")]
        first = marks[0] if marks else None
        lead = first.group(2) if marked and first and first.group(1) == "begin" and b["label"].startswith("%bb") else None
        # a block whose only comments are `begin X` ... `end X` is the whole body of the `if` that X wraps (the scheduler moves the
        # body's arithmetic across both comments: they have no operands): every instruction of the block is X's
        whole = (marks[0].group(2) if marked and len(marks) == 2 and marks[0].group(1) == "begin" and marks[1].group(1) == "end"
                 and marks[0].group(2) == marks[1].group(2) and not b.get("option") else None)
        for t in b["lines"]:
            m = re.match(r"^; RTMARK (begin|end) (\w+)", t)
            if m and marked:
                if m.group(1) == "begin":
                    stack.append(m.group(2)); lead = None
                elif len(stack) > 1 and stack[-1] == m.group(2):
                    stack.pop()
                elif m.group(2) in stack[1:]:
                    if tally:
                        problems.append(f"end {m.group(2)} while in {stack[-1]}")
                    while stack[-1] != m.group(2):
                        stack.pop()
                    stack.pop()
                elif tally:
                    problems.append(f"end {m.group(2)} outside it (in {stack[-1]})")
                continue
            if not tally or t.startswith((";", ".")):
                continue
            k = kind(t.split()[0])
            r = b.get("option") or whole or lead or stack[-1]
            if r == "loop" and not b.get("in_loop"):            # straight-line code outside the persistent loop: before it or after it
                r = "epilogue" if b.get("after_loop") else "prologue"
            (cold if b.get("cold") else regions[r])[k] += 1
            if k == "valu" and DETAIL is not None:
                key = ("cold" if b.get("cold") else r, b["label"])
                DETAIL[key] = DETAIL.get(key, 0) + 1
        return tuple(stack)

    problems, seen_loop = [], False
    for b in blocks:
        b["after_loop"] = seen_loop
        seen_loop = seen_loop or bool(b.get("in_loop"))
    def common(a, c):
        n = 0
        while n < min(len(a), len(c)) and a[n] == c[n]:
            n += 1
        return a[:max(n, 1)]

    # entry state of a block = the common prefix of its predecessors' exit states (a `break` leaves a region without passing its end
    # comment: at the join the region is over), to a fixed point
    entry = {0: ("loop",)}
    changed = True
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            if i not in entry:
                continue
            b["exit"] = walk(b, entry[i], False)
            for j in succ[i]:
                new = b["exit"] if j not in entry else common(entry[j], b["exit"])
                if entry.get(j) != new:
                    entry[j] = new; changed = True
    for i, b in enumerate(blocks):
        walk(b, entry.get(i, ("loop",)), True)
    return regions, cold, problems


def main():
    marked, plain = assemble(["-DRT_MARKERS"]), assemble([])
    km, kp = kernels(marked), kernels(plain)
    table = {"csrc_sha16": csrc_sha16(), "regions": list(REGIONS),
             "note": "static instruction counts per region of k_stream (exclusive: the innermost region owns an instruction; cold = blocks behind "
                     "RT_COLD_PATH, never executed on real inputs); template arguments <COUNT, PHILOX, H, TRI>",
             "kernels": {}}
    for name, lines in km.items():
        targs = re.search(r"k_streamILb(\d)ELb(\d)ELb(\d)ELb(\d)E", name).groups()
        if targs[0] == "1":
            continue                             # the counting instantiations carry the counters' own instructions
        regions, cold, problems = count(lines, True)
        pregions, pcold, _ = count(kp[name], False)
        total_marked = sum(r["valu"] for r in regions.values()) + cold["valu"]
        total_plain = sum(r["valu"] for r in pregions.values()) + pcold["valu"]
        key = "k_stream<false,%s,%s,%s>" % tuple("true" if a == "1" else "false" for a in targs[1:])
        table["kernels"][key] = {"valu": {r: regions[r]["valu"] for r in REGIONS}, "vmem": {r: regions[r]["vmem"] for r in REGIONS},
                                 "lds": {r: regions[r]["lds"] for r in REGIONS}, "salu": {r: regions[r]["salu"] for r in REGIONS},
                                 "cold_valu": cold["valu"], "total_valu": total_marked, "total_valu_without_markers": total_plain,
                                 "marker_drift": total_marked - total_plain, "marker_problems": problems}
    json.dump(table, open(OUT, "w"), indent=1)
    for key, e in table["kernels"].items():
        print(key, "VALU", e["valu"], "| cold", e["cold_valu"], "| total", e["total_valu"], "(unmarked:", e["total_valu_without_markers"], ")", e["marker_problems"][:2])


if __name__ == "__main__":
    main()
