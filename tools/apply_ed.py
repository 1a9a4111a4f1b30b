#!/usr/bin/env python3
"""Apply an ed script of the kind `diff -e` writes (commands a / c / d, highest line numbers first) to a text file:
    python tools/apply_ed.py Assets/Scripts/RayTracingManager.cs ray-tracing-extended_amd/host_cs/RayTracingManager.cs.ed [-o out.cs]
(the same as `patch -e FILE < SCRIPT`, for machines without ed).  An ed script names the replaced lines by number only: it carries
none of the text it replaces."""
import re
import sys


def apply_ed(text_lines, script_lines):
    out = list(text_lines)
    i = 0
    while i < len(script_lines):
        m = re.fullmatch(r"(\d+)(?:,(\d+))?([acd])", script_lines[i].rstrip("\n"))
        if not m:
            raise ValueError(f"line {i + 1} of the script: not an ed command: {script_lines[i]!r}")
        first, last, cmd = int(m.group(1)), int(m.group(2) or m.group(1)), m.group(3)
        i += 1
        new = []
        if cmd in "ac":
            while script_lines[i].rstrip("\n") != ".":
                new.append(script_lines[i] if script_lines[i].endswith("\n") else script_lines[i] + "\n")
                i += 1
            i += 1
        if cmd == "a":
            out[first:first] = new
        elif cmd == "c":
            out[first - 1:last] = new
        else:
            del out[first - 1:last]
    return out


if __name__ == "__main__":
    target, script = sys.argv[1], sys.argv[2]
    dest = sys.argv[sys.argv.index("-o") + 1] if "-o" in sys.argv else target
    res = apply_ed(open(target, newline="").read().replace("\r\n", "\n").splitlines(True), open(script).read().splitlines(True))
    open(dest, "w").write("".join(res))
