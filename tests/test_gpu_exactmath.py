"""The device evaluates the frozen IEEE `/` and `sqrt` of its hot spots with short sequences (hardware estimate + FMA corrections,
csrc/rt_math.hpp).  tools/exactmath/verify (built by __graft_entry__.build()) compares them on the GPU with the compiler's
correctly rounded operations: every float32 input for 1/x and sqrt, 4e10 operand pairs for the quotient, 2e10 vectors for
normalize — zero mismatches allowed."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_division_and_sqrt_sequences_are_correctly_rounded():
    exe = os.path.join(ROOT, "tools", "exactmath", "verify")
    if not os.path.exists(exe):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    counts = re.findall(r"^(\w+)\s+mismatches[^:]*:\s*(\d+)", out.stdout, re.M)
    assert [name for name, _ in counts] == ["rcp_", "sqrt_", "div_mid", "normalize", "log_unit"], out.stdout
    assert all(int(n) == 0 for _, n in counts), out.stdout
