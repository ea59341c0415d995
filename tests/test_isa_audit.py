"""The statistics kernels' loader waves hide their side work behind inline-asm loads that hipcc does not count
(csrc/mdbn_planes.hip, EARLYW): between such a load and the counted wait that covers it, no instruction may read or
write its destination registers -- hipcc believes they are written when the asm statement ends, and a copy or a reuse in
between is silent corruption (the first version of the gather-ahead faulted exactly so).  This test compiles the file to
assembly and runs the audit; it needs hipcc, not a GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_a_pending_asm_load_destination():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "experiments", "audit_asm_loads.py")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " asm loads; 0 touches" in out.stdout and not out.stdout.startswith("0 asm loads"), out.stdout
