#!/usr/bin/env python
"""Where the kernels wait: per kernel of the library, vector loads that are followed at once by `s_waitcnt vmcnt(0)` and the
number of batches its kernel arguments are fetched in (mdbn_amd/isa_audit.py, wait_report).  A reading aid, not a gate:
producer loops wait by design; an epilogue or a kernel's head should not.
    python scripts/experiments/isa_wait_report.py [min_waited_at_once=6]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mdbn_amd import build, isa_audit
floor = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for src in build.SOURCES:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.run([build.HIPCC if hasattr(build, "HIPCC") else "/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", out, src],
                       cwd=build.CSRC, check=True, stderr=subprocess.DEVNULL)
        rep = isa_audit.wait_report(open(out).read())
    print("== %s: %d kernels" % (src, len(rep)))
    for name, r in sorted(rep.items(), key=lambda kv: -kv[1]["waited_at_once"]):
        if r["waited_at_once"] >= floor or r["kernarg_batches"] >= 4:
            try:
                nice = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
            except Exception:
                nice = name
            print("  %3d of %4d loads waited for at once, kernel arguments in %d batches   %s" % (r["waited_at_once"], r["loads"], r["kernarg_batches"], nice[:110]))
