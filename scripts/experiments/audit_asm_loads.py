#!/usr/bin/env python
"""ISA audit of the inline-asm loads (mdbn_amd/isa_audit.py) as a command: compiles every kernel source that hides loads
from hipcc to gfx950 assembly with the build's flags and proves, by a dataflow over each kernel's control-flow graph, that
no instruction names a load's destination register before a wait retires the load on every path.  The build runs the same
audit (mdbn_amd/build.py); this prints the report.
    python scripts/experiments/audit_asm_loads.py          (exit status 1 if anything is found)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mdbn_amd import build, isa_audit
bad = 0
for src in build.SOURCES:
    path = os.path.join(build.CSRC, src)
    if not isa_audit.needs_audit(path):
        continue
    r = isa_audit.audit_source(path, build.HIPCC_FLAGS)
    print("%s: %d asm loads in %d kernels; %d findings   [%s]" % (src, r["loads"], len(r["kernels"]), len(r["findings"]), r["hipcc"]))
    for k, n in sorted(r["kernels"].items()):
        print("   %4d  %s" % (n, k))
    for f in r["findings"][:40]:
        print("  !!", f)
    bad += len(r["findings"]) + (0 if r["loads"] else 1)
sys.exit(1 if bad else 0)
