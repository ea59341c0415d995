#!/usr/bin/env python
"""ISA audit of the inline-asm loads in the statistics kernel's loader waves (mdbn_planes.hip, EARLYW): between an asm
`global_load_dwordx4` and the counted wait that covers it, hipcc must neither read nor write the destination registers
(it believes they are written when the statement ends).  Compiles mdbn_planes.hip to assembly and scans the kernel.
    python scripts/experiments/audit_asm_loads.py          (exit status 1 if anything is found)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "mdbn_amd", "csrc", "mdbn_planes.hip")
out = os.path.join(tempfile.gettempdir(), "mdbn_planes_audit.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                      cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
text = open(out).read()
kerns = ["_ZN4mdbn18gemm_planes_kernelILi1ELi1ELi3ELi2ELi16ELi128EEEvNS_13PlaneGemmArgsE",      # statistics + fused update
         "_ZN4mdbn18gemm_planes_kernelILi1ELi1ELi3ELi0ELi16ELi128EEEvNS_13PlaneGemmArgsE",      # statistics, data-parallel (deferred update)
         "_ZN4mdbn22gemm_planes_bal_kernelILi1ELi1ELi3ELi1EEEvNS_13PlaneGemmArgsE"]             # the same on balanced launches
body = ""
for kern in kerns:
    part = text[text.index("\n" + kern + ":"):]
    body += part[:part.index(".end_amdhsa_kernel")] + "\n;;#ASMSTART\ns_waitcnt vmcnt(0)\n;;#ASMEND\n;;#ASMSTART\ns_waitcnt vmcnt(0)\n;;#ASMEND\n"


def regs(tok):
    tok = tok.strip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


inasm, pend, bad, nloads = False, [], [], 0
for i, line in enumerate(body.split("\n")):
    t = line.strip()
    if t.startswith(";;#ASMSTART"):
        inasm = True; continue
    if t.startswith(";;#ASMEND"):
        inasm = False; continue
    if not t or t[0] in ";.":
        continue
    parts = re.split(r"[ ,]+", t)
    if inasm and parts[0] == "global_load_dwordx4":
        pend.append([i, regs(parts[1]), 0]); nloads += 1; continue
    if inasm and parts[0] == "s_waitcnt":
        # an item's loads survive the ring wait of the next stage (vmcnt(12)) and are covered by the next item wait; the
        # last item of a phase is covered by a vmcnt(12) item wait: count ring waits as 1, item waits as 2, retire at 3
        m = re.search(r"vmcnt\((\d+)\)", t)
        if m is None:                       # an lgkmcnt-only wait (LDS reads of the MFMA waves): no VMEM retires
            continue
        n = int(m.group(1))
        for p in pend:
            p[2] += 1 if n == 12 else 2
        pend = [p for p in pend if p[2] < 3]
        continue
    touched = set()
    for tok in parts[1:]:
        touched |= regs(tok)
    for p in pend:
        if touched & p[1]:
            bad.append((i + 1, t[:80], p[0] + 1))
print("%d asm loads; %d touches of a pending destination register" % (nloads, len(bad)))
for b in bad[:40]:
    print("  line %d: %s   (load at line %d)" % b)
sys.exit(1 if bad else 0)
