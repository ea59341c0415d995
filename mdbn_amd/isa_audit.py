"""ISA audit of inline-asm loads (gfx950 assembly as ``hipcc -S`` prints it).

Some loader waves hide memory-bound side work behind ``asm volatile("global_load_dwordx4 ...")`` statements that hipcc
does not count in ``vmcnt`` (csrc/mdbn_planes.hip, EARLYW): the destination registers are only valid after a hand-counted
``s_waitcnt vmcnt(N)``.  hipcc believes they are written when the asm statement ends, so nothing stops it from copying,
reading or reusing them earlier -- silent corruption, or a memory fault when a clobbered register is an address (both
happened during development).  This module proves, on the assembly of the build itself, that it did not:

* every kernel whose body contains an asm ``global_load`` is audited -- discovered from the text, no list of names;
* the retirement rule is derived from the ``vmcnt`` operands: vector-memory operations retire in issue order, so
  ``s_waitcnt vmcnt(N)`` retires an asm load once at least N vector-memory operations were issued after it.  A forward
  dataflow over the kernel's control-flow graph carries, per load that may still be pending on SOME path, the smallest
  such count over those paths (an operation inside a branch that another path skips does not help that path);
* between an asm load and the point where it is retired on every path no instruction may name one of its destination
  registers;
* a loop around pending asm loads, and an asm load that may still be pending at ``s_endpgm``, are reported too.

``audit_assembly(text)`` -> report dict; ``audit_source(path)`` compiles with the build's own flags first."""
import os
import re
import subprocess
import tempfile

VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)(load|store|atomic)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
ASM_LOAD_IN_SOURCE = re.compile(r'asm\s+volatile\s*\(\s*"global_load')


def _regs(tok):
    tok = tok.strip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def kernels(text):
    """(name, body lines) of every kernel in an assembly file."""
    out = []
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)", text, flags=re.M):
        name = m.group(1)
        start = text.find("\n" + name + ":")
        if start < 0:
            continue
        end = text.find(".Lfunc_end", start)
        out.append((name, text[start:end if end > 0 else m.start()].split("\n")))
    return out


def audit_kernel(lines):
    """Forward dataflow over the kernel's control-flow graph.  State at a point: the asm loads that may still be pending
    there (on ANY path) with the SMALLEST number of vector-memory operations issued after each (over the paths on which it
    is pending).  Returns (number of asm loads, list of findings)."""
    ins = []            # (mnemonic, operand tokens, inside an asm statement, raw text)
    labels = {}
    inasm = False
    for raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        m = LABEL.match(t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t[0] in ";.":
            continue
        t = t.split(";")[0].strip()
        parts = re.split(r"[ ,]+", t)
        ins.append((parts[0], parts[1:], inasm, t))
    n = len(ins)
    # basic blocks: leaders = entry, branch targets, instructions after a branch / s_endpgm
    leaders = {0} | set(labels.values())
    for i, (op, args, _, _) in enumerate(ins):
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            leaders.add(i + 1)
    starts = sorted(x for x in leaders if x < n)
    block_of = {}
    for b, lo in enumerate(starts):
        hi = starts[b + 1] if b + 1 < len(starts) else n
        for i in range(lo, hi):
            block_of[i] = b
    preds = [[] for _ in starts]
    for b, lo in enumerate(starts):
        hi = starts[b + 1] if b + 1 < len(starts) else n
        op, args, _, _ = ins[hi - 1]
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = labels.get(args[0]) if args else None
            if tgt is not None and tgt < n:
                preds[block_of[tgt]].append(b)
        if op not in ("s_branch", "s_endpgm", "s_setpc_b64") and hi < n:
            preds[block_of[hi]].append(b)
    info = {}           # load id (its instruction index) -> (destination registers, text)
    for i, (op, args, inasm_, raw) in enumerate(ins):
        if inasm_ and op.startswith("global_load") and not op.startswith("global_load_lds"):
            info[i] = (_regs(args[0]), raw)
    nloads = len(info)

    def transfer(b, state, findings=None):
        lo = starts[b]
        hi = starts[b + 1] if b + 1 < len(starts) else n
        state = dict(state)
        for i in range(lo, hi):
            op, args, inasm_, raw = ins[i]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", raw)
                if m is not None:
                    k = int(m.group(1))
                    state = {lid: cnt for lid, cnt in state.items() if cnt < k}
                continue
            if findings is not None and state:
                touched = set()
                for tok in args:
                    touched |= _regs(tok)
                for lid in state:
                    if touched & info[lid][0]:
                        findings.append("instruction %d `%s` names a destination register of the asm load `%s` (instruction "
                                        "%d) before a wait that retires it on every path" % (i, raw[:70], info[lid][1][:50], lid))
            if i in info or VMEM.match(op) or op.startswith("global_load_lds"):
                for lid in state:
                    state[lid] = min(state[lid] + 1, 1 << 20)
            if i in info:
                state[i] = 0
            if op == "s_endpgm" and state and findings is not None:
                first = min(state)
                findings.append("%d asm load(s) may still be pending at s_endpgm (first: instruction %d `%s`): no wait retires "
                                "them on every path" % (len(state), first, info[first][1][:50]))
        return state

    def merged(b, out_state):
        state = {}
        for p in preds[b]:
            if out_state[p] is None:
                continue
            for lid, cnt in out_state[p].items():
                state[lid] = min(cnt, state[lid]) if lid in state else cnt
        return state

    # fixpoint over ALL edges (hipcc places cold blocks out of line: a forward jump and a jump back are not a loop, and a real
    # loop is handled the same way: pending sets only grow, counts only shrink)
    out_state = [None] * len(starts)
    for _ in range(64):
        changed = False
        for b in range(len(starts)):
            new = transfer(b, merged(b, out_state))
            if new != out_state[b]:
                out_state[b], changed = new, True
        if not changed:
            break
    else:
        return nloads, ["the dataflow did not converge in 64 sweeps: not audited"]
    findings = []
    for b in range(len(starts)):
        transfer(b, merged(b, out_state), findings)
    return nloads, findings


def audit_assembly(text):
    """Audit every kernel that contains an asm global_load.  Returns {'kernels': {name: n_loads}, 'loads': N, 'findings': [...]}."""
    report = {"kernels": {}, "loads": 0, "findings": []}
    for name, lines in kernels(text):
        body = "\n".join(lines)
        if not re.search(r";;#ASMSTART\s*\n\s*global_load_dword", body):
            continue
        n, found = audit_kernel(lines)
        report["kernels"][name] = n
        report["loads"] += n
        report["findings"] += ["%s: %s" % (name, f) for f in found]
    return report


def needs_audit(source_path):
    with open(source_path) as f:
        return ASM_LOAD_IN_SOURCE.search(f.read()) is not None


def hipcc_version(hipcc):
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
        return " | ".join(line.strip() for line in out.splitlines()[:2])
    except Exception as exc:
        return "unknown (%r)" % (exc,)


def audit_source(source_path, flags, hipcc="/opt/rocm/bin/hipcc", defines=()):
    """Compile ``source_path`` to gfx950 assembly with the build's flags and audit it."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "audit.s")
        cmd = [hipcc] + [f for f in flags if f not in ("-shared", "-fPIC")] + list(defines) + \
              ["-S", "--cuda-device-only", "-o", out, os.path.basename(source_path)]
        subprocess.run(cmd, cwd=os.path.dirname(source_path), check=True, stderr=subprocess.DEVNULL)
        with open(out) as f:
            report = audit_assembly(f.read())
    report["source"] = os.path.basename(source_path)
    report["hipcc"] = hipcc_version(hipcc)
    return report


# ----------------------------------------------------------------------------------------------------------------------
# Wait report: where a kernel sits on a memory round trip it need not sit on.  Not a correctness audit -- a reading aid
# that found, in round 4, the serialised slab loads of the activation epilogues (a convert-and-add behind every predicated
# load), the four index lookups in a row at the head of every epilogue with a cost target, the gather-ahead's index lookups
# that drained the LDS-DMA ring mid-loop, and kernel arguments fetched in three or four batches at a kernel's head.
# ----------------------------------------------------------------------------------------------------------------------
def wait_report(text, window=3):
    """Per kernel of an assembly file: ``loads`` (vector-memory loads, LDS-DMA excluded), ``waited_at_once`` (loads with an
    ``s_waitcnt vmcnt(0)`` within ``window`` instructions behind them: each is a full round trip nothing overlaps) and
    ``kernarg_batches`` (groups of ``s_load`` from the kernarg pointer ``s[0:1]`` separated by a ``lgkmcnt`` wait: every
    batch after the first is a scalar round trip the kernel's head could have shared).  Returns {kernel: dict}."""
    out = {}
    for name, lines in kernels(text):
        code = [t.strip().split(";")[0].strip() for t in lines]
        code = [t for t in code if t and t[0] not in ".;" and not t.endswith(":")]
        loads = at_once = 0
        batches, open_batch = 0, False
        for i, t in enumerate(code):
            op = t.split()[0]
            if (op.startswith("global_load") and not op.startswith("global_load_lds")) or op.startswith("buffer_load") or \
                    op.startswith("flat_load"):
                loads += 1
                if any(c.startswith("s_waitcnt") and "vmcnt(0)" in c for c in code[i + 1:i + 1 + window]):
                    at_once += 1
            if op.startswith("s_load") and "s[0:1]" in t:
                if not open_batch:
                    batches, open_batch = batches + 1, True
            elif op == "s_waitcnt" and "lgkmcnt" in t:
                open_batch = False
        out[name] = {"loads": loads, "waited_at_once": at_once, "kernarg_batches": batches}
    return out

