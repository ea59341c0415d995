"""Build libmdbn_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Staleness is decided by CONTENT, not mtime: the SHA-256 of the sources is compiled into the library
(``mdbn_source_hash``) and compared with the sources on disk -- a prebuilt .so that travelled to
another box with rewritten mtimes is still recognised as current or stale."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmdbn_hip.so")
SOURCES = ["mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_thin.hip", "mdbn_gchain.hip", "mdbn_stream.hip", "mdbn_capi.hip"]
HEADERS = ["mdbn_kernels.h", "mdbn_device.h", "philox.h", "row_pool.h", "mdbn_small.h", "mdbn_thin.h", "mdbn_gchain.h", "mdbn_bf16x3.h", os.path.join("..", "..", "include", "mdbn_hip.h")]
HASH_TAG = b"MDBN_SOURCE_HASH_TAG="
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"]


def source_hash():
    """SHA-256 over the HIP sources, headers and compile flags (hex, 64 characters)."""
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p):
            h.update(f.encode())
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def audit_verdicts():
    """The ISA-audit verdicts of the objects the current library was linked from ({source: report})."""
    import glob
    import json
    out = {}
    for path in glob.glob(os.path.join(CSRC, ".obj", "*.audit.json")):
        with open(path) as fh:
            r = json.load(fh)
        out[r.get("source", os.path.basename(path))] = r
    return out


def built_hash(path=LIB):
    """The source hash compiled into an existing library, or None."""
    # Read from the file's bytes, never through dlopen: glibc keys loaded objects by name, so a handle opened
    # here would make a later CDLL of a library rebuilt at the same path return the OLD mapping.
    if not os.path.exists(path):
        return None
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    at = blob.find(HASH_TAG)
    if at < 0:
        return None
    end = blob.find(b"\0", at)
    return blob[at + len(HASH_TAG):end].decode("ascii", "replace")


def is_stale():
    return built_hash() != source_hash()


def build_lib(force=False, verbose=False):
    """Compile the HIP kernels + C-ABI into mdbn_amd/libmdbn_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # one object per source, compiled in parallel and cached by content (source + headers + flags):
    # editing one file does not recompile the others
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    hdr = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for f in HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            hdr.update(fh.read())
    objs, jobs = [], []
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    for src in srcs:
        h = hdr.copy()
        with open(os.path.join(CSRC, src), "rb") as fh:
            h.update(fh.read())
        define = []
        if src == "mdbn_capi.hip":          # the library-wide hash lives in this object
            define = ['-DMDBN_SRC_HASH="%s"' % source_hash()]
            h.update(define[0].encode())
        obj = os.path.join(objdir, "%s.%s.o" % (os.path.splitext(src)[0], h.hexdigest()[:16]))
        objs.append(obj)
        if force or not os.path.exists(obj):
            cmd = [hipcc] + cflags + define + ["-c", "-o", obj + ".tmp%d" % os.getpid(), src]
            if verbose:
                print(" ".join(cmd))
            jobs.append((subprocess.Popen(cmd, cwd=CSRC), obj))
    # ISA audit of the sources that hide loads from hipcc (inline-asm global_load + hand-counted waits, isa_audit.py): the
    # assembly hipcc makes of them HERE, with these flags, is proven free of premature uses of a load's destination; a
    # finding fails the build.  The verdict is cached beside the object (same content hash) with the compiler's version.
    import json
    from . import isa_audit
    audits = []
    for src, obj in zip(srcs, objs):
        path = os.path.join(CSRC, src)
        verdict = obj[:-2] + ".audit.json"
        if isa_audit.needs_audit(path) and (force or not os.path.exists(verdict)):
            asm = obj[:-2] + ".audit.tmp%d.s" % os.getpid()
            cmd = [hipcc] + [f for f in cflags if f != "-fPIC"] + ["-S", "--cuda-device-only", "-o", asm, src]
            audits.append((subprocess.Popen(cmd, cwd=CSRC, stderr=subprocess.DEVNULL), src, asm, verdict))
    for proc, obj in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, "hipcc -c (%s)" % os.path.basename(obj))
        os.replace(obj + ".tmp%d" % os.getpid(), obj)
    for proc, src, asm, verdict in audits:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, "hipcc -S for the ISA audit (%s)" % src)
        with open(asm) as fh:
            report = isa_audit.audit_assembly(fh.read())
        os.remove(asm)
        report["source"], report["hipcc"] = src, isa_audit.hipcc_version(hipcc)
        if report["findings"] or not report["loads"]:
            raise RuntimeError("ISA audit of %s failed with %s (%d asm loads in %d kernels):\n  %s" % (
                src, report["hipcc"], report["loads"], len(report["kernels"]),
                "\n  ".join(report["findings"][:20]) or "no asm load found although the source has some"))
        with open(verdict, "w") as fh:
            json.dump(report, fh, indent=1)
        if verbose:
            print("ISA audit of %s: %d asm loads in %d kernels, no finding (%s)" % (src, report["loads"], len(report["kernels"]), report["hipcc"]))
    keep = set(objs) | set(o[:-2] + ".audit.json" for o in objs)
    for old in os.listdir(objdir):          # drop objects (and audit verdicts) of superseded sources
        if os.path.join(objdir, old) not in keep and (old.endswith(".o") or old.endswith(".audit.json")):
            os.remove(os.path.join(objdir, old))
    tmp = LIB + ".tmp%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    os.replace(tmp, LIB)            # atomic: concurrent ranks never dlopen a half-written file
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
