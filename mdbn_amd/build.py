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
SOURCES = ["mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_capi.hip"]
HEADERS = ["mdbn_kernels.h", "mdbn_device.h", "philox.h", "row_pool.h", "mdbn_small.h", os.path.join("..", "..", "include", "mdbn_hip.h")]
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
    for proc, obj in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, "hipcc -c (%s)" % os.path.basename(obj))
        os.replace(obj + ".tmp%d" % os.getpid(), obj)
    for old in os.listdir(objdir):          # drop objects of superseded sources
        if os.path.join(objdir, old) not in objs and old.endswith(".o"):
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
