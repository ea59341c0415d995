#!/usr/bin/env python
"""(Needs the diagnostic variants: `git apply scripts/experiments/diagnostic_variants.patch` first -- they no longer live in
the product kernels.)
Diagnostic: bf16x6 GEMM timing with producer ablations (-DX6_ABLATE=1 no split arithmetic, =2 no producer
LDS stores; results are wrong in those builds).  Builds three libraries on the GPU box."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
src = [os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in ("mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_capi.hip")]
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
import mdbn_amd
from mdbn_amd import _lib
for abl in (0, 1, 2):
    so = os.path.join(out, "libmdbn_abl%d.so" % abl)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                           "-DX6_ABLATE=%d" % abl] + src + ["-o", so])
    code = r'''
import sys; sys.path.insert(0, %r)
import torch, mdbn_amd
from mdbn_amd import _lib
''' % (ROOT,)
    # run in a subprocess so that each library is loaded fresh
    prog = code + r'''
_lib.use_diagnostic_library(%r)          # diagnostic library instead of the product build
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
eng.set_option("fused_epilogue", 0)
V, H = 4096, 1024
W = eng.alloc_matrix(V, H); W.normal_(0, 0.05)
hb = eng.alloc_vector(H); vb = eng.alloc_vector(V)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    eng.synchronize(); eng.kernel_timing(True)
    for _ in range(reps): fn()
    eng.synchronize(); n, ms = eng.kernel_timing_read(); eng.kernel_timing(False)
    return ms * 1e3 / n
for B in (512, 4096):
    x = eng.alloc_matrix(B, V); x.normal_()
    up = timeit(lambda: eng.propup(x, W, hb, want_pre=False, want_sample=False))
    print("ablate=%d B=%%d up %%.1f us %%.1f TF" %% (B, up, 2.0 * B * V * H / up / 1e6), flush=True)
''' % (so, abl)
    subprocess.check_call([sys.executable, "-c", prog])
