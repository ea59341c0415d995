#!/usr/bin/env python
"""Phase timeline (s_memtime) of one bf16x6 GEMM block: builds the library with -DMDBN_STAMP on the GPU box."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = [os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in ("mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_capi.hip")]
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libmdbn_stamp.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DMDBN_STAMP"] + src + ["-o", so])
import numpy as np, torch
from mdbn_amd import _lib
_lib.use_diagnostic_library(so)
import mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
eng.set_option("fused_epilogue", 0)
lib = eng.lib
V, H = 4096, 1024
W = eng.alloc_matrix(V, H); W.normal_(0, 0.05)
hb = eng.alloc_vector(H)
for B in (512, 4096):
    x = eng.alloc_matrix(B, V); x.normal_()
    stamps = torch.zeros(12 * 64 * 8, dtype=torch.int64, device=eng.device)      # up to 12 waves per block
    for _ in range(3): eng.propup(x, W, hb, want_pre=False, want_sample=False)
    lib.mdbn_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    eng.propup(x, W, hb, want_pre=False, want_sample=False)
    eng.synchronize()
    lib.mdbn_debug_set_stamps(C.c_void_p(0))
    st = stamps.cpu().numpy().reshape(12, 64, 8).astype(np.int64)
    nt = int((st[0, :, 0] > 0).sum())
    print("B=%d: %d slices stamped" % (B, nt))
    for w in range(12):
        s_ = st[w, :nt]
        if not (s_[:, 0] > 0).any():
            continue
        tot = s_[1:, 0] - s_[:-1, 0]
        if w < 4:
            d = np.stack([s_[:, 1] - s_[:, 0], s_[:, 2] - s_[:, 1], s_[:, 3] - s_[:, 2]], 1)
            names = "frags+MMA(F0) %5.0f  barrier %5.0f  frags+MMA(F1) %5.0f"
        else:
            d = np.stack([s_[:, 4] - s_[:, 0], s_[:, 1] - s_[:, 4], s_[:, 2] - s_[:, 1], s_[:, 3] - s_[:, 2]], 1)
            names = "wait_loads %5.0f  split+store %5.0f  issue_loads %5.0f  barrier %5.0f"
        print(" wave %d | " % w + names % tuple(d[2:-2].mean(0)) + " | slice total %.0f (min %.0f max %.0f)" % (tot[2:-2].mean(), tot.min(), tot.max()))
