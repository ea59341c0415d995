#!/usr/bin/env python
"""Where the one-launch step's time goes: wall-clock stamps (100 MHz) of workgroup 0 at the phase boundaries of
small_cd_kernel (mdbn_debug_set_stamps), median over repeated steps.
    MDBN_AB_SHAPE=512,40,512,5,1 python scripts/experiments/small_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libmdbn_small_stamp.so")
src = [os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in ("mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_capi.hip")]
if not os.path.exists(so) or os.environ.get("MDBN_STAMP_REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DMDBN_STAMP"] + (["-DMDBN_STAMP_CLK"] if os.environ.get("MDBN_STAMP_CLK") else []) + src + ["-o", so, "-ldl"])
import numpy as np, torch
from mdbn_amd import _lib
_lib.use_diagnostic_library(so)
import mdbn_amd
from mdbn_amd import RngAddr
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, K, GAUSS = [int(x) for x in os.environ.get("MDBN_AB_SHAPE", "512,40,512,5,1").split(",")]
rs = np.random.RandomState(0)
W = (rs.uniform(-1, 1, size=(V, H)) * 4 * np.sqrt(6.0 / (V + H))).astype(np.float32)
x = rs.normal(size=(B, V)).astype(np.float32) if GAUSS else (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, np.zeros(H, np.float32), np.zeros(V, np.float32), x)]
stamps = torch.zeros(64, dtype=torch.int64, device=eng.device)
eng.lib.mdbn_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
rows = []
for it in range(30):
    eng.cd_step(dx, None, dW, dhb, dvb, bool(GAUSS), K, RngAddr(1, 0, it, 0, 0))
    eng.synchronize()
    s = stamps.cpu().numpy()
    n = int(s[0])
    rows.append(np.diff(s[1:1 + n]) / 100.0)
    mhz = s[63] / ((s[n] - s[1]) / 100.0)
eng.lib.mdbn_debug_set_stamps(C.c_void_p(0))
d = np.median(np.array(rows[5:]), axis=0)
names = None
names = ["stage W", "gather"]
for t in range(K + 1):
    if t > 0: names.append("down t=%d" % t)
    names.append("up t=%d" % t)
    if t == K: names.append("stats")
names.append("write partials")
print("shape %s: total %.1f us in workgroup 0; shader clock over the kernel: %.0f MHz" % (os.environ.get("MDBN_AB_SHAPE"), d.sum(), mhz))
print("  wave 0 cycles per step: up k-loop %d, sync %d, up epilogue %d, sync %d | down k-loop %d (cumulative), +epilogue %d, sync %d"
      % tuple(int(v) for v in s[48:55]))
for nm, v in zip(names, d):
    print("  %-16s %6.2f us" % (nm, v))
