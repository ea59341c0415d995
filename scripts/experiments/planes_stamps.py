#!/usr/bin/env python
"""Phase timeline of the propup launch inside the c2 step, from a -DMDBN_STAMP build made on the GPU box:
where the fixed cost of a launch sits (dispatch skew over the workgroups, first LDS-DMA round trip, main loop,
store tail).  Stamps are s_memrealtime (100 MHz): 10-ns resolution.
    python scripts/experiments/planes_stamps.py"""
import os, subprocess, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libmdbn_planes_stamp.so")
src = [os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in ("mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_capi.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DMDBN_STAMP"] + src + ["-o", so])
import numpy as np, torch
from mdbn_amd import _lib
_lib.use_diagnostic_library(so)
import mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, N = 4096, 1024, 512, 32768
g = torch.Generator(device="cpu").manual_seed(0)
data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
_, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
fn = mdbn_amd.function(up, data)
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
for it in range(30):
    fn(indexes=perm[it * B:(it + 1) * B], momentum=0.0)
eng.synchronize()
stamps = torch.zeros(256 * 16, dtype=torch.int64, device=eng.device)
eng.lib.mdbn_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
rows = []
for rep in range(5):
    stamps.zero_()
    fn(indexes=perm[(40 + rep) * B:(41 + rep) * B], momentum=0.0)
    eng.synchronize()
    st = stamps.cpu().numpy().reshape(256, 16).astype(np.int64)
    t0 = min(st[:, 0].min(), st[:, 8].min())
    us = lambda x: (x - t0) / 100.0
    rows.append(dict(
        first_wg_start=0.0, last_wg_start=us(np.maximum(st[:, 0], st[:, 8]).max()),
        dma_issued=np.median(st[:, 2] - st[:, 0]) / 100.0, stage0_landed=np.median(st[:, 3] - st[:, 2]) / 100.0,
        mfma_wait_stage0=np.median(st[:, 9] - st[:, 8]) / 100.0, loop=np.median(st[:, 10] - st[:, 9]) / 100.0,
        stores_issued=np.median(st[:, 11] - st[:, 10]) / 100.0, stores_acked=np.median(st[:, 12] - st[:, 11]) / 100.0,
        loop_shader_cycles=np.median(st[:, 14] - st[:, 13]), loop_clock_ghz=np.median((st[:, 14] - st[:, 13]) / np.maximum(1, (st[:, 10] - st[:, 9]) * 10.0)),
        barrier_wait_cycles=np.median(st[:, 15]), barrier_wait_frac=np.median(st[:, 15] / np.maximum(1, st[:, 14] - st[:, 13])),
        first_loop_end=us(st[:, 10].min()), last_loop_end=us(st[:, 10].max()), last_done=us(st[:, 12].max())))
eng.lib.mdbn_debug_set_stamps(C.c_void_p(0))
keys = list(rows[0].keys())
print("propup launch of the c2 step, microseconds (median over workgroups; 5 steps):")
for k in keys:
    print("  %-18s %s" % (k, "  ".join("%9.2f" % r[k] for r in rows)))
