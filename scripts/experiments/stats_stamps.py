#!/usr/bin/env python
"""Phase timeline of the propdown launch on narrow tiles (128 x 64, activation inside) in the c2 step, from a
-DMDBN_STAMP -DMDBN_STAMP_STATS build made on the GPU box.  Stamps are s_memrealtime (100 MHz): 10-ns resolution.
    python scripts/experiments/narrow_stamps.py"""
import os, subprocess, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libmdbn_stats_stamp.so")
src = [os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in ("mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_capi.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DMDBN_STAMP",
                       "-DMDBN_STAMP_STATS"] + src + ["-o", so])
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
    fn(indexes=perm[it * B:(it + 1) * B], momentum=0.0, next_indexes=perm[(it + 1) * B:(it + 2) * B])
eng.synchronize()
stamps = torch.zeros(256 * 16, dtype=torch.int64, device=eng.device)
eng.lib.mdbn_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
rows = []
for rep in range(5):
    stamps.zero_()
    fn(indexes=perm[(30 + rep) * B:(31 + rep) * B], momentum=0.0, next_indexes=perm[(31 + rep) * B:(32 + rep) * B])
    eng.synchronize()
    st = stamps.cpu().numpy().reshape(256, 16).astype(np.int64)
    t0 = min(st[:, 0].min(), st[:, 8].min())
    us = lambda x: (x - t0) / 100.0
    med = lambda a, b: float(np.median(st[:, a] - st[:, b])) / 100.0
    rows.append(dict(
        last_wg_start=us(np.maximum(st[:, 0], st[:, 8]).max()),
        loader_setup=med(1, 0), stage0_issue=med(2, 1), stage0_landed=med(3, 2), loader_loop_with_side_work=med(4, 3),
        mfma_finalize_units_done=med(8, 0), mfma_wait_stage0=med(9, 8), mfma_loop=med(10, 9), park_and_sync=med(11, 10),
        speed_epilogue=med(12, 11),
        loop_clock_ghz=float(np.median((st[:, 14] - st[:, 13]) / np.maximum(1, (st[:, 10] - st[:, 9]) * 10.0))),
        barrier_wait_frac=float(np.median(st[:, 15] / np.maximum(1, st[:, 14] - st[:, 13]))),
        first_done=us(st[:, 12].min()), last_done=us(st[:, 12].max())))
eng.lib.mdbn_debug_set_stamps(C.c_void_p(0))
print("statistics launch of the c2 step (with the next-minibatch hint), microseconds (median over workgroups; 5 steps):")
for k in rows[0]:
    print("  %-22s %s" % (k, "  ".join("%9.2f" % r[k] for r in rows)))
