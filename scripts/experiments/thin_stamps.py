#!/usr/bin/env python
"""Phase timeline (wall clock, 100 MHz) of every workgroup of thin_pass_kernel<1> (the Gibbs half-step pair of the
thin-batch path) inside a real training step: builds the library with -DMDBN_STAMP on the GPU box.
    MDBN_AB_SHAPE=19937,400,20,1,1 python scripts/experiments/thin_stamps.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "mdbn_amd", "csrc")
src = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libmdbn_thin_stamp.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DMDBN_STAMP"] + src + ["-o", so])
import numpy as np, torch
from mdbn_amd import _lib
_lib.use_diagnostic_library(so)
import mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, K, GAUSS = [int(x) for x in os.environ.get("MDBN_AB_SHAPE", "19937,400,20,1,1").split(",")]
N = 4096
g = torch.Generator(device="cpu").manual_seed(0)
if GAUSS:
    data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.001, k=K, lambda_2=0.1, batch_size=B)
else:
    data = mdbn_amd.shared((torch.rand((N, V), generator=g) < 0.3).float().to(eng.device))
    rbm = mdbn_amd.RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.05, k=K, weightcost=2e-4, batch_size=B)
fn = mdbn_amd.function(up, data)
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
def run(n, it0=0):
    for it in range(it0, it0 + n):
        mb, nb = it % (N // B), (it + 1) % (N // B)
        fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0, next_indexes=perm[nb * B:(nb + 1) * B])
run(20); eng.synchronize()
stamps = torch.zeros(256 * 16, dtype=torch.int64, device=eng.device)
names = ["stage W (loads + LDS stores)", "chain image", "barrier", "phase 1 (down)", "barrier", "red + barrier", "visible epilogue", "barrier",
         "phase 2 (up)", "part stores", "cost sum"]
acc = []
for rep in range(5):
    stamps.zero_()
    eng.lib.mdbn_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    run(1, 20 + rep); eng.synchronize()
    eng.lib.mdbn_debug_set_stamps(C.c_void_p(0))
    st = stamps.cpu().numpy().reshape(256, 16).astype(np.int64)
    live = st[:, 0] > 0
    st = st[live]
    acc.append(st)
st = acc[-1]
t0 = st[:, 0].min()
print("%d workgroups stamped; first start -> last end: %.2f us; starts spread over %.2f us" %
      (len(st), (st[:, 10].max() - t0) / 100.0, (st[:, 0].max() - t0) / 100.0))
d = np.diff(st[:, :11], axis=1) / 100.0
for i, n in enumerate(names[:10]):
    print("  %-30s mean %6.2f us  (min %5.2f  max %5.2f)" % (n, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
print("  workgroup lifetime mean %.2f us (min %.2f max %.2f)" % ((st[:, 10] - st[:, 0]).mean() / 100.0, (st[:, 10] - st[:, 0]).min() / 100.0,
                                                                 (st[:, 10] - st[:, 0]).max() / 100.0))
for w in (0, len(st) // 2, len(st) - 1):
    print("  wg %3d: " % w + " ".join("%6.2f" % ((x - t0) / 100.0) for x in st[w, :11]))
