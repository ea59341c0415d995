#!/usr/bin/env python
"""Phase timeline (wall clock, 100 MHz) of every workgroup of one streaming bf16x6 GEMM launch (csrc/mdbn_stream.hip) inside
a real training step: builds the library with -DMDBN_STAMP -DSTREAM_STAMP_SEL=<0 propup | 1 propdown | 2 statistics>.
    MDBN_AB_SHAPE=2048,400,512,1,1 python scripts/experiments/stream_stamps.py 0"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sel = int(sys.argv[1]) if len(sys.argv) > 1 else 0
csrc = os.path.join(ROOT, "mdbn_amd", "csrc")
src = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libmdbn_stream_stamp%d.so" % sel)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DMDBN_STAMP",
                       "-DSTREAM_STAMP_SEL=%d" % sel] + src + ["-o", so])
import numpy as np, torch
from mdbn_amd import _lib
_lib.use_diagnostic_library(so)
import mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, K, GAUSS = [int(x) for x in os.environ.get("MDBN_AB_SHAPE", "2048,400,512,1,1").split(",")]
N = 8192
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
        mb = it % (N // B)
        fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0)
run(20); eng.synchronize()
stamps = torch.zeros(2048 * 16, dtype=torch.int64, device=eng.device)
names = ["setup (+ finalize units)", "main loop", "drain", "park + reduce + epilogue (strip 0)", "strip 1"]
for rep in range(3):
    stamps.zero_()
    eng.lib.mdbn_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    run(1, 20 + rep); eng.synchronize()
    eng.lib.mdbn_debug_set_stamps(C.c_void_p(0))
st = stamps.cpu().numpy().reshape(2048, 16).astype(np.int64)
st = st[st[:, 0] > 0]
last = 5 if (st[:, 5] > 0).any() else 4
t0 = st[:, 0].min()
print("%s sel=%d: %d workgroups stamped; first start -> last end: %.2f us; starts spread over %.2f us" %
      (os.environ.get("MDBN_AB_SHAPE", ""), sel, len(st), (st[:, last].max() - t0) / 100.0, (st[:, 0].max() - t0) / 100.0))
d = np.diff(st[:, :last + 1], axis=1) / 100.0
for i in range(last):
    col = d[:, i][st[:, i + 1] > 0]
    print("  %-38s mean %6.2f us  (min %5.2f  max %5.2f)" % (names[i], col.mean(), col.min(), col.max()))
life = (st[:, 4] - st[:, 0]) / 100.0
print("  workgroup lifetime (to strip 0 done) mean %.2f us (min %.2f max %.2f)" % (life.mean(), life.min(), life.max()))
if (st[:, 12] > 0).any():
    cols = [8, 9, 10, 11, 12] if (st[:, 11] > 0).any() else [8, 9, 10, 12]
    names2 = ["request + park", "barrier", "reduce (LDS reads)", "epilogue arithmetic + stores"] if len(cols) == 5 else \
             ["request + park", "barrier", "reduce + epilogue arithmetic + stores"]
    e = np.diff(st[:, cols], axis=1) / 100.0
    print("    loop end -> epilogue entry              mean %5.2f us" % ((st[:, 8] - st[:, 3]).mean() / 100.0))
    for i, nm in enumerate(names2):
        print("    %-39s mean %5.2f us (min %5.2f max %5.2f)" % (nm, e[:, i].mean(), e[:, i].min(), e[:, i].max()))
    print("    after the stores (cost sum, end stamp)  mean %5.2f us" % ((st[:, 4] - st[:, 12]).mean() / 100.0))
for w in (0, len(st) // 2, len(st) - 1):
    print("  wg %4d: " % w + " ".join("%6.2f" % ((x - t0) / 100.0) for x in st[w, :last + 1]) + "  | epilogue slots 8.. " +
          " ".join("%6.2f" % ((x - t0) / 100.0) if x > t0 else "     -" for x in st[w, 8:15]))
