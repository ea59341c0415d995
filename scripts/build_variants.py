#!/usr/bin/env python
"""Build the library with different -D flags on the GPU box and time the step of one layer with each (default: the headline
shape; MDBN_AB_SHAPE="V,H,B,k,gauss" selects another, as scripts/step_ab.py).
    python scripts/build_variants.py "-DX6_SCHED=0" "-DX6_SCHED=1" ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = [os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in ("mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_thin.hip", "mdbn_gchain.hip", "mdbn_stream.hip", "mdbn_capi.hip")]
src = sorted(os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "mdbn_amd", "csrc")) if f.endswith(".hip"))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
for i, flags in enumerate(sys.argv[1:]):
    so = os.path.join(out, "libmdbn_var%d.so" % i)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"] + flags.split() + src + ["-o", so])
    prog = r'''
import sys, time; sys.path.insert(0, %r)
import numpy as np, torch, mdbn_amd
from mdbn_amd import _lib
_lib.use_diagnostic_library(%r)
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
import os
V, H, B, K, GAUSS = [int(x) for x in os.environ.get("MDBN_AB_SHAPE", "4096,1024,512,1,1").split(",")]
N = 32768
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
def run(n):
    c = None
    for it in range(n):           # (the next minibatch announced, as the trainers do)
        mb, nx = it %% (N // B), (it + 1) %% (N // B)
        c = fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0, next_indexes=perm[nx * B:(nx + 1) * B])
    return c
run(30); eng.synchronize()
ts = []
for r in range(5):
    t0 = time.perf_counter(); c = run(100); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e4)
eng.kernel_timing(True); run(50); eng.synchronize()
groups = {}
for ms, alg, pipe, kind in eng.kernel_timing_detail():
    groups.setdefault(kind, []).append(ms)
eng.kernel_timing(False)
print("%%-40s median %%.1f us/step   cost %%.6f   %%s" %% (%r, np.median(ts), float(c),
      ", ".join("kind %%d: %%.1f us" %% (k, 1e3 * np.mean(t)) for k, t in sorted(groups.items()))), flush=True)
''' % (ROOT, so, flags)
    # MDBN_AB_SHAPES="V,H,B,k,gauss;V,H,B,k,gauss;..." times several layers with one build
    for shape in os.environ.get("MDBN_AB_SHAPES", os.environ.get("MDBN_AB_SHAPE", "4096,1024,512,1,1")).split(";"):
        print(shape, end="  ", flush=True)
        subprocess.check_call([sys.executable, "-c", prog], env=dict(os.environ, MDBN_AB_SHAPE=shape))
