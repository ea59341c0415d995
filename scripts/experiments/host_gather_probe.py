#!/usr/bin/env python
"""PCIe gather of a minibatch from a pinned host table (mdbn_gather_rows_host): time alone, by workgroup count.
    python scripts/experiments/host_gather_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, B, N = 4096, 512, 32768
x = np.random.RandomState(0).normal(size=(N, V)).astype(np.float32)
t = mdbn_amd.shared(x, resident="host")
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
out = eng.alloc_matrix(B, V)
for wg in (8, 16, 32, 64):
    eng.host_gather_workgroups = wg
    for it in range(5):
        t.rows(perm[it * B:(it + 1) * B], out=out)
    eng.synchronize()
    t0 = time.perf_counter()
    for it in range(50):
        mb = it % (N // B)
        t.rows(perm[mb * B:(mb + 1) * B], out=out)
    eng.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    print("workgroups %2d: %.1f us per 8.4 MB minibatch = %.1f GB/s" % (wg, us, 4.0 * B * V / us / 1e3), flush=True)
assert np.array_equal(out.cpu().numpy(), x[perm[(49 % (N // B)) * B:(49 % (N // B) + 1) * B].cpu().numpy()])
# the wide device gather kernel on the same pinned source, for comparison
import ctypes as C
idx = perm[:B]
eng.synchronize(); t0 = time.perf_counter()
for it in range(20):
    mdbn_amd._lib.check(eng.lib.mdbn_gather_rows(eng.ctx, eng._stream(), C.c_void_p(t.host.data_ptr()), N, V, V, eng._p(idx), 1, B,
                                               eng._p(out), out.stride(0)), "gather")
eng.synchronize()
us = (time.perf_counter() - t0) / 20 * 1e6
print("wide kernel (2048 x 256 threads): %.1f us = %.1f GB/s" % (us, 4.0 * B * V / us / 1e3))

# --- contention: the c2 step with FREE-RUNNING PCIe gathers on a side stream (no dependency between them)
H, Bb = 1024, 512
g = torch.Generator(device="cpu").manual_seed(0)
data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
_, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=Bb)
fn = mdbn_amd.function(up, data)
side = torch.cuda.Stream(eng.device)
def steps(n):
    for it in range(n):
        mb = it % (N // Bb)
        fn(indexes=perm[mb * Bb:(mb + 1) * Bb], momentum=0.0)
steps(30); eng.synchronize()
t0 = time.perf_counter(); steps(200); eng.synchronize()
alone = (time.perf_counter() - t0) / 200 * 1e6
for wg, thr in ((32, 256), (128, 64), (256, 64), (512, 64), (64, 128)):
    eng.host_gather_workgroups, eng.host_gather_threads = wg, thr
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side)
        for it in range(200):
            mb = it % (N // B)
            t.rows(perm[mb * B:(mb + 1) * B], out=out)
        e1.record(side)
    t0 = time.perf_counter(); steps(200); eng.synchronize()
    both = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    print("step alone %.1f us; beside free-running PCIe gathers of %d workgroups x %d threads: step %.1f us, gather %.1f us each"
          % (alone, wg, thr, both, e0.elapsed_time(e1) * 1e3 / 200), flush=True)
