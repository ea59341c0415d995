#!/usr/bin/env python
"""Per-rank compute cost of the data-parallel step on ONE GPU: the overlapped code path (cd_step ->
async all-reduce -> phase-3 update) with a stand-in group whose all-reduce is a no-op, next to the
fused single-device step.  Shows what the DP path pays before any RCCL traffic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd

class _Work:
    def wait(self): return True

class FakeGroup:
    rank, world_size = 0, 2
    def shard(self, n): return 0, n                # this "rank" owns every row: per-rank work of a 512-row shard
    def all_reduce_sum(self, t, engine=None): return t
    def all_reduce_sum_async(self, t, engine=None): return _Work()

eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, N = 4096, 1024, 512, 32768
g = torch.Generator(device="cpu").manual_seed(0)
data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
fns = {}
for name, grp in (("single-device fused step", None), ("data-parallel path (no-op all-reduce)", FakeGroup())):
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
    fns[name] = mdbn_amd.function(up, data, data_parallel=grp)
def run(fn, n):
    for it in range(n):
        mb = it % (N // B)
        fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0)
res = {k: [] for k in fns}
for k, fn in fns.items():
    run(fn, 20)
eng.synchronize()
for rnd in range(5):
    for k, fn in fns.items():
        run(fn, 5); eng.synchronize()
        t0 = time.perf_counter(); run(fn, 100); eng.synchronize()
        res[k].append((time.perf_counter() - t0) * 1e4)
for k in fns:
    print("%-42s median %.1f us/step" % (k, np.median(res[k])))
