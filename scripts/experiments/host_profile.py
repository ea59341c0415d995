#!/usr/bin/env python
"""Where the HOST time of a small-layer step goes (cProfile of the step function, GPU work asynchronous).
    MDBN_AB_SHAPE=100,24,20,1,0 python scripts/experiments/host_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, K, GAUSS = [int(x) for x in os.environ.get("MDBN_AB_SHAPE", "100,24,20,1,0").split(",")]
N = 4096
g = torch.Generator(device="cpu").manual_seed(0)
data = mdbn_amd.shared((torch.randn((N, V), generator=g) if GAUSS else (torch.rand((N, V), generator=g) < 0.3).float()).to(eng.device))
cls = mdbn_amd.GRBM if GAUSS else mdbn_amd.RBM
rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
_, up = rbm.get_cost_updates(lr=0.01, k=K, batch_size=B, **({"lambda_2": 0.1} if GAUSS else {"weightcost": 2e-4}))
fn = mdbn_amd.function(up, data)
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
views = [perm[i * B:(i + 1) * B] for i in range(N // B)]
def run(n):
    for it in range(n):
        fn(indexes=views[it % len(views)], momentum=0.5, next_indexes=views[(it + 1) % len(views)])
run(200); eng.synchronize()
t0 = time.perf_counter(); run(4000); t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
print("shape %s: host %.1f us per call (enqueue only), %.1f us per step with the GPU drained" % (os.environ.get("MDBN_AB_SHAPE"), (t1 - t0) / 4000 * 1e6, (t2 - t0) / 4000 * 1e6))
pr = cProfile.Profile(); pr.enable(); run(3000); pr.disable(); eng.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
