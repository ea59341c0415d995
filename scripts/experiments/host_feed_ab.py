#!/usr/bin/env python
"""The c2 step fed from a pinned host table through the row feeder (mdbn_feeder_*): copy streams 1 / 2, gather threads.
    python scripts/experiments/host_feed_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, N = 4096, 1024, 512, 32768
x = torch.randn((N, V), generator=torch.Generator().manual_seed(0)).numpy()
table = mdbn_amd.shared(x, resident="host")
perm = np.random.RandomState(1).permutation(N)
hperm = torch.from_numpy(perm)
dperm = hperm.to(eng.device)
n_mb = N // B
for streams, threads in ((1, 4), (1, 6), (1, 8), (1, 12), (1, 4), (1, 6), (1, 8)):
    eng.set_option("feed_copy_streams", streams)
    eng.host_feed_threads = threads
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
    fn = mdbn_amd.function(up, table)
    def run(n):
        views = [dperm[(it % n_mb) * B:(it % n_mb + 1) * B] for it in range(n)]
        fn.announce(views, host_indexes=[hperm[(it % n_mb) * B:(it % n_mb + 1) * B] for it in range(n)])
        for it in range(n):
            fn(indexes=views[it], momentum=0.0)
    run(50); eng.synchronize()
    fn._staging["feeder"].stats()
    res = []
    for rep in range(3):
        t0 = time.perf_counter(); run(500); eng.synchronize()
        res.append((time.perf_counter() - t0) / 500 * 1e6)
    print("copy streams %d, gather threads %2d: %.1f us/step (runs %s)  %s" % (
        streams, threads, np.median(res), " ".join("%.1f" % r for r in res), fn._staging["feeder"].stats()), flush=True)
    del fn
