#!/usr/bin/env python
"""Interleaved A/B of a library option on CD-1 steps of several small shapes.
    python scripts/option_ab.py OPTION v1 v2"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
opt, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
for (V, H, B, gauss) in [(784, 500, 20, 0), (16384, 400, 20, 1), (100, 128, 512, 0), (400, 40, 512, 0), (1024, 256, 512, 0), (2048, 400, 512, 1)]:
    N = 4096
    rs = np.random.RandomState(0)
    data = rs.randn(N, V).astype(np.float32) if gauss else (rs.rand(N, V) < 0.2).astype(np.float32)
    rbm = (mdbn_amd.GRBM if gauss else mdbn_amd.RBM)(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.001, k=1, batch_size=B, lambda_2=0.1)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data))
    perm = torch.from_numpy(rs.permutation(N)).to(eng.device)
    def run(n):
        for it in range(n):
            mb = it % (N // B)
            fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.5)
    res = {v: [] for v in values}
    run(20); torch.cuda.synchronize()
    for rnd in range(4):
        for v in values:
            eng.set_option(opt, v)
            run(5); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(200); torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / 200 * 1e6)
    print(json.dumps({"V": V, "H": H, "B": B, **{"%s=%d" % (opt, v): round(float(np.median(res[v])), 1) for v in values}}), flush=True)
