#!/usr/bin/env python
"""Minimum K per split-K job of the tiled GEMM: step time of mid-size layers at B = 512."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
for (V, H, B, gauss) in [(4096, 1024, 512, 1), (2048, 400, 512, 1), (1024, 256, 512, 0), (784, 500, 256, 0), (2000, 1000, 128, 1)]:
    N = 4096
    rs = np.random.RandomState(0)
    data = rs.randn(N, V).astype(np.float32) if gauss else (rs.rand(N, V) < 0.2).astype(np.float32)
    row = {"V": V, "H": H, "B": B}
    for rep in range(2):
        for mk in (128, 256, 512):
            eng.set_option("gemm_min_splitk", mk)
            rbm = (mdbn_amd.GRBM if gauss else mdbn_amd.RBM)(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
            _, up = rbm.get_cost_updates(lr=0.001, k=1, batch_size=B, lambda_2=0.1)
            fn = mdbn_amd.function(up, mdbn_amd.shared(data))
            perm = torch.from_numpy(rs.permutation(N)).to(eng.device)
            def run(n):
                for it in range(n):
                    mb = it % (N // B)
                    fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.5)
            run(20); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(300); torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 300 * 1e6
            key = "mink_%d" % mk
            row[key] = round(min(us, row.get(key, 1e30)), 1)
    print(json.dumps(row), flush=True)
