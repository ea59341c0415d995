#!/usr/bin/env python
"""Sweep the skinny-GEMM knobs (register batch depth, single-launch K limit) on small-batch steps."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
res = []
for (V, H, B, gauss) in [(784, 500, 20, 0), (16384, 400, 20, 1), (2048, 400, 20, 1), (400, 40, 20, 0),
                         (4096, 1024, 20, 1), (20000, 1000, 10, 1), (1024, 256, 32, 0)]:
    N = 2048
    rs = np.random.RandomState(0)
    data = rs.randn(N, V).astype(np.float32) if gauss else (rs.rand(N, V) < 0.2).astype(np.float32)
    row = {"V": V, "H": H, "B": B}
    for rep in range(2):
        for u in (4, 8):
            for maxk in (0, 1024, 2048):
                eng.set_option("skinny_u", u); eng.set_option("skinny_fused_max_k", maxk)
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
                key = "u%d_k%d" % (u, maxk)
                row[key] = round(min(us, row.get(key, 1e30)), 1)
    res.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/skinny_tune.json", "w"), indent=1)
