#!/usr/bin/env python
"""B = 512 steps of small layers: register-streaming kernel above 64 rows on (MAC limit) / off (0)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
res = []
for (V, H, B, gauss, k) in [(1024, 256, 512, 0, 1), (400, 40, 512, 0, 1), (512, 40, 512, 1, 5), (256, 200, 512, 1, 1),
                            (100, 128, 512, 0, 1), (2048, 400, 512, 1, 1), (784, 500, 100, 0, 1), (500, 500, 128, 0, 1)]:
    N = 4096
    rs = np.random.RandomState(0)
    data = rs.randn(N, V).astype(np.float32) if gauss else (rs.rand(N, V) < 0.2).astype(np.float32)
    row = {"V": V, "H": H, "B": B, "k": k}
    for rep in range(2):
        for macs in (0, 32 << 20):
            eng.set_option("skinny_max_macs", macs)
            rbm = (mdbn_amd.GRBM if gauss else mdbn_amd.RBM)(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
            _, up = rbm.get_cost_updates(lr=0.001, k=k, batch_size=B, lambda_2=0.1)
            fn = mdbn_amd.function(up, mdbn_amd.shared(data))
            perm = torch.from_numpy(rs.permutation(N)).to(eng.device)
            def run(n):
                for it in range(n):
                    mb = it % (N // B)
                    fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.5)
            run(20); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(300); torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 300 * 1e6
            key = "macs_%dM" % (macs >> 20)
            row[key] = round(min(us, row.get(key, 1e30)), 1)
    res.append(row); print(json.dumps(row), flush=True)
eng.set_option("skinny_max_macs", 32 << 20)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/skinny_macs_ab.json", "w"), indent=1)
