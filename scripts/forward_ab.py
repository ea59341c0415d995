#!/usr/bin/env python
"""Forward-pass A/B: fused MFMA-accumulator activation epilogue vs slab + epilogue kernel.
    python scripts/forward_ab.py      (on the GPU box)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mdbn_amd
from mdbn_amd.engine import RngAddr

eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
res = []
for (rows, V, H) in [(4096, 4096, 1024), (32768, 4096, 1024), (32768, 1024, 512), (8192, 16384, 1024), (512, 200, 1024)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn((rows, V), generator=g).to(eng.device)
    W = eng.to_device((0.01 * np.random.RandomState(0).randn(V, H)).astype(np.float32))
    hb = eng.to_device(np.zeros(H, np.float32))
    row = {"rows": rows, "V": V, "H": H}
    for rep in range(2):
        for fused in (1, 0):
            eng.set_option("fused_epilogue", fused)
            for _ in range(3):
                eng.propup(x, W, hb, rng=RngAddr(1, 0, 0, 0))
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 20
            for _ in range(n):
                eng.propup(x, W, hb, rng=RngAddr(1, 0, 0, 0))
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / n * 1e6
            row["fused_us" if fused else "unfused_us"] = min(us, row.get("fused_us" if fused else "unfused_us", 1e30))
    row["tflops_fused"] = 2.0 * rows * V * H / row["fused_us"] / 1e6
    row["tflops_unfused"] = 2.0 * rows * V * H / row["unfused_us"] / 1e6
    res.append(row); print(json.dumps(row), flush=True)
eng.set_option("fused_epilogue", 1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/forward_ab.json", "w"), indent=1)
