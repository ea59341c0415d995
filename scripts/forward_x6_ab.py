#!/usr/bin/env python
"""Large forward passes (get_output / lower-layer cache): bf16x6 GEMM + epilogue kernel vs the exact-f32
kernel with its fused activation epilogue."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
from mdbn_amd.engine import RngAddr
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
for (rows, V, H) in [(4096, 4096, 1024), (32768, 4096, 1024), (32768, 1024, 256), (8192, 16384, 1024)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn((rows, V), generator=g).to(eng.device)
    W = eng.to_device((0.01 * np.random.RandomState(0).randn(V, H)).astype(np.float32))
    hb = eng.to_device(np.zeros(H, np.float32))
    row = {"rows": rows, "V": V, "H": H}
    for rep in range(2):
        for mode in (3, 0):
            eng.set_option("gemm_bf16x6", mode)
            for _ in range(3): eng.propup(x, W, hb, want_pre=False, want_sample=False)
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
            for _ in range(n): eng.propup(x, W, hb, want_pre=False, want_sample=False)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / n * 1e6
            key = "bf16x6_us" if mode else "exact_fused_us"
            row[key] = round(min(us, row.get(key, 1e30)), 1)
    print(json.dumps(row), flush=True)
eng.set_option("gemm_bf16x6", 3)
