#!/usr/bin/env python
"""Upper bound of hipGraph replay for launch-bound shapes: one CD-1 train step captured by stream
capture and replayed (fixed arguments: timing only) vs the eager C call vs the full Python step."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import mdbn_amd
from mdbn_amd import _lib
from mdbn_amd.engine import RngAddr

eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
res = []
for (V, H, B, gauss) in [(400, 40, 20, True), (1000, 400, 20, True), (16384, 400, 20, True), (784, 500, 20, False),
                         (100, 128, 512, False), (4096, 1024, 512, True)]:
    N = 4096
    rs = np.random.RandomState(0)
    data = rs.randn(N, V).astype(np.float32) if gauss else (rs.rand(N, V) < 0.2).astype(np.float32)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.001, k=1, batch_size=B, lambda_2=0.1)
    shared = mdbn_amd.shared(data)
    fn = mdbn_amd.function(up, shared)
    perm = torch.from_numpy(rs.permutation(N)).to(eng.device)
    nmb = N // B
    def run_py(n):
        for it in range(n):
            mb = it % nmb
            fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.5)
    run_py(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run_py(300); torch.cuda.synchronize(); py_us = (time.perf_counter() - t0) / 300 * 1e6

    # fixed-argument eager C call
    idx = perm[:B].contiguous()
    Wd, hb, vb = rbm.W.tensor, rbm.hbias.tensor, rbm.vbias.tensor
    Ws, hs, vs = rbm.W_speed.tensor, rbm.hbias_speed.tensor, rbm.vbias_speed.tensor
    a, stats, sc, keep = eng._cd_args(shared.tensor, idx, Wd, hb, vb, gauss, 1, RngAddr(1, 0, 5, 0), None, False, 0, False)
    u, cost = eng._update_args(Wd, Ws, None, hb, hs, vb, vs, stats, 0.0, 0.0, 0.1, 0.0, 0.5, B, B, 1.0, 0, a.ldv)
    def run_c(n):
        for _ in range(n):
            _lib.check(eng.lib.mdbn_cd_train_step(eng.ctx, eng._stream(), C.byref(a), C.byref(u)), "step")
    run_c(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run_c(300); torch.cuda.synchronize(); c_us = (time.perf_counter() - t0) / 300 * 1e6

    # captured graph of the same call
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run_c(3)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            _lib.check(eng.lib.mdbn_cd_train_step(eng.ctx, eng._stream(), C.byref(a), C.byref(u)), "step")
    torch.cuda.synchronize()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): g.replay()
    torch.cuda.synchronize(); g_us = (time.perf_counter() - t0) / 300 * 1e6
    row = {"V": V, "H": H, "B": B, "python_step_us": py_us, "c_call_us": c_us, "graph_replay_us": g_us}
    res.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/graph_probe.json", "w"), indent=1)
