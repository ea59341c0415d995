#!/usr/bin/env python
"""Training steps of a ragged layer (4096 -> 1000 on ld 1024) on the plane path and on the f32-operand path: which columns of
hbias / W_speed differ, after how many steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B = 4096, 1000, 512
for steps in (1, 2, 4):
    runs = []
    for planes in (1, 0):
        eng.set_option("gemm_planes", planes)
        rs = np.random.RandomState(7)
        N = 4 * B
        data = (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
        rbm = mdbn_amd.RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5), engine=eng)
        _, up = rbm.get_cost_updates(k=1, batch_size=B, lr=0.05, weightcost=2e-4)
        fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel=None)
        for _ in range(steps):
            fn(indexes=rs.permutation(N)[:B], momentum=0.5)
        runs.append((rbm.hbias.get_value(), rbm.hbias_speed.get_value(), rbm.W_speed.get_value(), rbm.W.get_value()))
    a, b = runs
    for name, x, y in zip(("hbias", "hbias_speed", "W_speed", "W"), a, b):
        d = np.abs(x - y)
        d = d.max(axis=0) if d.ndim == 2 else d
        top = np.argsort(d)[-4:][::-1]
        print("steps %d %-11s max diff %.2e; worst columns %s (%s); median %.1e" % (steps, name, d.max(), top.tolist(),
              " ".join("%.1e" % d[t] for t in top), np.median(d)))
