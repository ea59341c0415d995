#!/usr/bin/env python
"""Per-column error of the positive-phase probabilities of a ragged layer (H = 1000 on ld = 1024) against float64, on the
plane path and on the f32-operand path: is the last live column treated differently?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mdbn_amd
from mdbn_amd import RngAddr
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
eng.keep_f32 = True
V, H, B, LD = 4096, 1000, 512, 1024
rs = np.random.RandomState(0)
W = (rs.uniform(-1, 1, size=(V, H)) * 4 * np.sqrt(6.0 / (V + H))).astype(np.float32)
hb = rs.normal(0, 0.2, H).astype(np.float32); vb = rs.normal(0, 0.2, V).astype(np.float32)
x = (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
ref = 1.0 / (1.0 + np.exp(-(x.astype(np.float64) @ W.astype(np.float64) + hb)))
for ld in (LD, 1000):
    dW = eng.alloc_matrix(V, H, ld=ld); dW.copy_(torch.from_numpy(W))
    dhb, dvb, dx = [eng.to_device(a) for a in (hb, vb, x)]
    for planes in (1, 0):
        eng.set_option("gemm_planes", planes)
        eng.kernel_timing(True)
        stats, sc = eng.cd_step(dx, None, dW, dhb, dvb, False, 1, RngAddr(11, 1, 2, 0, 0))
        eng.synchronize()
        kinds = sorted(set(k for _, _, _, k in eng.kernel_timing_detail()))
        eng.kernel_timing(False)
        ph = sc.P2[:B].cpu().numpy()[:, :H].astype(np.float64)
        err = np.abs(ph - ref).max(axis=0)
        print("ld %d planes %d kinds %s: max err %.2e at col %d; last 6 cols %s; median col err %.2e" %
              (ld, planes, kinds, err.max(), err.argmax(), " ".join("%.1e" % e for e in err[-6:]), np.median(err)))
