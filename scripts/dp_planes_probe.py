"""Probe: statistics + separate update (the data-parallel code path) against the fused train step, plane path against
f32-operand path, with and without the float32 copies, at batch 512 and 1024 of the c2 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
from mdbn_amd import RngAddr
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H = 4096, 1024
for B in (512, 1024):
    rs = np.random.RandomState(0)
    data = eng.to_device(rs.normal(size=(2 * B, V)).astype(np.float32))
    res = {}
    for planes in (0, 1):
        for keep in (0, 1):
            for path in ("fused", "split"):
                eng.set_option("gemm_planes", planes); eng.keep_f32 = bool(keep)
                rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(3), engine=eng)
                for t in range(3):
                    idx = torch.arange(B, device=eng.device) + (t % 2) * B
                    a = (rbm.W.tensor, rbm.W_speed.tensor, None, rbm.hbias.tensor, rbm.hbias_speed.tensor, rbm.vbias.tensor, rbm.vbias_speed.tensor)
                    if path == "fused":
                        eng.cd_train_step(data, idx, *a, True, 1, RngAddr(3, 0, t, 0, 0), 0.002, 0.0, 0.1, 0.0, 0.3, B, B, 1.0)
                    else:
                        stats, _ = eng.cd_step(data, idx, rbm.W.tensor, rbm.hbias.tensor, rbm.vbias.tensor, True, 1, RngAddr(3, 0, t, 0, 0))
                        eng.apply_update(*a, stats, 0.002, 0.0, 0.1, 0.0, 0.3, B, B, 1.0, phase=0, ldv=V)
                res[(planes, keep, path)] = (rbm.W.get_value(), rbm.W_speed.get_value())
    ref = res[(0, 1, "fused")]
    for k, v in sorted(res.items()):
        print("B=%d planes=%d keep=%d %-5s  max|dW| %.3e  max|dWs| %.3e  (|Ws|max %.3f)" % ((B,) + k + (np.abs(v[0] - ref[0]).max(), np.abs(v[1] - ref[1]).max(), np.abs(ref[1]).max())))
