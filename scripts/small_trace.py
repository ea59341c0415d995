#!/usr/bin/env python
"""Run 60 CD-1 steps of one small shape (for rocprofv3 --kernel-trace; see scripts/trace_gaps.py).
    python scripts/small_trace.py V H B gauss"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
V, H, B, gauss = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
N = 4096
rs = np.random.RandomState(0)
data = rs.randn(N, V).astype(np.float32) if gauss else (rs.rand(N, V) < 0.2).astype(np.float32)
rbm = (mdbn_amd.GRBM if gauss else mdbn_amd.RBM)(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
_, up = rbm.get_cost_updates(lr=0.001, k=1, batch_size=B, lambda_2=0.1)
fn = mdbn_amd.function(up, mdbn_amd.shared(data))
perm = torch.from_numpy(rs.permutation(N)).to(eng.device)
for it in range(60):
    mb = it % (N // B)
    fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.5)
torch.cuda.synchronize()
