#!/usr/bin/env python
"""Soak of the row feeder: the c2 step for many epochs from a pinned host table (every epoch announced, as the trainers do),
final parameters compared BIT FOR BIT with the device-resident run of the same minibatch sequence.
    python scripts/experiments/host_feed_soak.py [epochs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, N = 4096, 1024, 512, 32768
EPOCHS = int(sys.argv[1]) if len(sys.argv) > 1 else 80
x = torch.randn((N, V), generator=torch.Generator().manual_seed(0)).numpy()
out = []
for resident in ("host", "device"):
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(9))
    _, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
    fn = mdbn_amd.function(up, mdbn_amd.shared(x, resident=resident))
    rs = np.random.RandomState(5)
    t0 = time.perf_counter()
    steps = 0
    for ep in range(EPOCHS):
        perm = rs.permutation(N)
        # ragged last minibatch every third epoch; an early stop (unconsumed announcements) every fifth
        cuts = list(range(0, N, B)) if ep % 3 else list(range(0, N - 100, B))
        batches = [perm[c:c + B] for c in cuts]
        dev = [eng.index_tensor(b) for b in batches]
        fn.announce(dev, host_indexes=batches)
        for i in range(len(batches) - (7 if ep % 5 == 4 else 0)):
            cost = fn(indexes=dev[i], momentum=0.0, next_indexes=dev[i + 1] if i + 1 < len(dev) else None)
            steps += 1
    eng.synchronize()
    dt = time.perf_counter() - t0
    print("%s-resident: %d steps, %.1f us/step, last cost %.6f" % (resident, steps, dt / steps * 1e6, float(cost)), flush=True)
    out.append((rbm.W.get_value(), rbm.W_speed.get_value(), rbm.vbias.get_value(), float(cost)))
same = all(np.array_equal(a, b) for a, b in zip(out[0][:3], out[1][:3])) and out[0][3] == out[1][3]
print("host-resident == device-resident, bit for bit:", same)
assert same and np.isfinite(out[0][0]).all()
