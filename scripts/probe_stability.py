"""Cost trajectory of the bench workload for a few learning rates (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, N = 4096, 1024, 512, 32768
g = torch.Generator(device="cpu").manual_seed(0)
data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
for lr, scale in ((0.005, 1.0), (0.001, 1.0), (0.005, 0.25)):
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    if scale != 1.0:
        rbm.W.tensor.mul_(scale)
    _, up = rbm.get_cost_updates(lr=lr, k=1, lambda_1=0.0, lambda_2=0.1, batch_size=B)
    fn = mdbn_amd.function(up, data)
    out = []
    for it in range(600):
        mb = it % (N // B)
        c = fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0)
        if it % 50 == 0 or it == 599:
            out.append("%d:%.4g|W|max=%.3g" % (it, float(c), float(rbm.W.tensor.abs().max())))
    print("lr", lr, "Wscale", scale, " ".join(out), flush=True)
