import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
def make(V, H, B=512, N=8192):
    rs = np.random.RandomState(0)
    data = (rs.uniform(size=(N, V)) < 0.13).astype(np.float32)
    rbm = mdbn_amd.RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.1, k=1, batch_size=B, weightcost=2e-4)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data))
    perm = torch.from_numpy(rs.permutation(N)).to(eng.device)
    def run(n):
        for it in range(n):
            mb = it % (N // B)
            fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.5)
    return run
for V, H in ((400, 40), (100, 128), (256, 200)):
    run = make(V, H)
    run(20); eng.synchronize()
    t0 = time.perf_counter(); run(200); t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
    print("V=%d H=%d: host enqueue %.1f us/step, total %.1f us/step" % (V, H, (t1 - t0) * 5e3, (t2 - t0) * 5e3), flush=True)
run = make(100, 128)
run(20); eng.synchronize()
pr = cProfile.Profile(); pr.enable(); run(200); pr.disable(); eng.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
