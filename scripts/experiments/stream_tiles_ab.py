"""Tile shapes of the streaming bf16x6 kernel (csrc/mdbn_stream.hip) on one layer: step time under stream_mi / stream_ni
overrides (0 = the library's own choice).   MDBN_AB_SHAPE=2048,400,512,5,1 python scripts/experiments/stream_tiles_ab.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, K, GAUSS = [int(x) for x in os.environ.get("MDBN_AB_SHAPE", "2048,400,512,5,1").split(",")]
N = 32768
g = torch.Generator(device="cpu").manual_seed(0)
if GAUSS:
    data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.001, k=K, lambda_2=0.1, batch_size=B)
else:
    data = mdbn_amd.shared((torch.rand((N, V), generator=g) < 0.3).float().to(eng.device))
    rbm = mdbn_amd.RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.05, k=K, weightcost=2e-4, batch_size=B)
fn = mdbn_amd.function(up, data)
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
def run(n):
    for it in range(n):
        mb, nb = it % (N // B), (it + 1) % (N // B)
        fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0, next_indexes=perm[nb * B:(nb + 1) * B])
cases = [(0, 0), (1, 1), (2, 1), (2, 2)]
res = {c: [] for c in cases}
run(20); eng.synchronize()
for rnd in range(4):
    for c in cases:
        eng.set_option("stream_mi", c[0]); eng.set_option("stream_ni", c[1])
        run(5); eng.synchronize()
        t0 = time.perf_counter(); run(100); eng.synchronize()
        res[c].append((time.perf_counter() - t0) * 1e4)
for c in cases:
    eng.set_option("stream_mi", c[0]); eng.set_option("stream_ni", c[1])
    run(5); eng.synchronize()
    eng.kernel_timing(True); run(50); eng.synchronize()
    groups = {}
    for ms, alg, pipe, kind in eng.kernel_timing_detail():
        groups.setdefault(kind, []).append(ms)
    eng.kernel_timing(False)
    print("V=%d H=%d B=%d k=%d mi=%d ni=%d: median %.1f us/step | %s" % (V, H, B, K, c[0], c[1], np.median(res[c]),
          ", ".join("kind %d: %.1f us x%d" % (k, 1e3 * np.mean(t), len(t) // 50) for k, t in sorted(groups.items()))))
