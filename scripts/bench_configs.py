"""Step throughput of the non-headline BASELINE configs (parity-test shapes), GPU vs the numpy
float32 oracle on the host.  Not the driver's bench (that is bench.py): a table for DESIGN.md."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
from oracle import rbm_np
from oracle.philox_np import PhiloxDraws
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())

def gpu_steps(cls, V, H, B, k, hp, N, steps=300):
    rs = np.random.RandomState(0)
    gauss = cls is mdbn_amd.GRBM
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.13).astype(np.float32)
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(k=k, batch_size=B, **hp)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data))
    perm = torch.from_numpy(rs.permutation(N)).to(eng.device)
    nmb = N // B
    def run(n):           # (the trainers announce the next minibatch: dbn.py's epoch order is drawn up front)
        for it in range(n):
            mb, nx = it % nmb, (it + 1) % nmb
            fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.5, next_indexes=perm[nx * B:(nx + 1) * B])
    run(20); eng.synchronize()
    t0 = time.perf_counter(); run(steps); eng.synchronize()
    return steps / (time.perf_counter() - t0)

def cpu_steps(gauss, V, H, B, k, hp, budget=6.0):
    rs = np.random.RandomState(0)
    data = rs.normal(size=(4 * B, V)).astype(np.float32) if gauss else (rs.uniform(size=(4 * B, V)) < 0.13).astype(np.float32)
    st = rbm_np.RBMState(V, H, W=rbm_np.init_W(np.random.RandomState(123), V, H, np.float32), dtype=np.float32, gauss=gauss)
    if hp.get("weightcost"): st.freeze_W0()
    draws = {d: rs.uniform(size=(B, H if d % 2 == 0 else V)).astype(np.float32) for d in range(2 * k + 1)}
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        rbm_np.cd_step(st, data[(n % 4) * B:(n % 4 + 1) * B], rbm_np.ArrayDraws(draws), k=k, batch_size=B, momentum=0.5, **hp)
        n += 1
    return n / (time.perf_counter() - t0)

R, G = mdbn_amd.RBM, mdbn_amd.GRBM
cases = [
    ("c1 RBM 784->500 B=20 k=1 (reference shape)", R, 784, 500, 20, 1, dict(lr=0.1, weightcost=2e-4), 4000),
    ("c2 GRBM 4096->1024 B=512 k=1 (headline)", G, 4096, 1024, 512, 1, dict(lr=0.001, lambda_2=0.1), 8192),
    ("c4 L1 RBM 1024->256 B=512 k=1", R, 1024, 256, 512, 1, dict(lr=0.1, weightcost=2e-4), 8192),
    ("c5 GE GRBM 2048->400 B=512 k=1", G, 2048, 400, 512, 1, dict(lr=0.002, lambda_2=0.1), 8192),
    ("c5 GE GRBM 2048->400 B=512 k=5", G, 2048, 400, 512, 5, dict(lr=0.002, lambda_2=0.1), 8192),
    ("c5 GE L1 RBM 400->40 B=512 k=1", R, 400, 40, 512, 1, dict(lr=0.1, weightcost=2e-4), 8192),
    ("c5 miRNA GRBM 512->40 B=512 k=5", G, 512, 40, 512, 5, dict(lr=0.002, lambda_2=0.1), 8192),
    ("c5 SM GRBM 256->200 B=512 k=1", G, 256, 200, 512, 1, dict(lr=0.002, lambda_2=0.1), 8192),
    ("c5 SM GRBM 256->200 B=512 k=5", G, 256, 200, 512, 5, dict(lr=0.002, lambda_2=0.1), 8192),
    ("c5 joint RBM 100->128 B=512 k=1", R, 100, 128, 512, 1, dict(lr=0.1, weightcost=2e-4), 8192),
    ("real GE shape GRBM 19937->400 B=20 k=1", G, 19937, 400, 20, 1, dict(lr=0.0005, lambda_2=0.1), 170),
]
gpu = {}
for name, cls, V, H, B, k, hp, N in cases:          # all GPU runs first: the BLAS-threaded CPU
    gpu[name] = gpu_steps(cls, V, H, B, k, hp, max(N, B))   # baseline disturbs host-bound shapes
out = []
NO_CPU = "--no-cpu" in sys.argv
F32_MFMA_PEAK_TF = 157.3          # dense f32 MFMA peak (MI355X_MICROARCH.md): the roof of the north-star metric's arithmetic
for name, cls, V, H, B, k, hp, N in cases:
    g = gpu[name]
    c = float("nan") if NO_CPU else cpu_steps(cls is G, V, H, B, k, hp)
    flops = 2.0 * B * V * H * (2 * k + 3)
    row = dict(config=name, gpu_steps_per_s=g, gpu_us_per_step=1e6 / g, gpu_samples_per_s=g * B, gpu_tflops=flops * g / 1e12,
               frac_of_f32_mfma_peak=flops * g / 1e12 / F32_MFMA_PEAK_TF, cpu_steps_per_s=c, speedup=g / c)
    out.append(row)
    print("%-46s GPU %8.0f steps/s (%7.1f us, %6.2f TF)   CPU oracle %8.1f steps/s   x%.0f" %
          (name, g, 1e6 / g, row["gpu_tflops"], c, g / c), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_configs.json"), "w"), indent=1)
