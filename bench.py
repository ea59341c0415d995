#!/usr/bin/env python
"""Headline benchmark: CD-1 training steps of the Gaussian-Bernoulli RBM 4096 -> 1024 at
batch 512 per GPU (BASELINE.json configs[1]; configs[2] when launched with N > 1 ranks).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one call of the compiled step function of reference src/rbm.py:258-376 as
src/dbn.py:302-312 builds it: minibatch gather, positive phase, one gibbs_hvh, the
statistics GEMM, (all-reduce over ranks,) parameter update and monitoring cost -- nothing
skipped.  Inputs are synthetic N(0,1) rows already resident in HBM.  Rank 0 prints ONE JSON
line (see the driver contract) with `roofline` (dominant kernel = the f32 MFMA GEMM, timed
by HIP events on its stream) and `cpu_baseline` (the numpy float32 oracle, timed on the
host cores: Theano is not installable offline, so this is kind "port").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V, H, B_PER_GPU, N_DATA, K_GIBBS = 4096, 1024, 512, 32768, 1
LR, LAMBDA_1, LAMBDA_2 = 0.001, 0.0, 0.1          # lr 0.005 (MDBN.py:49) diverges on this synthetic shape; see DESIGN.md
MFMA_F32_PEAK_TFLOPS = 157.3                      # MI355X_MICROARCH.md: f32-input MFMA, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0                    # MI355X_MICROARCH.md: bf16 MFMA, dense


def gemm_traffic_bytes():
    """HBM bytes per GEMM launch from the committed PMC profile (FETCH_SIZE / WRITE_SIZE collected
    in separate rocprofv3 passes and corrected as MI355X_MICROARCH.md prescribes); counters cannot
    be read from inside this process, so the figure is the profiled one, or null if absent."""
    path = os.path.join(ROOT, "profiles", "r01w_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["gemm_avg_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(budget_s=20.0):
    """NumPy float32 restatement of the same step (oracle/rbm_np.py), all host cores via BLAS."""
    from oracle import rbm_np
    from oracle.philox_np import PhiloxDraws
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    rs = np.random.RandomState(123)
    rs.randint(2 ** 30)
    st = rbm_np.RBMState(V, H, W=rbm_np.init_W(rs, V, H, np.float32), dtype=np.float32, gauss=True)
    n_rows = 4 * B_PER_GPU
    data = np.random.default_rng(0).standard_normal((n_rows, V), dtype=np.float32)
    draws = [PhiloxDraws(1, 0, t) for t in range(4)]
    pre = [{0: d.u(0, B_PER_GPU, H), 2: d.u(2, B_PER_GPU, H)} for d in draws]   # RNG not timed
    done, t0 = 0, None
    for it in range(10000):
        if it == 3:
            t0 = time.perf_counter()
        idx = np.arange(B_PER_GPU) + (it % 4) * B_PER_GPU
        rbm_np.cd_step(st, data[idx], rbm_np.ArrayDraws(pre[it % 4]), lr=LR, k=K_GIBBS, lambda_1=LAMBDA_1,
                       lambda_2=LAMBDA_2, batch_size=B_PER_GPU, momentum=0.0)
        if t0 is not None:
            done += 1
            if time.perf_counter() - t0 > budget_s and done >= 5:
                break
    dt = time.perf_counter() - t0
    return {"value": done * B_PER_GPU / dt, "unit": "samples/s", "cores": int(threads), "kind": "port",
            "steps_per_s": done / dt,
            "sample": "%d CD-1 steps of the same GRBM 4096->1024, B=512, numpy float32 oracle "
                      "(BLAS threads=%d, host cpus=%d), injected uniforms" % (done, threads, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import mdbn_amd
    from mdbn_amd import dist

    rank, local_rank, world = dist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
    dev = eng.device
    B_global = B_PER_GPU * world

    # synthetic z-scored features, identical on every rank
    g = torch.Generator(device="cpu").manual_seed(0)
    data = torch.randn((N_DATA, V), generator=g, dtype=torch.float32).to(dev)
    train_set_x = mdbn_amd.shared(data, engine=eng)
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), engine=eng)
    _, updates = rbm.get_cost_updates(lr=LR, k=K_GIBBS, lambda_1=LAMBDA_1, lambda_2=LAMBDA_2,
                                      batch_size=B_global)
    step_fn = mdbn_amd.function(updates, train_set_x)
    perm = torch.from_numpy(np.random.RandomState(1).permutation(N_DATA).astype(np.int64)).to(dev)
    n_mb = N_DATA // B_global

    def run(n, first):
        cost = None
        for it in range(first, first + n):
            mb = it % n_mb
            cost = step_fn(indexes=perm[mb * B_global:(mb + 1) * B_global], momentum=0.0)
        return cost

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    run(args.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    cost = run(args.steps, args.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_cost = float(cost)

    # dominant kernel: the MFMA GEMM (4 launches per CD-1 step); average duration from HIP events on
    # the launch stream over the same K steps
    eng.kernel_timing(True)
    run(args.steps, args.warmup + args.steps)
    torch.cuda.synchronize(dev)
    n_launch, gemm_ms = eng.kernel_timing_read()
    eng.kernel_timing(False)

    # the same K steps with the GEMMs forced onto the exact-f32 MFMA (v_mfma_f32_32x32x2_f32): by default
    # they run on the bf16 matrix pipe with three-way split operands and f32 accumulation (f32 accuracy,
    # DESIGN.md section 3); both numbers are reported
    exact_ms = exact_gemm_us = None
    if world == 1:
        eng.set_option("gemm_bf16x6", 0)
        run(args.warmup, 0)
        barrier()
        t0 = time.perf_counter()
        run(args.steps, args.warmup)
        barrier()
        exact_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        eng.kernel_timing(True)
        run(args.steps, args.warmup + args.steps)
        torch.cuda.synchronize(dev)
        n2, ms2 = eng.kernel_timing_read()
        eng.kernel_timing(False)
        exact_gemm_us = 1e3 * ms2 / max(n2, 1)
        eng.set_option("gemm_bf16x6", 3)

    # free-energy parity of the trained model vs the float64 oracle (north star: <= 1e-4 rel)
    fe_rel = fe_rel_elem = None
    if rank == 0:
        from oracle import rbm_np
        st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), hbias=rbm.hbias.get_value(),
                             vbias=rbm.vbias.get_value(), gauss=True)
        x = data[:B_PER_GPU].cpu().numpy()
        F = rbm.free_energy(x).get_value()
        F_o = rbm_np.free_energy(st, x.astype(np.float64))
        # relative to the largest |F| of the batch (a row whose two O(V) terms nearly cancel would
        # otherwise dominate with a meaningless ratio); the element-wise maximum is reported too
        fe_rel = float(np.max(np.abs(F - F_o)) / np.max(np.abs(F_o)))
        fe_rel_elem = float(np.max(np.abs(F - F_o) / np.abs(F_o)))

    if rank != 0:
        return
    steps_per_s = args.steps / elapsed
    # algorithmic FLOPs: 2*B*V*H per product, 2k+3 products per CD-k step.  The
    # engine issues them as 2k+2 launches (the two statistic products run as ONE GEMM over
    # the stacked batch), so the per-launch figure is the step's FLOPs / launches per step.
    flop_per_step = 2.0 * B_PER_GPU * V * H * (2 * K_GIBBS + 3)
    launches_per_step = n_launch / float(args.steps) if n_launch else float(2 * K_GIBBS + 2)
    flop_per_gemm = flop_per_step / launches_per_step
    avg_gemm_s = gemm_ms / 1e3 / max(n_launch, 1)
    achieved = flop_per_gemm / avg_gemm_s / 1e12 if n_launch else None
    out = {
        "metric": "CD-k Gibbs steps/sec (samples/sec), GRBM 4096->1024 CD-1",
        "value": steps_per_s * B_global * K_GIBBS,
        "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "arithmetic": "f32 operands split exactly into 3 bf16 pieces, 6 products on v_mfma_f32_32x32x16_bf16, f32 accumulate "
                      "(error vs float64 = rocBLAS sgemm's); exact_f32_mfma_* = the same step on v_mfma_f32_32x32x2_f32",
        "exact_f32_mfma_ms_per_step": exact_ms,
        "exact_f32_mfma_value": (B_global * K_GIBBS * 1e3 / exact_ms) if exact_ms else None,
        "config": {"workload": "GRBM 4096->1024 CD-1, batch %d per GPU, fp32, N(0,1) rows resident in HBM "
                               "(BASELINE configs[%d])" % (B_PER_GPU, 1 if world == 1 else 2),
                   "global_batch": B_global, "k": K_GIBBS, "n_data": N_DATA,
                   "parallelism": "dp%d" % world},
        "cd_steps_per_s": steps_per_s,
        "step_tflops": flop_per_step * world * steps_per_s / 1e12,          # whole job
        "step_frac_of_mfma_f32_peak": flop_per_step * steps_per_s / 1e12 / MFMA_F32_PEAK_TFLOPS,   # per GPU
        "final_cost": final_cost,
        "free_energy_max_rel_err_vs_f64_oracle": fe_rel,
        "free_energy_max_elementwise_rel_err": fe_rel_elem,
        # achieved = ALGORITHMIC f32 FLOPs per launch / average launch duration; peak = the dense f32 MFMA
        # peak (the dtype's).  The default kernel issues 6 bf16 products per algorithmic product on the
        # bf16 pipe (2.5 PFLOP/s dense): pipe_frac prices that work against that pipe.
        "roofline": {"bound": "mfma", "kernel": "gemm_bf16x6_kernel (v_mfma_f32_32x32x16_bf16, 3-way split operands)",
                     "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": (achieved / MFMA_F32_PEAK_TFLOPS) if achieved else None,
                     "traffic": gemm_traffic_bytes(),
                     "launches_timed": n_launch, "launches_per_step": launches_per_step, "avg_launch_us": 1e6 * avg_gemm_s,
                     "algorithmic_flop_per_launch": flop_per_gemm,
                     "pipe": "bf16 MFMA", "pipe_flop_per_launch": 6 * flop_per_gemm, "pipe_peak": MFMA_BF16_PEAK_TFLOPS,
                     "pipe_frac": (6 * achieved / MFMA_BF16_PEAK_TFLOPS) if achieved else None,
                     "exact_f32_mfma_avg_launch_us": exact_gemm_us,
                     "exact_f32_mfma_frac": (flop_per_gemm / (exact_gemm_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS)
                                            if exact_gemm_us else None},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
