#!/usr/bin/env python
"""EXPERIMENT: GEMM on pre-split bf16 planes fed by LDS-DMA (scripts/experiments/planes_gemm.hip): checks the
result against float64 and the engine's current bf16x6 kernel (f32 operands split inside the kernel), and times
the CD step's three GEMM shapes plus a steady-state one.
    python scripts/experiments/planes_gemm.py [build]"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(ROOT, "scripts", "libplanes_exp.so")


def build(flags=(), out=so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"] + list(flags) +
                          [os.path.join(here, "planes_gemm.hip"), "-o", out])
    return out


if len(sys.argv) > 1 and sys.argv[1] == "build":
    build()
    sys.exit(0)
ABLATE = len(sys.argv) > 1 and sys.argv[1] == "ablate"
build()

import numpy as np, torch
import mdbn_amd
lib = C.CDLL(so)
vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
lib.exp_split.argtypes = [vp, i64, i64, vp, vp]
lib.exp_gemm.argtypes = [i32, i32, i32, i32, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp, vp]
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
dev = eng.device
ROW, COL = 0, 1


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def split(x):
    rows, ld = x.shape
    P = torch.empty((3, rows, ld), dtype=torch.int16, device=dev)
    assert lib.exp_split(x.data_ptr(), rows, ld, P.data_ptr(), stream()) == 0
    return P


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


res = []
# (name, M, N, K, la, lb, ap, splitk): C[M,N] = A B with A as [M][K] (ROW) or [K][M] (COL), B as [N][K] (ROW) or [K][N] (COL)
cases = [("propup", 512, 1024, 4096, ROW, COL, 3, 8), ("propup_hs", 512, 1024, 4096, ROW, COL, 1, 8),
         ("propdown", 512, 4096, 1024, ROW, ROW, 1, 2), ("propdown3", 512, 4096, 1024, ROW, ROW, 3, 2),
         ("stats", 4096, 1024, 1024, COL, COL, 3, 1), ("steady_up", 4096, 1024, 4096, ROW, COL, 3, 1),
         ("steady_stats", 4096, 4096, 4096, COL, COL, 3, 1)]
VARIANTS = len(sys.argv) > 2 and sys.argv[1] == "variants"
for name, M, N, K, la, lb, ap, sk in ([] if (ABLATE or VARIANTS) else cases):
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((M, K) if la == ROW else (K, M), generator=g).to(dev)
    if ap == 1:
        A = (A > 0).float()
    B = (0.05 * torch.randn((N, K) if lb == ROW else (K, N), generator=g)).to(dev)
    Ap, Bp = split(A), split(B)
    # exactness of the split
    rec = lambda P: sum(((P[i].to(torch.int32) & 0xffff) << 16).view(torch.float32) for i in (2, 1, 0))
    assert torch.equal(rec(Ap), A) and torch.equal(rec(Bp), B), "split is not exact"
    Cs = torch.zeros((sk, M, N), device=dev)
    for lw in (4, 8):
        def run():
            rc = lib.exp_gemm(la, lb, ap, lw, Ap.data_ptr(), A.shape[1], A.numel(), Bp.data_ptr(), B.shape[1], B.numel(),
                              Cs.data_ptr(), N, M * N, M, N, K, sk, stream(), None)
            assert rc == 0, rc
        Cs.zero_()
        run(); torch.cuda.synchronize()
        got = Cs.sum(0).double()
        Am = A.double() if la == ROW else A.double().t()
        Bm = B.double().t() if lb == ROW else B.double()
        ref = Am @ Bm
        err = float((got - ref).abs().max() / ref.abs().max())
        t = timeit(run)
        f = 2.0 * M * N * K
        issued = f * (6 if ap == 3 else 3)
        row = {"case": name, "M": M, "N": N, "K": K, "ap": ap, "splitk": sk, "lw": lw, "us": round(t, 2), "relerr_vs_f64": err,
               "f32eq_tflops": round(f / t / 1e6, 1), "bf16_pipe_frac": round(issued / t / 1e6 / 2500.0, 3)}
        res.append(row); print(json.dumps(row), flush=True)
    # the engine's current kernel on the same operands (f32 in HBM, split in the kernel)
    if name in ("propup", "steady_up"):
        hb = eng.alloc_vector(N)
        eng.set_option("fused_epilogue", 0)
        x = eng.to_device(A); W = eng.to_device(B)
        eng.kernel_timing(True)
        for _ in range(10): eng.propup(x, W, hb, want_mean=False, want_sample=False)
        torch.cuda.synchronize(); n, ms = eng.kernel_timing_read(); eng.kernel_timing(False)
        pre = eng.propup(x, W, hb, want_mean=False, want_sample=False)[0]
        row = {"case": name + "_engine_bf16x6", "us": round(ms * 1e3 / max(n, 1), 2),
               "bitwise_equal_to_planes": bool(torch.equal(pre[:, :N].double(), got))}
        res.append(row); print(json.dumps(row), flush=True)
if ABLATE:
    # where does a stage's time go?  diagnostic builds (wrong results): 1 no DMA, 2 no MFMA, 3 no fragment reads, 4 barriers only
    names = {0: "full", 1: "no_dma", 2: "no_mfma", 3: "no_frag_reads", 4: "consumers_idle"}
    libs = {}
    for abl in names:
        libs[abl] = C.CDLL(build(["-DABL=%d" % abl], so.replace(".so", "_abl%d.so" % abl)))
        libs[abl].exp_gemm.argtypes = lib.exp_gemm.argtypes
    for name, M, N, K, la, lb, ap, sk in cases:
        A = torch.randn((M, K) if la == ROW else (K, M)).to(dev)
        B = (0.05 * torch.randn((N, K) if lb == ROW else (K, N))).to(dev)
        Ap, Bp = split(A), split(B)
        Cs = torch.zeros((sk, M, N), device=dev)
        for lw in (4, 8):
            row = {"case": name, "lw": lw, "stages_per_job": K // sk // 32}
            for abl, nm in names.items():
                f = lambda: libs[abl].exp_gemm(la, lb, ap, lw, Ap.data_ptr(), A.shape[1], A.numel(), Bp.data_ptr(), B.shape[1],
                                               B.numel(), Cs.data_ptr(), N, M * N, M, N, K, sk, stream(), None)
                row[nm + "_us"] = round(timeit(f), 2)
            res.append(row); print(json.dumps(row), flush=True)
if len(sys.argv) > 2 and sys.argv[1] == "variants":
    # python planes_gemm.py variants "-DSCHED=1" "-DSCHED=2 -DFOO=3" ...: full kernels, checked and timed per build
    for vi, flags in enumerate([""] + sys.argv[2:]):
        vlib = C.CDLL(build(flags.split(), so.replace(".so", "_v%d.so" % vi)))
        vlib.exp_gemm.argtypes = lib.exp_gemm.argtypes
        for name, M, N, K, la, lb, ap, sk in cases:
            g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
            A = torch.randn((M, K) if la == ROW else (K, M), generator=g).to(dev)
            if ap == 1:
                A = (A > 0).float()
            B = (0.05 * torch.randn((N, K) if lb == ROW else (K, N), generator=g)).to(dev)
            Ap, Bp = split(A), split(B)
            Cs = torch.zeros((sk, M, N), device=dev)
            ref = (A.double() if la == ROW else A.double().t()) @ (B.double().t() if lb == ROW else B.double())
            for lw in (4, 104):                     # 104 = v_mfma_f32_16x16x32_bf16 variant, 4 loader waves
                f = lambda: vlib.exp_gemm(la, lb, ap, lw, Ap.data_ptr(), A.shape[1], A.numel(), Bp.data_ptr(), B.shape[1],
                                          B.numel(), Cs.data_ptr(), N, M * N, M, N, K, sk, stream(), None)
                Cs.zero_()
                rc = f(); torch.cuda.synchronize()
                err = float((Cs.sum(0).double() - ref).abs().max() / ref.abs().max())
                row = {"variant": flags or "base", "case": name, "lw": lw, "rc": rc, "us": round(timeit(f), 2), "relerr": err}
                nblk = (M // 128) * (N // 128) * sk
                dbg = torch.zeros((nblk, 4), dtype=torch.int64, device=dev)
                for _ in range(20): f()                      # clock under sustained load
                vlib.exp_gemm(la, lb, ap, lw, Ap.data_ptr(), A.shape[1], A.numel(), Bp.data_ptr(), B.shape[1], B.numel(),
                              Cs.data_ptr(), N, M * N, M, N, K, sk, stream(), dbg.data_ptr())
                torch.cuda.synchronize()
                d = dbg.double().median(0).values.tolist()
                nst = K // sk // 32
                row.update({"prologue_cyc": d[0], "cyc_per_stage": round(d[1] / nst, 1), "clock_ghz": round(d[2] / d[3] * 0.1, 3)})
                res.append(row); print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "planes_gemm.json"), "w"), indent=1)
