#!/usr/bin/env python
"""EXPERIMENT: f32-grade GEMM on the bf16 matrix pipe (three-way bf16 split, six MFMA products).
Compiles scripts/experiments/bf16x6_gemm.hip on the GPU box, checks the error against float64 next
to rocBLAS f32 and the engine's exact-f32 MFMA GEMM, and times all three.
    python scripts/experiments/bf16x6_gemm.py"""
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import mdbn_amd

here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libbf16x6.so")
libs = {}
for tag, flags in (("one_acc", []), ("two_acc", ["-DX6_TWO_ACC"])):
    so_t = so.replace(".so", "_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"] + flags +
                          [os.path.join(here, "bf16x6_gemm.hip"), "-o", so_t])
    libs[tag] = C.CDLL(so_t)
    libs[tag].bf16x6_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
lib = libs["one_acc"]
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
dev = eng.device

def split3(x):
    p1 = x.to(torch.bfloat16); r = x - p1.float()
    p2 = r.to(torch.bfloat16); r = r - p2.float()
    p3 = r.to(torch.bfloat16)
    return torch.stack([p1, p2, p3]).contiguous()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

res = []
for (M, N, K) in [(512, 1024, 4096), (4096, 1024, 1024), (4096, 4096, 4096), (16384, 1024, 4096)]:
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g).to(dev)
    w = (0.05 * torch.randn((N, K), generator=g)).to(dev)          # B as [N][K]: C = x w^T
    xa, wb = split3(x), split3(w)
    assert float((xa.float().sum(0) - x).abs().max()) == 0.0, "split is not exact"
    Cm = torch.empty((M, N), device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        rc = lib.bf16x6_gemm(stream, xa.data_ptr(), wb.data_ptr(), Cm.data_ptr(), M, N, K)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    scale = float(ref.abs().max())
    err6 = float((Cm.double() - ref).abs().max()) / scale
    med6 = float((Cm.double() - ref).abs().median())
    C2 = torch.empty((M, N), device=dev)
    rc = libs["two_acc"].bf16x6_gemm(stream, xa.data_ptr(), wb.data_ptr(), C2.data_ptr(), M, N, K); assert rc == 0
    torch.cuda.synchronize()
    err6b = float((C2.double() - ref).abs().max()) / scale
    med6b = float((C2.double() - ref).abs().median())
    t6b = timeit(lambda: libs["two_acc"].bf16x6_gemm(stream, xa.data_ptr(), wb.data_ptr(), C2.data_ptr(), M, N, K))
    medb = float(((x @ w.t()).double() - ref).abs().median())
    sg = torch.sign(ref)
    bias6 = float(((Cm.double() - ref) * sg).mean()); bias6b = float(((C2.double() - ref) * sg).mean())
    biasb = float((((x @ w.t()).double() - ref) * sg).mean())
    errb = float(((x @ w.t()).double() - ref).abs().max()) / scale
    # the engine's exact-f32 MFMA GEMM: down-pass layout (W [V=N][H=K], h [B=M][H=K])
    Wd = eng.alloc_matrix(N, K); Wd.copy_(w)
    vb = eng.alloc_vector(N)
    eng.set_option("fused_epilogue", 0)
    pre = eng.propdown(x, Wd, vb, gauss=True)[0]
    erre = float((pre[:, :N].double() - ref).abs().max()) / scale
    mede = float((pre[:, :N].double() - ref).abs().median())
    biase = float(((pre[:, :N].double() - ref) * sg).mean())
    t6 = timeit(run)
    tb = timeit(lambda: torch.matmul(x, w.t()))
    eng.kernel_timing(True)
    for _ in range(10): eng.propdown(x, Wd, vb, gauss=True)
    torch.cuda.synchronize(); n, ms = eng.kernel_timing_read(); eng.kernel_timing(False)
    te = ms * 1e3 / max(n, 1) * (n / 10.0)        # GEMM launches only (HIP events), per call
    f = 2.0 * M * N * K
    row = {"M": M, "N": N, "K": K,
           "bf16x6_us": round(t6, 1), "bf16x6_tflops_f32eq": round(f / t6 / 1e6, 1), "bf16x6_relerr": err6,
           "bf16x6_median_abs_err": med6, "two_acc_us": round(t6b, 1), "two_acc_relerr": err6b, "two_acc_median_abs_err": med6b,
           "rocblas_median_abs_err": medb, "engine_median_abs_err": mede,
           "signed_bias_toward_zero": {"bf16x6": bias6, "two_acc": bias6b, "rocblas": biasb, "engine": biase},
           "rocblas_f32_us": round(tb, 1), "rocblas_tflops": round(f / tb / 1e6, 1), "rocblas_relerr": errb,
           "engine_f32_mfma_us": round(te, 1), "engine_tflops": round(f / te / 1e6, 1), "engine_relerr": erre}
    res.append(row); print(json.dumps(row), flush=True)
json.dump(res, open(os.path.join(out, "bf16x6_gemm.json"), "w"), indent=1)
