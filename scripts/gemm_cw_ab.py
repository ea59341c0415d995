"""Tiled GEMM: one vs two MFMA (consumer) waves per SIMD, GEMM-only HIP-event timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
eng.set_option("fused_epilogue", 0)
V, H = 4096, 1024
W = eng.alloc_matrix(V, H); W.normal_(0, 0.05)
hb = eng.alloc_vector(H); vb = eng.alloc_vector(V)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    eng.synchronize(); eng.kernel_timing(True)
    for _ in range(reps): fn()
    eng.synchronize(); n, ms = eng.kernel_timing_read(); eng.kernel_timing(False)
    return ms * 1e3 / n
ref = {}
for B in (128, 256, 512, 1024, 2048, 4096):
    x = eng.alloc_matrix(B, V); x.normal_()
    h = eng.alloc_matrix(B, H); h.uniform_()
    for rep in range(3):
        for cw in (1, 2):
            eng.set_option("gemm_cw", cw)
            up = timeit(lambda: eng.propup(x, W, hb, want_pre=False, want_sample=False))
            dn = timeit(lambda: eng.propdown(h, W, vb, gauss=True))
            out = eng.propup(x, W, hb, want_pre=True, want_sample=False)[0].cpu().numpy()
            if cw == 1: ref[B] = out
            err = float(np.abs(out - ref[B]).max())
            f = 2.0 * B * V * H
            print("B=%5d cw=%d  up %7.1f us %5.1f TF (%2.0f%%)   down %7.1f us %5.1f TF (%2.0f%%)  max|diff vs cw1| %.2e" %
                  (B, cw, up, f / up / 1e6, 100 * f / up / 1e6 / 157.3, dn, f / dn / 1e6, 100 * f / dn / 1e6 / 157.3, err), flush=True)
