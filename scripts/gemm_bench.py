"""GEMM-only timing (HIP events via the library's timing hook) for a few shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H = 4096, 1024
W = eng.alloc_matrix(V, H); W.normal_(0, 0.05)
hb = eng.alloc_vector(H); vb = eng.alloc_vector(V)
def timeit(label, fn, flop, reps=20):
    for _ in range(3): fn()
    eng.synchronize(); eng.kernel_timing(True)
    for _ in range(reps): fn()
    eng.synchronize(); n, ms = eng.kernel_timing_read(); eng.kernel_timing(False)
    us = ms * 1e3 / n
    print("%-28s launches/call=%d  %8.1f us  %6.1f TF (%.0f%% of 157.3)" % (label, n // reps, us, flop / us / 1e6, 100 * flop / us / 1e6 / 157.3), flush=True)
for B in (512, 2048, 16384):
    x = eng.alloc_matrix(B, V); x.normal_()
    h = eng.alloc_matrix(B, H); h.uniform_()
    timeit("up   B=%d" % B, lambda: eng.propup(x, W, hb, want_pre=False, want_sample=False), 2.0 * B * V * H)
    timeit("down B=%d" % B, lambda: eng.propdown(h, W, vb, gauss=True), 2.0 * B * V * H)
