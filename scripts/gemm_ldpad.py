"""Does a non-power-of-two leading dimension speed the GEMM loads up? (L2 channel test)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B = 4096, 1024, 512
def mat(rows, cols, pad):
    t = torch.zeros((rows, cols + pad), dtype=torch.float32, device=eng.device)[:, :cols]
    t.normal_(0, 0.05)
    return t
def timeit(label, fn, flop, reps=20):
    for _ in range(3): fn()
    eng.synchronize(); eng.kernel_timing(True)
    for _ in range(reps): fn()
    eng.synchronize(); n, ms = eng.kernel_timing_read(); eng.kernel_timing(False)
    us = ms * 1e3 / n
    print("%-34s %8.1f us  %6.1f TF" % (label, us, flop / us / 1e6), flush=True)
hb = eng.alloc_vector(H); vb = eng.alloc_vector(V)
for padv, padh in ((0, 0), (32, 0), (0, 32), (32, 32), (16, 16), (64, 64), (4, 4)):
    W = mat(V, H, padh); x = mat(B, V, padv); h = mat(B, H, padh)
    timeit("up   padV=%d padH=%d" % (padv, padh), lambda: eng.propup(x, W, hb, want_pre=False, want_sample=False), 2.0 * B * V * H)
    timeit("down padV=%d padH=%d" % (padv, padh), lambda: eng.propdown(h, W, vb, gauss=True), 2.0 * B * V * H)
