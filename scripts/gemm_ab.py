"""A/B of GEMM slice depth (gemm_bk 32 vs 64) interleaved in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B = 4096, 1024, 512
W = eng.alloc_matrix(V, H); W.normal_(0, 0.05)
hb = eng.alloc_vector(H); vb = eng.alloc_vector(V)
x = eng.alloc_matrix(B, V); x.normal_(); h = eng.alloc_matrix(B, H); h.uniform_()
V2 = eng.alloc_matrix(2 * B, V); V2.normal_(); P2 = eng.alloc_matrix(2 * B, H); P2.uniform_()
stats = eng.stats_buffer(V, H); ws = eng.workspace(B, V, H)
import ctypes as C
def stats_call():
    mdbn_amd._lib.check(eng.lib.mdbn_cd_stats(eng.ctx, eng._stream(), eng._p(V2), eng._p(P2), B, V, H, V2.stride(0), P2.stride(0), eng._p(stats), eng._p(ws), ws.numel() * 4), "cd_stats")
calls = {"up": lambda: eng.propup(x, W, hb, want_pre=False, want_sample=False),
         "down": lambda: eng.propdown(h, W, vb, gauss=True), "stats": stats_call}
flop = {"up": 2.0 * B * V * H, "down": 2.0 * B * V * H, "stats": 4.0 * B * V * H}
res = {}
for rnd in range(6):
    for bk in (32, 64):
        eng.set_option("gemm_bk", bk)
        for name, fn in calls.items():
            fn(); eng.synchronize(); eng.kernel_timing(True)
            for _ in range(10): fn()
            eng.synchronize(); n, ms = eng.kernel_timing_read(); eng.kernel_timing(False)
            res.setdefault((name, bk), []).append(ms * 1e3 / n)
for name in calls:
    for bk in (32, 64):
        v = np.array(res[(name, bk)])
        print("%-6s bk=%d  median %.1f us  min %.1f  (%.1f TF at median)" % (name, bk, np.median(v), v.min(), flop[name] / np.median(v) / 1e6))
