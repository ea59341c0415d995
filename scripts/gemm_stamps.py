"""Phase timeline of one GEMM block from the -DMDBN_STAMP diagnostic build."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdbn_amd import _lib
_lib.use_diagnostic_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMP_LIB", "libmdbn_stamp.so")))
import mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
lib = eng.lib
V, H = 4096, 1024
W = eng.alloc_matrix(V, H); W.normal_(0, 0.05)
hb = eng.alloc_vector(H)
for B in (512, 16384):
    x = eng.alloc_matrix(B, V); x.normal_()
    stamps = torch.zeros(12 * 64 * 8, dtype=torch.int64, device=eng.device)
    for _ in range(3):
        eng.propup(x, W, hb, want_pre=False, want_sample=False)
    lib.mdbn_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    eng.propup(x, W, hb, want_pre=False, want_sample=False)
    eng.synchronize()
    lib.mdbn_debug_set_stamps(C.c_void_p(0))
    st = stamps.cpu().numpy().reshape(12, 64, 8)
    nt = int((st[0, :, 0] > 0).sum())
    t0 = st[:, 0, 0].min()
    print("B=%d: %d slices stamped" % (B, nt))
    for w in range(8):
        s_ = st[w, :nt].astype(np.int64)
        if w < 4:   # consumer: 0 = slice start, 1 = MFMAs issued, 2 = past barrier
            d = np.stack([s_[:, 1] - s_[:, 0], s_[:, 2] - s_[:, 1]], 1)
            names = "compute %5.0f  barrier %5.0f"
        else:       # producer: 0 = start, 1 = LDS stores issued, 2 = loads issued, 3 = past barrier
            d = np.stack([s_[:, 1] - s_[:, 0], s_[:, 2] - s_[:, 1], s_[:, 3] - s_[:, 2]], 1)
            names = "lds_store %5.0f  issue_loads %5.0f  barrier %5.0f"
        tot = s_[1:, 0] - s_[:-1, 0]
        print(" wave %d first-stamp +%6d | per-slice cycles: " % (w, s_[0, 0] - t0) + names % tuple(d[1:-1].mean(0)) +
              " | slice total %.0f (min %.0f max %.0f) | whole loop %d" % (tot.mean(), tot.min(), tot.max(), s_[-1, 2 if w < 4 else 3] - s_[0, 0]))
