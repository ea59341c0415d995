"""A few GEMM launches of fixed shapes, for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H = 4096, 1024
W = eng.alloc_matrix(V, H); W.normal_(0, 0.05)
hb = eng.alloc_vector(H)
for B in (512, 16384):
    x = eng.alloc_matrix(B, V); x.normal_()
    for _ in range(5):
        eng.propup(x, W, hb, want_pre=False, want_sample=False)
    eng.synchronize()
