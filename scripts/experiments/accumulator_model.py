#!/usr/bin/env python
"""Numpy model of the plane GEMM's accumulation (32-deep stages, six bf16 piece products per stage, f32 accumulator, eight
split-K slabs at K = 4096): which rounding dominates the error of a probability -- the slab sum (f32 / float64) or the
accumulator (six roundings per stage at its own magnitude / one with a stage-local accumulator).  DESIGN.md section 3,
"Stage-local accumulators".   python scripts/experiments/accumulator_model.py"""
import numpy as np
rs=np.random.RandomState(0)
K=4096; M=64; N=256
x=rs.normal(size=(M,K)).astype(np.float32); W=rs.uniform(-0.137,0.137,size=(K,N)).astype(np.float32)
def split3(a):
    b=a.view(np.uint32); p1=(b&0xffff0000).view(np.float32); r=a-p1
    p2=(r.view(np.uint32)&0xffff0000).view(np.float32); r2=r-p2
    p3=(r2.view(np.uint32)&0xffff0000).view(np.float32); return p1,p2,p3
xs=split3(x); ws=split3(W)
pairs=[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]
ref=x.astype(np.float64)@W.astype(np.float64)
def run(mode, nsl=8, slabsum='f32'):
    slabs=[]
    ks=K//nsl
    for s in range(nsl):
        acc=np.zeros((M,N),np.float32)
        for k0 in range(s*ks,(s+1)*ks,32):
            sl=slice(k0,k0+32)
            if mode=='six':
                for (i,j) in pairs:
                    acc=(acc.astype(np.float64)+xs[i][:,sl].astype(np.float64)@ws[j][sl].astype(np.float64)).astype(np.float32)
            elif mode=='six_small_first':
                for (i,j) in pairs[::-1]:
                    acc=(acc.astype(np.float64)+xs[i][:,sl].astype(np.float64)@ws[j][sl].astype(np.float64)).astype(np.float32)
            elif mode=='temp':
                t=np.zeros((M,N),np.float32)
                for (i,j) in pairs[::-1]:
                    t=(t.astype(np.float64)+xs[i][:,sl].astype(np.float64)@ws[j][sl].astype(np.float64)).astype(np.float32)
                acc=acc+t
            elif mode=='twoacc':
                pass
        slabs.append(acc)
    if slabsum=='f32':
        tot=np.zeros((M,N),np.float32)
        for a in slabs: tot=tot+a
    else:
        tot=sum(a.astype(np.float64) for a in slabs).astype(np.float32)
    err=np.abs(tot-ref)
    p=1/(1+np.exp(-tot.astype(np.float64))); pr=1/(1+np.exp(-ref))
    return err.max(), np.abs(p-pr).max()
for mode in ['six','six_small_first','temp']:
    for ss in ['f32','f64']:
        print(mode, ss, run(mode,8,ss))
