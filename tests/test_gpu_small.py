"""The one-launch CD-k step of LDS-resident layers (csrc/mdbn_small.hip; reference: the scan of rbm.py:318-336 at the
layer sizes of MDBN.py:45-52 / AMLsm2.py:242-340) against the float64 oracle, teacher-forced along the device's own chain,
and beside the multi-launch path it replaces."""
import numpy as np
import pytest
import torch

from oracle import rbm_np
from oracle.philox_np import PhiloxDraws
from _margins import check

pytestmark = pytest.mark.gpu

SHAPES = [  # V, H, B, k, gauss, through an index list
    (512, 40, 512, 5, True, True),       # c5 miRNA
    (400, 40, 512, 1, False, True),      # c5 GE second layer
    (200, 20, 512, 1, False, False),     # c5 SM second layer
    (100, 128, 512, 1, False, True),     # c5 joint layer
    (100, 24, 20, 1, False, True),       # the reference's own joint DBN (MDBN.py:31-35) at its batch size
    (24, 3, 20, 2, False, False),
    (512, 40, 37, 2, True, True),        # ragged last slab
    (130, 70, 100, 3, False, True),      # no dimension a multiple of 16
    (300, 60, 1100, 1, True, False),     # more slabs than workgroups: a workgroup loops
]


def _step(eng, V, H, B, k, gauss, indexed, seed, fused):
    from mdbn_amd import RngAddr
    rs = np.random.RandomState(seed)
    W = rbm_np.init_W(rs, V, H, np.float32)
    hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
    N = B + 13 if indexed else B
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    idx = None
    if indexed:
        idx = rs.permutation(N)[:B].astype(np.int64)
        idx[::7] -= N                                   # numpy-style negative indices
    dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, hb, vb, data)]
    eng.set_option("small_fused", int(fused))
    eng.trace_chain = True
    eng.kernel_timing(True)
    try:
        stats, sc = eng.cd_step(dx, idx, dW, dhb, dvb, gauss, k, RngAddr(5, 3, 11, 0, 0))
        eng.synchronize()
        n_gemm = len(eng.kernel_timing_detail())
        th = sc.trace_h.cpu().numpy()[:, :, :H]
        tv = None if gauss else sc.trace_v.cpu().numpy()[:, :, :V]
    finally:
        eng.kernel_timing(False)
        eng.trace_chain = False
        eng.set_option("small_fused", 1)
    x = data[idx] if idx is not None else data
    return dict(W=W, hb=hb, vb=vb, x=x, stats=stats.cpu().numpy(), sc=sc, th=th, tv=tv, n_gemm=n_gemm)


@pytest.mark.parametrize("V,H,B,k,gauss,indexed", SHAPES)
def test_one_launch_step_against_forced_oracle(hip_engine, V, H, B, k, gauss, indexed):
    r = _step(hip_engine, V, H, B, k, gauss, indexed, seed=V + H + k, fused=True)
    assert r["n_gemm"] == 0, "the step went through %d GEMM launches: not the one-launch path" % r["n_gemm"]
    st = rbm_np.RBMState(V, H, W=r["W"], hbias=r["hb"], vbias=r["vb"], gauss=gauss)
    v0 = r["x"].astype(np.float64)
    ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(5, 3, 11, 0), k, r["th"], r["tv"])
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
    sc = r["sc"]
    ldh, ldv = sc.P2.stride(0), sc.V2.stride(0)
    d = r["stats"]
    S, s_h, s_v = d[:V * ldh].reshape(V, ldh), d[V * ldh:V * ldh + H], d[V * ldh + ldh:V * ldh + ldh + V]
    cost = d[V * ldh + ldh + ldv]
    tag = "one-launch CD-%d %d->%d B=%d %s" % (k, V, H, B, "GRBM" if gauss else "RBM")
    assert not S[:, H:].any(), "pad columns of S must stay zero"
    check(tag + ": S / max|S|", np.abs(S[:, :H] - S_o).max() / max(1.0, np.abs(S_o).max()), 1e-5, "stats")
    check(tag + ": s_h / max", np.abs(s_h - s_h_o).max() / max(1.0, np.abs(s_h_o).max()), 1e-5, "stats")
    check(tag + ": s_v / max", np.abs(s_v - s_v_o).max() / max(1.0, np.abs(s_v_o).max()), 1e-5, "stats")
    check(tag + ": ph_mean", np.abs(sc.P2[:B].cpu().numpy()[:, :H] - ph).max(), 2e-6, "prob")
    check(tag + ": nh_mean", np.abs(-sc.P2[B:2 * B].cpu().numpy()[:, :H] - out[4]).max(), 4e-6, "prob")
    check(tag + ": nv_mean / max|nv|", np.abs(sc.V2[B:2 * B].cpu().numpy()[:, :V] - out[1]).max() / max(1.0, np.abs(out[1]).max()),
          2e-6, "nv_mean")
    np.testing.assert_array_equal(sc.V2[:B].cpu().numpy()[:, :V], r["x"])                    # the gathered rows
    # monitoring cost (sum form, before the step function's scale): rbm.py:479-480 / :697 on the last pre-sigmoid activation
    pre = out[0]
    if gauss:
        want = ((rbm_np.sigmoid(pre) - v0) ** 2).sum()
    else:
        want = (v0 * rbm_np.softplus(-pre) + (1 - v0) * rbm_np.softplus(pre)).sum()
    check(tag + ": cost sum rel", abs(cost - want) / abs(want), 2e-6)
    assert flips <= 3


@pytest.mark.parametrize("V,H,B,k,gauss,indexed", SHAPES[:4])
def test_one_launch_step_draws_the_uniforms_of_the_multi_launch_path(hip_engine, V, H, B, k, gauss, indexed):
    """Same Philox addressing: the positive-phase samples of the two paths agree except where a probability lies within
    rounding of its uniform, and the statistics agree to fp32 summation order (teacher-forced statistics are compared with the
    oracle above; here the two device paths are laid side by side)."""
    a = _step(hip_engine, V, H, B, k, gauss, indexed, seed=3, fused=True)
    b = _step(hip_engine, V, H, B, k, gauss, indexed, seed=3, fused=False)
    assert a["n_gemm"] == 0 and b["n_gemm"] > 0
    differ = int((a["th"][0] != b["th"][0]).sum())
    assert differ <= 2, differ                                  # h0 ~ Bernoulli(ph): the same uniforms, the same ph to ~1e-7
    if differ == 0 and k == 1:
        scale = max(1.0, np.abs(b["stats"]).max())
        assert np.abs(a["stats"] - b["stats"]).max() <= 2e-5 * scale


def test_cached_argument_structs_give_the_same_training_run(hip_engine):
    """Small layers are host-bound: a step function reuses the argument structs of its previous call (only the index list, the
    Philox step, lr / momentum and the cost slot change).  Twelve steps with changing index tensors, momentum and lr through the
    cached path equal the same steps with the cache switched off, bit for bit -- and the cache really was used."""
    import mdbn_amd
    eng = mdbn_amd.HipEngine()                                   # product defaults (no inspection copies, no taps)
    runs, used = [], []
    for cached in (True, False):
        rs = np.random.RandomState(4)
        data = (rs.uniform(size=(400, 100)) < 0.3).astype(np.float32)
        rbm = mdbn_amd.RBM(n_visible=100, n_hidden=24, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(9),
                           engine=eng)
        _, up = rbm.get_cost_updates(k=1, batch_size=20, lr=0.05, weightcost=2e-4)
        fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel=None)
        batches = [eng.index_tensor(rs.permutation(400)[:20]) for _ in range(12)]
        costs, hits = [], 0
        for t in range(12):
            if not cached:
                del fn._fast[:]
            hits += int(len(fn._fast) == 8)
            costs.append(float(fn(indexes=batches[t], momentum=0.5 if t < 6 else 0.9, lr=0.05 if t % 2 else 0.02)))
        used.append(hits)
        runs.append(dict(costs=np.array(costs), W=rbm.W.get_value(), Ws=rbm.W_speed.get_value(), hb=rbm.hbias.get_value(),
                         vbs=rbm.vbias_speed.get_value(), n=np.array([rbm._n_updates, rbm._rng_step])))
    assert used == [11, 0], used
    for key in runs[0]:
        assert np.array_equal(runs[0][key], runs[1][key]), key
