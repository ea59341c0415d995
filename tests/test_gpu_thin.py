"""The thin-batch CD-k step (csrc/mdbn_thin.hip: minibatches of <= 32 rows, the reference's batch_size = 20 of MDBN.py:46 /
AMLsm2.py:245, on layers that are not LDS-resident) against the float64 oracle, teacher-forced along the device's own chain,
and beside the register-streaming path it replaces.  Reference arithmetic: rbm.py:303 (positive phase), :242-248 / :662-671
(gibbs_hvh), :392-419 (statistics), :347-365 (update)."""
import numpy as np
import pytest
import torch

from oracle import rbm_np
from oracle.philox_np import PhiloxDraws
from _margins import check

pytestmark = pytest.mark.gpu

SHAPES = [  # V, H, B, k, gauss, through an index list
    (784, 500, 20, 1, False, True),       # BASELINE configs[0]: MNIST RBM at the reference's batch size
    (19937, 400, 20, 1, True, True),      # the real gene-expression layer (AMLsm2.py:242-251)
    (784, 500, 20, 1, True, False),
    (1000, 300, 10, 3, False, True),      # RBM.training's default batch (rbm.py:484-491), CD-3
    (600, 96, 32, 2, True, True),         # a full 32-row tile
    (2000, 512, 7, 1, False, True),       # ragged batch (two Philox blocks, the second one partial), widest hidden layer
    (300, 130, 20, 2, False, False),      # fewer workgroups than CUs, hidden width not a multiple of 32
    (4099, 72, 25, 1, True, True),        # prime number of visible units: ragged last sub-block and last workgroup
    (17, 5, 3, 2, False, True),           # tiny everything: one workgroup, one Philox block, H < one tile
    (700, 512, 32, 1, True, False),       # the widest hidden layer at the largest batch: the LDS budget cuts the row blocks
    (5000, 8, 1, 1, False, True),         # a single row
]


def _step(eng, V, H, B, k, gauss, indexed, seed, thin):
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
    eng.set_option("small_fused", 0)                    # (an LDS-resident shape would take the one-launch path first)
    eng.set_option("thin_fused", int(thin))
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
        eng.set_option("thin_fused", 1)
    x = data[idx] if idx is not None else data
    return dict(W=W, hb=hb, vb=vb, x=x, stats=stats.cpu().numpy(), sc=sc, th=th, tv=tv, n_gemm=n_gemm)


@pytest.mark.parametrize("V,H,B,k,gauss,indexed", SHAPES)
def test_thin_step_against_forced_oracle(hip_engine, V, H, B, k, gauss, indexed):
    r = _step(hip_engine, V, H, B, k, gauss, indexed, seed=V + H + k, thin=True)
    assert r["n_gemm"] == 0, "the step went through %d GEMM launches: not the thin path" % r["n_gemm"]
    st = rbm_np.RBMState(V, H, W=r["W"], hbias=r["hb"], vbias=r["vb"], gauss=gauss)
    v0 = r["x"].astype(np.float64)
    ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(5, 3, 11, 0), k, r["th"], r["tv"])
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
    sc = r["sc"]
    ldh, ldv = sc.P2.stride(0), sc.V2.stride(0)
    d = r["stats"]
    S, s_h, s_v = d[:V * ldh].reshape(V, ldh), d[V * ldh:V * ldh + H], d[V * ldh + ldh:V * ldh + ldh + V]
    cost = d[V * ldh + ldh + ldv]
    tag = "thin CD-%d %d->%d B=%d %s" % (k, V, H, B, "GRBM" if gauss else "RBM")
    assert not S[:, H:].any(), "pad columns of S must stay zero"
    check(tag + ": S / max|S|", np.abs(S[:, :H] - S_o).max() / max(1.0, np.abs(S_o).max()), 1e-5, "stats")
    check(tag + ": s_h / max", np.abs(s_h - s_h_o).max() / max(1.0, np.abs(s_h_o).max()), 1e-5, "stats")
    check(tag + ": s_v / max", np.abs(s_v - s_v_o).max() / max(1.0, np.abs(s_v_o).max()), 1e-5, "stats")
    check(tag + ": ph_mean", np.abs(sc.P2[:B].cpu().numpy()[:, :H] - ph).max(), 2e-6, "prob")
    check(tag + ": nh_mean", np.abs(-sc.P2[B:2 * B].cpu().numpy()[:, :H] - out[4]).max(), 4e-6, "prob")
    check(tag + ": nv_mean / max|nv|", np.abs(sc.V2[B:2 * B].cpu().numpy()[:, :V] - out[1]).max() / max(1.0, np.abs(out[1]).max()),
          2e-6, "nv_mean")
    np.testing.assert_array_equal(sc.V2[:B].cpu().numpy()[:, :V], r["x"])                    # the gathered rows
    assert not sc.P2.cpu().numpy()[:, H:].any() and not sc.V2.cpu().numpy()[:, V:].any(), "pad columns must stay zero"
    pre = out[0]
    if gauss:
        want = ((rbm_np.sigmoid(pre) - v0) ** 2).sum()
    else:
        want = (v0 * rbm_np.softplus(-pre) + (1 - v0) * rbm_np.softplus(pre)).sum()
    check(tag + ": cost sum rel", abs(cost - want) / abs(want), 2e-6)
    assert flips <= 3


@pytest.mark.parametrize("V,H,B,k,gauss,indexed", [SHAPES[0], SHAPES[1], SHAPES[3]])
def test_thin_step_draws_the_uniforms_of_the_streaming_path(hip_engine, V, H, B, k, gauss, indexed):
    """Same Philox addressing as the register-streaming GEMM path (thin_fused = 0): the positive-phase samples agree except
    within rounding of a uniform, the statistics to fp32 summation order."""
    a = _step(hip_engine, V, H, B, k, gauss, indexed, seed=3, thin=True)
    b = _step(hip_engine, V, H, B, k, gauss, indexed, seed=3, thin=False)
    assert a["n_gemm"] == 0 and b["n_gemm"] > 0
    differ = int((a["th"][0] != b["th"][0]).sum())
    assert differ <= 2, differ
    if differ == 0 and k == 1:
        scale = max(1.0, np.abs(b["stats"]).max())
        assert np.abs(a["stats"] - b["stats"]).max() <= 2e-5 * scale


@pytest.mark.parametrize("cls_gauss,hp", [(False, dict(lr=0.1, weightcost=2e-4)), (True, dict(lr=0.005, lambda_2=0.1)),
                                          (True, dict(lr=0.002, lambda_1=0.01, lambda_2=0.05))],
                         ids=["rbm_weightcost", "grbm_l2", "grbm_l1_l2"])
@pytest.mark.parametrize("fused_update", [1, 0])
def test_thin_training_steps_follow_the_oracle(hip_engine, cls_gauss, hp, fused_update):
    """Four calls of the compiled step function (rbm.py:258-376) through the RBM classes at batch 20: parameters, speeds and
    costs against the float64 oracle that follows the device's recorded chain -- the update consumed row by row inside
    thin_update_kernel (default) and as statistics + update_kernel (fused_update = 0)."""
    import mdbn_amd
    eng = hip_engine
    V, H, B, N = 1200, 340, 20, 128
    rs = np.random.RandomState(7)
    data = rs.normal(size=(N, V)).astype(np.float32) if cls_gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    cls = mdbn_amd.GRBM if cls_gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(7), engine=eng)
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), gauss=cls_gauss)
    if hp.get("weightcost"):
        st.freeze_W0()
    eng.set_option("fused_update", fused_update)
    eng.trace_chain = True
    eng.kernel_timing(True)
    try:
        _, updates = rbm.get_cost_updates(k=2, batch_size=B, **hp)
        fn = mdbn_amd.function(updates, mdbn_amd.shared(data, engine=eng), data_parallel=None)
        for t in range(4):
            idx = rs.permutation(N)[:B]
            mom = 0.5 if t < 2 else 0.9
            cost = float(fn(indexes=idx, momentum=mom))
            sc = eng.last_scratch
            forced = (sc.trace_h.cpu().numpy()[:, :, :H], None if cls_gauss else sc.trace_v.cpu().numpy()[:, :, :V])
            want = rbm_np.cd_step(st, data[idx], PhiloxDraws(7, rbm.stream_id, t), k=2, batch_size=B, momentum=mom,
                                  forced=forced, **hp)
            check("thin training: cost rel", abs(cost - want) / abs(want), 1e-5)
        eng.synchronize()
        assert len(eng.kernel_timing_detail()) == 0, "not the thin path"
    finally:
        eng.kernel_timing(False)
        eng.trace_chain = False
        eng.set_option("fused_update", 1)
    for name in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed"):
        got, ref = getattr(rbm, name).get_value(), getattr(st, name)
        check("thin training: %s after 4 steps / max" % name, np.abs(got - ref).max() / max(1.0, np.abs(ref).max()), 2e-6, "update")


AHEAD = [  # V, H, B, gauss, hyper-parameters
    (784, 500, 20, False, dict(lr=0.1, weightcost=2e-4)),          # BASELINE configs[0], frozen weight-cost snapshot
    (19937, 400, 20, True, dict(lr=0.005, lambda_2=0.1)),          # the gene-expression layer
    (700, 512, 32, True, dict(lr=0.002, lambda_1=0.01, lambda_2=0.05)),   # widest layer at the largest batch (4 rows in flight)
    (4099, 72, 25, False, dict(lr=0.05)),                          # ragged everything
    (17, 5, 3, False, dict(lr=0.1)),                               # one workgroup
]


@pytest.mark.parametrize("V,H,B,gauss,hp", AHEAD)
def test_thin_positive_phase_ahead_is_bit_identical(hip_engine, V, H, B, gauss, hp):
    """mdbn_cd_args.next_indexes on the thin path: the update kernel of step t also gathers minibatch t + 1 and leaves the
    partials of its positive phase (thin_update_ahead_kernel).  Same products in the same order as thin_pass_kernel<0>:
    parameters, speeds and costs must be BIT-identical with and without the hint -- including after a wrong hint, a
    parameter written from outside between two steps, and a step without a hint in the middle."""
    import mdbn_amd
    eng = hip_engine
    N = 96
    rs = np.random.RandomState(11)
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    order = [rs.permutation(N)[:B] for _ in range(9)]

    def run(hinted):
        rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5), theano_rng=mdbn_amd.RandomStreams(3), engine=eng)
        _, updates = rbm.get_cost_updates(k=1, batch_size=B, **hp)
        fn = mdbn_amd.function(updates, mdbn_amd.shared(data, engine=eng), data_parallel=None)
        idx = [eng.index_tensor(o, N) for o in order]
        costs, taken = [], 0
        for t in range(8):
            hint = None
            if hinted and t != 4:                        # step 4 announces nothing: step 5 gathers for itself
                hint = idx[t + 1] if t != 2 else idx[0]  # step 2 announces the WRONG minibatch: step 3 must not use it
            if t == 6:
                w = rbm.W.get_value()
                rbm.W.set_value(w * np.float32(0.5))     # parameters written behind the step function's back
            costs.append(fn(indexes=idx[t], momentum=0.5, next_indexes=hint))
            sc = eng.last_scratch
            taken += int(sc.ahead is not None)
        eng.synchronize()
        out = {n: getattr(rbm, n).get_value() for n in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed")}
        out["cost"] = np.array([float(c) for c in costs])
        return out, taken

    eng.keep_f32 = False                                 # the product default (the fixture keeps the f32 copies for inspection)
    eng.set_option("small_fused", 0)                     # (an LDS-resident shape would take the one-launch path first)
    try:
        plain, n0 = run(False)
        ahead, n1 = run(True)
    finally:
        eng.keep_f32 = True
        eng.set_option("small_fused", 1)
    assert n0 == 0
    assert n1 == 7, "the update kernel prepared %d of the 7 announced minibatches" % n1
    for name in plain:
        np.testing.assert_array_equal(ahead[name], plain[name], err_msg=name)
