"""The reference's class surface ON THE HIP ENGINE, checked step by step against the float64 oracle with
teacher forcing (tests/_shadow.py): eager gibbs_hvh / gibbs_vhv and their RNG accounting, the stand-alone
trainers RBM.training (CD and PCD) / GRBM.training / learn_model, DBN.training, MDBN.train_bottom_layer /
train_top, CD-5 at the c5 miRNA shape, and 100-step weight drift at the shapes SURVEY 8d names."""
import numpy as np
import pytest
import torch

from oracle import philox_np, rbm_np
from oracle.philox_np import PhiloxDraws
from _margins import check

pytestmark = pytest.mark.gpu


@pytest.fixture()
def shadow(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdbn_amd
    import mdbn_amd.engine as E
    from _shadow import ShadowEngine
    prev = E._default_engine
    eng = mdbn_amd.set_engine(ShadowEngine())
    rbm_np.FLIP_GAP["max"] = 0.0
    yield eng
    E._default_engine = prev


def _verdict(shadow, name, rbms=(), cost_tol=None, param_tol=None, stat_tol=None):
    """Record the shadow's worst deviations under their tolerances (tests/_margins.py)."""
    if cost_tol is not None:
        check(name + ": step cost rel", shadow.cost_err, cost_tol)
    if stat_tol is not None:
        check(name + ": S / s_h / s_v rel-to-max", shadow.stat_err, stat_tol, "stats")
    for r in rbms:
        check(name + ": parameters vs shadow, rel-to-max", shadow.param_err(r), param_tol)
    check(name + ": |u - p| of a flipped draw", rbm_np.FLIP_GAP["max"], shadow.tie, "tie")


def _u(rbm, step, rows, cols):
    return philox_np.uniform(rows, cols, rbm.theano_rng.seed, rbm.stream_id, step, 0, 0)


@pytest.mark.parametrize("gauss", [False, True])
@pytest.mark.parametrize("V,H,B", [(12, 7, 5), (784, 500, 20), (1024, 256, 512)])
def test_eager_gibbs_steps_and_rng_accounting(hip_engine, gauss, V, H, B):
    """gibbs_hvh / gibbs_vhv (rbm.py:242-256; GRBM :662-682) as eager device calls: the six outputs in
    the documented order against the oracle, and the Philox bookkeeping: every sample_* call takes one
    ``step`` (draw 0), so one Gibbs step advances the layer's counter by exactly two."""
    import mdbn_amd
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(V + H), theano_rng=mdbn_amd.RandomStreams(31),
              engine=hip_engine)
    rs = np.random.RandomState(1)
    rbm.hbias.set_value(rs.normal(0, 0.2, H).astype(np.float32))
    rbm.vbias.set_value(rs.normal(0, 0.2, V).astype(np.float32))
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), hbias=rbm.hbias.get_value(), vbias=rbm.vbias.get_value(), gauss=gauss)
    tol_h = 2e-6 * max(1.0, (V / 1024.0) ** 0.5)
    tol_v = 4e-6 * max(1.0, (H / 1024.0) ** 0.5)

    def same_sample(got, mean, u):
        want = (u < mean).astype(np.float64)
        bad = got != want
        assert np.all(np.abs(u - mean)[bad] < 1e-6)

    # --- gibbs_hvh from a binary hidden state
    h0 = (rs.uniform(size=(B, H)) < 0.5).astype(np.float32)
    s0 = rbm._rng_step
    pre_v, v_mean, v_sample, pre_h, h_mean, h_sample = [a.get_value() for a in rbm.gibbs_hvh(h0)]
    assert rbm._rng_step == s0 + 2
    o_pre_v, o_v_mean, _ = rbm_np.sample_v_given_h(st, h0.astype(np.float64), _u(rbm, s0, B, V))
    scale = max(1.0, np.abs(o_v_mean).max())
    assert np.abs(v_mean - o_v_mean).max() <= tol_v * scale and np.abs(pre_v - o_pre_v).max() <= 2 * tol_v * max(1.0, np.abs(o_pre_v).max())
    if gauss:
        assert np.array_equal(v_sample, v_mean)                 # error_free: the sample IS the mean (rbm.py:652-653)
        v_in = v_mean
    else:
        same_sample(v_sample, o_v_mean, _u(rbm, s0, B, V))
        v_in = v_sample
    o_pre_h, o_h_mean = rbm_np.propup(st, v_in.astype(np.float64))   # teacher-forced on the device's visible state
    assert np.abs(h_mean - o_h_mean).max() <= tol_h and np.abs(pre_h - o_pre_h).max() <= 1e-5 * max(1.0, np.abs(o_pre_h).max())
    same_sample(h_sample, o_h_mean, _u(rbm, s0 + 1, B, H))

    # --- gibbs_vhv from data
    v0 = rs.normal(size=(B, V)).astype(np.float32) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
    s1 = rbm._rng_step
    pre_h, h_mean, h_sample, pre_v, v_mean, v_sample = [a.get_value() for a in rbm.gibbs_vhv(v0)]
    assert rbm._rng_step == s1 + 2
    _, o_h_mean = rbm_np.propup(st, v0.astype(np.float64))
    assert np.abs(h_mean - o_h_mean).max() <= tol_h
    same_sample(h_sample, o_h_mean, _u(rbm, s1, B, H))
    h_in = h_mean if gauss else h_sample                       # GRBM: v1 from the hidden MEAN (rbm.py:680)
    _, o_v_mean, _ = rbm_np.sample_v_given_h(st, h_in.astype(np.float64), _u(rbm, s1 + 1, B, V))
    assert np.abs(v_mean - o_v_mean).max() <= tol_v * max(1.0, np.abs(o_v_mean).max())
    if not gauss:
        same_sample(v_sample, o_v_mean, _u(rbm, s1 + 1, B, V))


def test_noisy_grbm_sample(hip_engine):
    """GRBM(error_free=False): v1_sample = v1_mean + N(0,1) (rbm.py:655-658), Box-Muller twin."""
    import mdbn_amd
    V, H, B = 96, 40, 33
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(3), theano_rng=mdbn_amd.RandomStreams(8),
                        error_free=False, engine=hip_engine)
    h0 = (np.random.RandomState(0).uniform(size=(B, H)) < 0.5).astype(np.float32)
    s0 = rbm._rng_step
    _, mean, sample = [a.get_value() for a in rbm.sample_v_given_h(h0)]
    z = philox_np.normal(B, V, 8, rbm.stream_id, s0, 0, 0)
    np.testing.assert_allclose(sample - mean, z, atol=3e-5)


@pytest.mark.parametrize("persistent", [False, True])
def test_rbm_training_on_device(shadow, persistent):
    """RBM.training -> learn_model (rbm.py:484-629) on the HIP engine: CD-2 and PCD-2, momentum switch at
    epoch 6, every step replayed by the oracle along the device's chain."""
    import mdbn_amd
    V, H, N, B = 100, 60, 120, 20
    rs = np.random.RandomState(0)
    data = (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    rbm = mdbn_amd.RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5),
                       engine=shadow)
    np.random.seed(4)                                        # the reference shuffles with the global state (utils.py:62)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        history = rbm.training(data, data[:30], training_epochs=8, batch_size=B, learning_rate=0.1, k=2,
                               initial_momentum=0.6, final_momentum=0.9, weightcost=2e-4, persistent=persistent)
    assert len(history) == 8 and shadow.steps == 8 * (N // B)
    assert all(np.isfinite(c) and g is not None and np.isfinite(g) for c, g in history)
    if persistent:
        # pseudo-likelihood monitor (rbm.py:421-447): epoch means of the per-step costs, bit index rotating
        per_epoch = np.array(shadow.pl_costs).reshape(8, N // B).mean(axis=1)
        np.testing.assert_allclose([c for c, _ in history], per_epoch, rtol=2e-4)
        assert rbm.bit_i_idx == (8 * (N // B)) % V
        _verdict(shadow, "RBM.training PCD-2 (100->60)", [rbm], stat_tol=1e-5, param_tol=5e-6)
    else:
        _verdict(shadow, "RBM.training CD-2 (100->60)", [rbm], cost_tol=2e-6, param_tol=5e-6)


def test_grbm_training_on_device(shadow):
    """GRBM.training (rbm.py:701-728: always CD, lambda_2 forwarded) on the HIP engine."""
    import io, contextlib
    import mdbn_amd
    V, H, N, B = 130, 70, 96, 32
    rs = np.random.RandomState(1)
    data = rs.normal(size=(N, V)).astype(np.float32)
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(6),
                        engine=shadow)
    np.random.seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        history = rbm.training(data, data[:16], training_epochs=7, batch_size=B, learning_rate=0.005, k=1,
                               lambda_1=0.01, lambda_2=0.1, persistent=True)      # persistent is ignored (rbm.py:701-728)
    assert len(history) == 7 and shadow.steps == 7 * 3
    _verdict(shadow, "GRBM.training (130->70, 21 steps)", [rbm], cost_tol=2e-6, param_tol=5e-6)


def test_mdbn_glue_on_device(shadow):
    """MDBN.train_bottom_layer / train_top (MDBN.py:31-76) on the HIP engine: two modalities with a
    Gaussian first layer, concatenated, joint Bernoulli DBN 24 -> 3; every CD step oracle-checked."""
    import mdbn_amd
    from mdbn_amd import MDBN
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(0)
    N = 64
    rng = np.random.RandomState(123)
    np.random.seed(11)
    shadow.tie = 4e-6            # thousands of steps: fp32 parameters drift from the float64 shadow (measured: no flipped draw at all)
    outs, nets = [], []
    for width, sizes in ((96, [24, 8]), (40, [6])):
        x = rs.normal(size=(N, width)).astype(np.float32)
        net, out_t, out_v = MDBN.train_bottom_layer(x, x[:8], batch_size=16, k=1, layers_sizes=sizes,
                                                    pretraining_epochs=[6] * len(sizes),
                                                    pretrain_lr=[0.005] + [0.1] * (len(sizes) - 1), rng=rng)
        assert out_t.shape == (N, sizes[-1]) and out_v.shape == (8, sizes[-1])
        outs.append(out_t)
        nets.append(net)
    joint = np.concatenate(outs, axis=1)
    top = MDBN.train_top(16, False, joint, None, rng)
    # train_top's fixed schedule: 800 "epochs" budget compared with the iteration count -> stops after ~800 iterations
    assert top.number_of_nodes() == [14, 24, 3]
    assert shadow.steps > 100
    _verdict(shadow, "MDBN glue (thousands of steps)", [r for net in nets + [top] for r in net.rbm_layers],
             cost_tol=2e-6, param_tol=2e-5)
    assert top.get_output(joint).shape == (N, 3)


@pytest.mark.parametrize("V,H,B,k,gauss", [(512, 40, 512, 5, True), (512, 40, 512, 5, False), (130, 70, 37, 3, False),
                                           (2048, 400, 512, 2, True), (100, 128, 512, 5, False)])
def test_cd_k_chain_teacher_forced(hip_engine, V, H, B, k, gauss):
    """CD-k with k > 1 (c5's miRNA layer: 512 -> 40, CD-5 at B = 512): statistics and cost of one step with the
    oracle following the device's chain at every half-step."""
    from mdbn_amd import RngAddr
    rs = np.random.RandomState(V + k)
    W = rbm_np.init_W(rs, V, H, np.float32)
    hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
    x = rs.normal(size=(B, V)).astype(np.float32) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
    dW, dhb, dvb, dx = [hip_engine.to_device(a) for a in (W, hb, vb, x)]
    hip_engine.trace_chain = True
    try:
        stats, sc = hip_engine.cd_step(dx, None, dW, dhb, dvb, gauss, k, RngAddr(99, 2, 7, 0, 0))
        th = sc.trace_h.cpu().numpy()[:, :, :H]
        tv = None if gauss else sc.trace_v.cpu().numpy()[:, :, :V]
    finally:
        hip_engine.trace_chain = False
    st = rbm_np.RBMState(V, H, W=W, hbias=hb, vbias=vb, gauss=gauss)
    v0 = x.astype(np.float64)
    ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(99, 2, 7, 0), k, th, tv)
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
    ldh, ldv = sc.P2.stride(0), sc.V2.stride(0)
    d = stats.cpu().numpy()
    S, s_h, s_v = d[:V * ldh].reshape(V, ldh)[:, :H], d[V * ldh:V * ldh + H], d[V * ldh + ldh:V * ldh + ldh + V]
    tag = "CD-%d %d->%d %s" % (k, V, H, "GRBM" if gauss else "RBM")
    check(tag + ": S / max|S|", np.abs(S - S_o).max() / max(1.0, np.abs(S_o).max()), 1e-5, "stats")
    check(tag + ": s_h / max", np.abs(s_h - s_h_o).max() / max(1.0, np.abs(s_h_o).max()), 1e-5, "stats")
    check(tag + ": s_v / max", np.abs(s_v - s_v_o).max() / max(1.0, np.abs(s_v_o).max()), 1e-5, "stats")
    check(tag + ": nv_mean / max|nv|", np.abs(sc.V2[B:].cpu().numpy() - out[1]).max() / max(1.0, np.abs(out[1]).max()),
          2e-6, "nv_mean")
    assert flips <= 3


@pytest.mark.parametrize("V,H,B,gauss,hp", [
    (784, 500, 20, False, dict(lr=0.1, weightcost=2e-4)),                     # c1, the reference's own shape
    (4096, 1024, 512, True, dict(lr=0.001, lambda_2=0.1)),                    # c2 (lr as bench.py: 0.005 diverges)
])
def test_hundred_step_drift_teacher_forced(shadow, V, H, B, gauss, hp):
    """W after 100 teacher-forced steps within 1e-4 relative of the float64 oracle, at the named
    parity shapes."""
    import mdbn_amd
    N = 8 * B
    rs = np.random.RandomState(7)
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.13).astype(np.float32)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(21),
              engine=shadow)
    _, up = rbm.get_cost_updates(k=1, batch_size=B, **hp)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=shadow), data_parallel=None)
    shadow.tie = 4e-6            # weights drifting from the float64 shadow move the probabilities (measured: flips up to 1.1e-6)
    for t in range(100):
        fn(indexes=rs.permutation(N)[:B], momentum=0.6 if t < 50 else 0.9)
    assert shadow.steps == 100
    st = shadow.shadow[rbm.W.tensor.data_ptr()]
    W, W_o = rbm.W.get_value(), st.W
    tag = "100 steps %d->%d" % (V, H)
    check(tag + ": W drift rel", np.abs(W - W_o).max() / np.abs(W_o).max(), 1e-4, "drift100")
    _verdict(shadow, tag, cost_tol=2e-6)
    F = rbm.free_energy(data[:B]).get_value()
    F_o = rbm_np.free_energy(st, data[:B].astype(np.float64))
    check(tag + ": free energy rel", np.abs(F - F_o).max() / np.abs(F_o).max(), 1e-4, "free_energy")   # the north star's parity quantity


@pytest.mark.parametrize("gauss", [False, True])
def test_monitoring_costs_on_device(hip_engine, gauss):
    """get_reconstruction_cost (rbm.py:449-482, :690-699) and get_pseudo_likelihood_cost (rbm.py:421-447) as HIP
    kernels against the oracle, including the rotating bit index and tensor.round's half-away-from-zero."""
    import mdbn_amd
    V, H, B = 130, 70, 37
    rs = np.random.RandomState(5)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(9), engine=hip_engine)
    rbm.hbias.set_value(rs.normal(0, 0.3, H).astype(np.float32))
    rbm.vbias.set_value(rs.normal(0, 0.3, V).astype(np.float32))
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), hbias=rbm.hbias.get_value(), vbias=rbm.vbias.get_value(), gauss=gauss)
    pre = rs.normal(0, 2, (B, V)).astype(np.float32)
    v0 = rs.normal(size=(B, V)).astype(np.float32) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
    got = rbm.get_reconstruction_cost(pre, v0)
    want = rbm_np.reconstruction_cost(st, pre.astype(np.float64), v0.astype(np.float64))
    assert abs(got - want) <= 2e-6 * abs(want)
    x = (rs.randint(0, 7, (B, V)) / 4.0 - 0.5).astype(np.float32)        # quarters: exercises x.5 -> away from zero
    for step in range(3):
        st.bit_i_idx = rbm.bit_i_idx
        got = rbm.get_pseudo_likelihood_cost(x)
        want = rbm_np.pseudo_likelihood_cost(st, x.astype(np.float64))
        assert abs(got - want) <= 1e-4 * abs(want) + 1e-6
    assert rbm.bit_i_idx == 3


def test_tanh_hidden_layer_and_finite_check(hip_engine):
    import mdbn_amd
    from mdbn_amd import mlp
    rs = np.random.RandomState(0)
    layer = mdbn_amd.HiddenLayer(rng=np.random.RandomState(1), input=None, n_in=37, n_out=21, activation=mlp.tanh,
                                 engine=hip_engine)
    x = rs.normal(size=(9, 37)).astype(np.float32)
    out = layer.forward(x)
    want = np.tanh(x.astype(np.float64) @ layer.W.get_value().astype(np.float64) + layer.b.get_value())
    assert np.abs(out.cpu().numpy() - want).max() <= 2e-6
    assert not out._base[:, 21:].any()                              # pad columns stay zero
    t = hip_engine.alloc_matrix(5, 6)
    assert hip_engine.count_nonfinite(t) == 0
    t[2, 3] = float("nan"); t[4, 0] = float("inf"); t[0, 5] = -float("inf")
    assert hip_engine.count_nonfinite(t) == 3


def test_nan_guard_raises_on_divergence(hip_engine):
    """The c2 hyper-parameters of MDBN.py:49 (lr 0.005) diverge on N(0,1) data of this width; with nan_guard the
    step function says so instead of training on NaNs (NanGuardMode's role, rbm.py:542-543)."""
    import mdbn_amd
    V, H, B = 1024, 256, 128
    rs = np.random.RandomState(0)
    data = (40 * rs.normal(size=(4 * B, V))).astype(np.float32)
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), engine=hip_engine)
    _, up = rbm.get_cost_updates(lr=0.5, k=1, lambda_2=0.0, batch_size=B)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=hip_engine), data_parallel=None)
    fn.nan_guard = True
    with pytest.raises(FloatingPointError):
        for t in range(200):
            fn(indexes=np.arange(B) + B * (t % 4), momentum=0.0)


@pytest.mark.parametrize("gauss", [False, True])
def test_gibbs_chain_call_equals_eager_steps_and_oracle(hip_engine, gauss):
    """The sampling loop of rbm.py:806-865 (20 chains, 500 gibbs_vhv steps, 784 -> 500) as ONE library call:
    bit-identical to 500 eager gibbs_vhv calls, which are themselves checked against the oracle at EVERY step
    (the oracle recomputes each half-step from the device's previous state: teacher forcing)."""
    import mdbn_amd
    V, H, B, n = 784, 500, 20, 500
    rs = np.random.RandomState(0)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM

    def make():
        r = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(77),
                engine=hip_engine)
        r.hbias.set_value(np.random.RandomState(1).normal(0, 0.1, H).astype(np.float32))
        r.vbias.set_value(np.random.RandomState(2).normal(0, 0.1, V).astype(np.float32))
        r.W.set_value((0.3 * r.W.get_value()).astype(np.float32))
        return r
    v0 = rs.normal(size=(B, V)).astype(np.float32) if gauss else (rs.uniform(size=(B, V)) < 0.2).astype(np.float32)
    a, b = make(), make()
    st = rbm_np.RBMState(V, H, W=a.W.get_value(), hbias=a.hbias.get_value(), vbias=a.vbias.get_value(), gauss=gauss)
    out_chain = [t.get_value() for t in a.gibbs_vhv_chain(v0, n)]
    assert a._rng_step == 2 * n
    v, worst_h, worst_v, flips = v0, 0.0, 0.0, 0
    for t in range(n):
        out = [x.get_value() for x in b.gibbs_vhv(v)]
        # oracle, from the device's state of the previous step
        _, o_h_mean = rbm_np.propup(st, v.astype(np.float64))
        worst_h = max(worst_h, np.abs(out[1] - o_h_mean).max())
        u = philox_np.uniform(B, H, 77, b.stream_id, 2 * t, 0, 0)
        bad = out[2] != (u < o_h_mean)
        assert np.all(np.abs(u - o_h_mean)[bad] < 1e-6)
        flips += int(bad.sum())
        h_in = out[1] if gauss else out[2]
        uv = philox_np.uniform(B, V, 77, b.stream_id, 2 * t + 1, 0, 0)
        _, o_v_mean, _ = rbm_np.sample_v_given_h(st, h_in.astype(np.float64), uv)
        worst_v = max(worst_v, np.abs(out[4] - o_v_mean).max() / max(1.0, np.abs(o_v_mean).max()))
        if not gauss:
            badv = out[5] != (uv < o_v_mean)
            assert np.all(np.abs(uv - o_v_mean)[badv] < 1e-6)
            flips += int(badv.sum())
        v = out[5]
    assert worst_h <= 2e-6 and worst_v <= 4e-6, (worst_h, worst_v)
    for got, want in zip(out_chain, out):
        assert np.array_equal(got, want)                       # one call == 500 eager calls, bit for bit
    # the reference's sample_fn protocol on top of it
    fn = a.make_sample_fn(v0, n_steps=7)
    mf, smp = fn()
    mf2, smp2 = fn()
    assert mf.shape == (B, V) and not np.array_equal(smp, smp2) or gauss


def test_rccl_communicator_through_the_c_abi(hip_engine):
    """mdbn_comm_unique_id / mdbn_comm_init_rank / mdbn_allreduce_stats / mdbn_comm_destroy with a one-rank
    communicator (RCCL wants one GPU per rank, the box has one): the sum over one rank leaves the buffer as is,
    on the caller's stream, without synchronising."""
    import ctypes as C
    from mdbn_amd import _lib
    eng = hip_engine
    buf = C.create_string_buffer(128)
    _lib.check(eng.lib.mdbn_comm_unique_id(buf), "mdbn_comm_unique_id")
    _lib.check(eng.lib.mdbn_comm_init_rank(eng.ctx, buf.raw, 1, 0), "mdbn_comm_init_rank")
    try:
        x = torch.randn(4096 * 1024 + 5124, device=eng.device)
        want = x.clone()
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(torch.cuda.current_stream())
        _lib.check(eng.lib.mdbn_allreduce_stats(eng.ctx, C.c_void_p(side.cuda_stream), C.c_void_p(x.data_ptr()), x.numel()),
                   "mdbn_allreduce_stats")
        side.synchronize()
        assert torch.equal(x, want)
        assert eng.lib.mdbn_comm_init_rank(eng.ctx, buf.raw, 1, 0) != 0          # one communicator per context
    finally:
        _lib.check(eng.lib.mdbn_comm_destroy(eng.ctx), "mdbn_comm_destroy")
    assert eng.lib.mdbn_allreduce_stats(eng.ctx, None, C.c_void_p(x.data_ptr()), 4) != 0


def test_config5_three_modality_mdbn_at_batch_512(shadow):
    """BASELINE configs[4] at its stated batch size: GE 2048 -> 400 -> 40, miRNA 512 -> 40, SM 256 -> 200 -> 20 with
    CD-5 on the Gaussian first layers, B = 512, concatenated 100 -> joint Bernoulli layer 128 -> 3; one numpy
    RandomState threaded miRNA -> GE -> SM -> top as AMLsm2.py:38-62.  Every CD step on the device is replayed by the
    float64 oracle along the device's chain (k = 5 included)."""
    import mdbn_amd
    from mdbn_amd import MDBN
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(0)
    N, B = 1024, 512
    ge = rs.normal(size=(N, 2048)).astype(np.float32)
    me = rs.normal(size=(N, 512)).astype(np.float32)
    sm = (rs.uniform(size=(N, 256)) < 0.02).astype(np.float32)
    sm = ((sm - sm.mean(0)) / (sm.std(0) + 1e-3)).astype(np.float32)
    rng = np.random.RandomState(123)
    np.random.seed(3)
    shadow.tie = 4e-6
    outs, nets = [], []
    for data, sizes, lr in ((me, [40], [0.002]), (ge, [400, 40], [0.001, 0.1]), (sm, [200, 20], [0.002, 0.1])):
        net, out_t, _ = MDBN.train_bottom_layer(data, None, batch_size=B, k=5, layers_sizes=sizes,
                                                pretraining_epochs=[3] * len(sizes), pretrain_lr=lr, rng=rng)
        outs.append(out_t)
        nets.append(net)
    joint = np.concatenate(outs, axis=1)
    assert joint.shape == (N, 100)
    top = mdbn_amd.DBN(numpy_rng=rng, n_ins=100, gauss=False, hidden_layers_sizes=[128], n_outs=3, engine=shadow)
    top.training(mdbn_amd.shared(joint, engine=shadow), batch_size=B, k=1, pretraining_epochs=[4, 4], pretrain_lr=[0.1, 0.1])
    assert shadow.steps >= 2 * (3 + 6 + 6), shadow.steps
    _verdict(shadow, "c5 MDBN at B = 512, CD-5", [r for net in nets + [top] for r in net.rbm_layers], cost_tol=2e-6,
             param_tol=5e-6)


def _write_table(path, data):
    with open(path, "w") as f:
        f.write("gene\t" + "\t".join("p%d" % i for i in range(data.shape[1])) + "\n")
        for i, row in enumerate(data):
            f.write("g%d\t" % i + "\t".join("%.6f" % v for v in row) + "\n")


def test_loader_to_dbn_training_host_resident_equals_device_resident(shadow, tmp_path):
    """f3 on the device (utils.py:78-119 -> dbn.py:334-517): a TSV table goes through load_n_preprocess_data into
    DBN.training twice -- device-resident (the reference's theano.shared) and host-resident (pinned memory; every
    minibatch's rows gathered over PCIe into a double buffer one step ahead, lower-layer activations streamed in
    chunks) -- on the ShadowEngine, so every CD step is also replayed by the float64 oracle.  Same records, same
    parameters, same outputs, bit for bit."""
    import mdbn_amd
    from mdbn_amd import utils
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(2)
    raw = rs.normal(4, 2, size=(260, 300))                       # 260 features x 300 persons
    raw[7] = 2.0                                                 # zero variance: dropped by the z-score (utils.py:97)
    _write_table(str(tmp_path / "ge.tsv"), raw)
    runs = []
    for resident in ("device", "host"):
        tr, va = utils.load_n_preprocess_data("ge.tsv", holdout=0.1, repeats=1, clip=(-3, 3), shuffle=True,
                                              datadir=str(tmp_path), rng=np.random.RandomState(1), resident=resident)
        assert tr.shape == (270, 259) and va.shape == (30, 259)
        assert isinstance(tr, mdbn_amd.HostTable) == (resident == "host")
        dbn = mdbn_amd.DBN(numpy_rng=np.random.RandomState(123), n_ins=259, hidden_layers_sizes=[64], n_outs=16,
                           engine=shadow)
        dbn.shuffle_rng = np.random.RandomState(5)
        steps0 = shadow.steps
        hist = dbn.training(tr, batch_size=32, k=1, pretraining_epochs=[40, 40], pretrain_lr=[0.005, 0.1],
                            lambda_2=0.1, validation_set_x=va)
        assert shadow.steps - steps0 >= 16
        if resident == "host":
            assert tr._mirror is None, "the training path must not upload the whole table"
        runs.append((hist, [p.get_value() for p in dbn.params], dbn.get_output(tr), dbn.rbm_layers))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(runs[0][2], runs[1][2])
    # a table longer than DBN.host_chunk_rows streams through the lower layers in row chunks: another GEMM plan per
    # chunk (the split-K factor follows the row count), so the activations agree to fp32 summation order
    dbn.host_chunk_rows = 100
    tr._mirror = None
    np.testing.assert_allclose(dbn.get_output(tr), runs[1][2], rtol=0, atol=2e-6)
    assert tr._mirror is None
    _verdict(shadow, "TSV -> DBN.training (259 -> 64 -> 16)", [r for run in runs for r in run[3]], cost_tol=2e-6, param_tol=1e-5)


def test_host_table_prefetch_announces_the_next_minibatch(hip_engine):
    """StepFunction.prefetch: announced rows are staged beside the current step, an unannounced or different minibatch is
    gathered on the spot -- either way the step sees exactly table[indexes] (bitwise equal to the device-resident run)."""
    import mdbn_amd
    V, H, B, N = 256, 128, 64, 1024
    rs = np.random.RandomState(9)
    x = rs.normal(size=(N, V)).astype(np.float32)
    outs = []
    for resident in ("device", "host"):
        rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123),
                            theano_rng=mdbn_amd.RandomStreams(4), engine=hip_engine)
        _, up = rbm.get_cost_updates(lr=0.002, k=1, lambda_2=0.1, batch_size=B)
        fn = mdbn_amd.function(up, mdbn_amd.shared(x, engine=hip_engine, resident=resident), data_parallel=None)
        r2 = np.random.RandomState(1)
        batches = [r2.permutation(N)[:B] for _ in range(8)]
        costs = []
        for t, idx in enumerate(batches):
            costs.append(float(fn(indexes=idx, momentum=0.2)))
            if t in (0, 1, 4):
                fn.prefetch(batches[t + 1])                      # announced correctly
            elif t == 2:
                fn.prefetch(batches[0])                          # announced WRONG: must be ignored
        outs.append((costs, rbm.W.get_value(), rbm.vbias_speed.get_value()))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_row_feeder_ring(hip_engine):
    """mdbn_feeder_*: minibatches submitted ahead come back in order, each slot's device rows equal table[indexes] (pad
    columns zero), ragged row counts, identity batches, more submissions than slots, a bad index refused at acquire
    without stalling the ring, cancel, and the ring-exhaustion guard."""
    import mdbn_amd
    from mdbn_amd import _lib
    eng = hip_engine
    rs = np.random.RandomState(4)
    N, cols = 5000, 300
    t = mdbn_amd.shared(rs.normal(size=(N, cols)).astype(np.float32), engine=eng, resident="host")
    fd = eng.row_feeder(t.host, cols, 256, slots=3, threads=4)
    host = t.host.numpy()
    batches = [torch.from_numpy(rs.randint(0, N, size=n).astype(np.int64)) for n in (256, 17, 1, 256, 100, 255, 3, 256, 64)]
    tickets = [fd.submit(b) for b in batches]
    for b, tk in zip(batches, tickets):
        got = fd.acquire(tk, len(b))
        assert got.shape == (len(b), cols)
        full = got._base[:len(b)] if got._base is not None else got
        want = torch.from_numpy(host[b.numpy()]).to(eng.device)
        assert torch.equal(full[:, :want.shape[1]], want)           # pad columns: the table's own zeros
        fd.release(tk)
    tk = fd.submit(None, 40)                                        # identity: rows 0 .. 39
    assert torch.equal(fd.acquire(tk, 40), torch.from_numpy(host[:40, :cols]).to(eng.device))
    fd.release(tk)
    bad = fd.submit(torch.tensor([1, N, 2], dtype=torch.int64))
    good = fd.submit(batches[1])
    with pytest.raises(_lib.MdbnError, match="out of range"):
        fd.acquire(bad, 3)
    assert torch.equal(fd.acquire(good, 17), torch.from_numpy(host[batches[1].numpy()][:, :cols]).to(eng.device))
    fd.release(good)
    # cancel: nothing submitted before it can be acquired afterwards; the ring keeps working
    old = [fd.submit(b) for b in batches[:5]]
    fd.cancel()
    with pytest.raises(_lib.MdbnError, match="not pending"):
        fd.acquire(old[0], 256)
    tk = fd.submit(batches[4])
    assert torch.equal(fd.acquire(tk, 100), torch.from_numpy(host[batches[4].numpy()][:, :cols]).to(eng.device))
    # all slots held: a fourth acquire must fail instead of waiting for ever
    more = [fd.submit(b) for b in batches[:3]]
    fd.acquire(more[0], 256), fd.acquire(more[1], 17)
    with pytest.raises(_lib.MdbnError, match="slots are held"):
        fd.acquire(more[2], 1)
    for x in (tk, more[0], more[1]):
        fd.release(x)
    assert torch.equal(fd.acquire(more[2], 1), torch.from_numpy(host[batches[2].numpy()][:, :cols]).to(eng.device))
    fd.release(more[2])
    eng.synchronize()
    fd.close()


def test_announced_epoch_feeds_a_host_table(hip_engine):
    """StepFunction.announce: a whole epoch announced up front (as the trainers do) is fed three slots deep; a call that
    departs from the announced order drops the rest and gathers on the spot; a second announce replaces the first -- the
    steps always see table[indexes] (bitwise equal to the device-resident run)."""
    import mdbn_amd
    V, H, B, N = 512, 128, 128, 4096
    rs = np.random.RandomState(9)
    x = rs.normal(size=(N, V)).astype(np.float32)
    r2 = np.random.RandomState(1)
    epoch = [r2.permutation(N)[:n] for n in (B, B, B, 77, B, B, B, B, 5, B)]
    epoch[1] = epoch[1] - N                                          # numpy-style negative indexes (count from the end)
    outs = []
    for resident in ("device", "host"):
        rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123),
                            theano_rng=mdbn_amd.RandomStreams(4), engine=hip_engine)
        _, up = rbm.get_cost_updates(lr=0.002, k=1, lambda_2=0.1, batch_size=B)
        fn = mdbn_amd.function(up, mdbn_amd.shared(x, engine=hip_engine, resident=resident), data_parallel=None)
        dev = [hip_engine.index_tensor(b) for b in epoch]
        costs = []
        fn.announce(dev, host_indexes=epoch)                        # device views + the host values
        for t in range(4):
            costs.append(float(fn(indexes=dev[t], momentum=0.2)))
        costs.append(float(fn(indexes=dev[7], momentum=0.2)))       # NOT the announced one (4): the rest is dropped
        costs.append(float(fn(indexes=dev[5], momentum=0.2)))       # unannounced
        fn.announce(epoch[6:])                                      # host lists only
        fn.announce(dev[4:])                                        # replaces it; device tensors only (one D2H copy)
        for t in range(4, len(epoch)):
            costs.append(float(fn(indexes=dev[t], momentum=0.2, next_indexes=dev[t + 1] if t + 1 < len(epoch) else None)))
        if resident == "host":
            st = fn._staging
            assert st["feeder"] is not None and not st["queue"] and st["held"] is None
        outs.append((costs, rbm.W.get_value(), rbm.vbias_speed.get_value()))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("resident,chunk", [("device", None), ("host", None), ("host", 16)])
def test_free_energy_gap_uses_the_reference_row_sets_on_device(hip_engine, resident, chunk):
    """dbn.py:476-501 on the HIP engine: the layer-0 gap compares the validation set with the WHOLE training set
    (streamed in row chunks from a host-resident table), the upper layers' with its first n_val rows seen through
    get_output(., i-1); each recorded gap equals the float64 oracle's on those rows (tests/_fe_gap.py)."""
    import mdbn_amd
    import _fe_gap
    _fe_gap.run_and_check(mdbn_amd, hip_engine, resident=resident, host_chunk_rows=chunk, tol=2e-5)
