"""Host-side classes (mirror of the reference's RBM/DBN surface) driven on the CPU checker
engine: construction/draw order, step-function protocol, training loops, caching."""
import numpy as np
import pytest
import torch

import mdbn_amd
from mdbn_amd import DBN, GRBM, RBM, RandomStreams, function, shared
from oracle import rbm_np
from oracle.philox_np import PhiloxDraws


def test_rbm_init_matches_reference_draw_order(oracle_engine):
    rng = np.random.RandomState(123)
    seed = np.random.RandomState(123).randint(2 ** 30)
    rbm = RBM(n_visible=784, n_hidden=500, numpy_rng=rng, theano_rng=RandomStreams(rng.randint(2 ** 30)))
    W = rbm.W.get_value()
    assert rbm.theano_rng.seed == seed == 843828734
    np.testing.assert_allclose(W[0, :4], [0.11645861, -0.03911701, 0.10438896, 0.11984645], atol=1e-7)
    assert abs(float(W.astype(np.float32).sum()) - 256.2355563) < 1e-2
    assert [p.name for p in rbm.params] == ['W', 'hbias', 'vbias']
    assert rbm.W_speed.shape == (784, 500) and not rbm.W_speed.get_value().any()


def test_default_rngs(oracle_engine):
    rbm = RBM(n_visible=6, n_hidden=4)
    assert rbm.theano_rng.seed == 822569775                  # RandomState(1234).randint(2**30)
    dbn = DBN(n_ins=6, hidden_layers_sizes=[5], n_outs=3)
    assert dbn.rbm_layers[0].theano_rng.seed == 843828734    # RandomState(123).randint(2**30)


def test_dbn_structure_and_sharing(oracle_engine):
    DBN.verbose = False
    rng = np.random.RandomState(123)
    dbn = DBN(numpy_rng=rng, n_ins=12, gauss=True, hidden_layers_sizes=[8, 6], n_outs=3)
    assert dbn.number_of_nodes() == [12, 8, 6, 3]
    assert isinstance(dbn.rbm_layers[0], GRBM) and not isinstance(dbn.rbm_layers[1], GRBM)
    assert [p.name for p in dbn.params] == ['W', 'b'] * 3
    for sl, rl in zip(dbn.sigmoid_layers, dbn.rbm_layers):
        assert rl.W is sl.W and rl.hbias is sl.b           # shared (dbn.py:193-194)
    assert len({r.stream_id for r in dbn.rbm_layers}) == 3
    # same numpy draw order as dbn.py:110-114,155-159
    ref = np.random.RandomState(123); ref.randint(2 ** 30)
    for (n_in, n_out), sl in zip([(12, 8), (8, 6), (6, 3)], dbn.sigmoid_layers):
        b = 4 * np.sqrt(6. / (n_in + n_out))
        np.testing.assert_allclose(sl.W.get_value(), ref.uniform(-b, b, (n_in, n_out)).astype(np.float32))
    # W_list / b_list rebuild (AMLsm2.py load_network)
    dbn2 = DBN(n_ins=12, hidden_layers_sizes=[8, 6], n_outs=3,
               W_list=[p.get_value() for p in dbn.params[0::2]], b_list=[p.get_value() for p in dbn.params[1::2]])
    x = np.random.RandomState(0).normal(size=(5, 12)).astype(np.float32)
    np.testing.assert_allclose(dbn.get_output(x), dbn2.get_output(x), rtol=1e-6)
    np.testing.assert_allclose(
        dbn.get_output(x, 1),
        rbm_np.mlp_forward([p.get_value().astype(np.float64) for p in dbn.params[0::2]],
                           [p.get_value().astype(np.float64) for p in dbn.params[1::2]], x.astype(np.float64), 1),
        rtol=1e-6)
    assert dbn.get_output(None) is None


@pytest.mark.parametrize("cls,kw", [(RBM, dict(weightcost=2e-4)), (GRBM, dict(lambda_1=0.01, lambda_2=0.1))])
def test_step_function_equals_oracle_cd_step(oracle_engine, cls, kw):
    V, H, N, B = 10, 6, 24, 8
    data = np.random.RandomState(0).uniform(size=(N, V))
    if cls is RBM:
        data = (data < 0.4).astype(np.float64)
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5), theano_rng=RandomStreams(99))
    st = rbm_np.RBMState(V, H, W=rbm.W.tensor.numpy(), gauss=rbm.gauss)
    if kw.get("weightcost"):
        st.freeze_W0()
    cost, updates = rbm.get_cost_updates(lr=0.05, k=2, batch_size=B, **kw)
    fn = function(updates, shared(data))
    idx = np.array([3, 1, 17, 9, 23, 0, 4, 12], dtype=np.int32)
    for step in range(3):
        got = float(fn(indexes=idx, momentum=0.6))
        want = rbm_np.cd_step(st, data[idx], PhiloxDraws(99, rbm.stream_id, step), lr=0.05, k=2,
                              batch_size=B, momentum=0.6, **kw)
        assert abs(got - want) < 1e-9
    np.testing.assert_allclose(rbm.W.tensor.numpy(), st.W, rtol=1e-12)
    np.testing.assert_allclose(rbm.W_speed.tensor.numpy(), st.W_speed, rtol=1e-12)
    np.testing.assert_allclose(rbm.vbias.tensor.numpy(), st.vbias, rtol=1e-12, atol=1e-15)


def test_step_function_needs_lr_when_symbolic(oracle_engine):
    rbm = GRBM(n_visible=6, n_hidden=4)
    _, updates = rbm.get_cost_updates(mdbn_amd.Scalar('lr'), batch_size=3)
    fn = function(updates, shared(np.zeros((3, 6))))
    with pytest.raises(TypeError):
        fn(indexes=np.arange(3))
    fn(indexes=np.arange(3), momentum=0.0, lr=0.01)
    noisy = GRBM(n_visible=6, n_hidden=4, error_free=False)
    with pytest.raises(NotImplementedError):
        noisy.get_cost_updates(0.1, symbolic_grad=True)


def test_eager_sampling_api(oracle_engine):
    rbm = RBM(n_visible=7, n_hidden=5, theano_rng=RandomStreams(3))
    v = (np.random.RandomState(0).uniform(size=(4, 7)) < 0.5).astype(np.float64)
    pre, mean, sample = rbm.sample_h_given_v(v)
    assert pre.shape == mean.shape == sample.shape == (4, 5)
    s = np.asarray(sample)
    assert set(np.unique(s)) <= {0.0, 1.0}
    out = rbm.gibbs_vhv(v)
    assert len(out) == 6 and out[5].shape == (4, 7)
    out = rbm.gibbs_hvh(sample)
    assert len(out) == 6 and out[2].shape == (4, 7) and out[5].shape == (4, 5)
    assert rbm._rng_step == 5          # every sampling call advanced the stream
    F = rbm.free_energy(v)
    assert F.shape == (4,)
    gap = rbm.free_energy_gap(v[:2], v[2:])
    a, b = rbm.free_energies(v[:2], v[2:])
    assert abs(gap - (b.mean() - a.mean())) < 1e-6
    g = GRBM(n_visible=7, n_hidden=5)
    o = g.gibbs_hvh(sample)
    np.testing.assert_allclose(np.asarray(o[0]), np.asarray(o[1]))   # "pre" is the mean (rbm.py:660)
    np.testing.assert_allclose(np.asarray(o[2]), np.asarray(o[1]))   # error_free: sample == mean


def test_pcd_training_runs_and_advances_chain(oracle_engine):
    data = (np.random.RandomState(0).uniform(size=(40, 9)) < 0.3).astype(np.float64)
    rbm = RBM(n_visible=9, n_hidden=5, numpy_rng=np.random.RandomState(1), theano_rng=RandomStreams(2))
    hist = rbm.learn_model(data, data[:8], 2, 10, 0.0, 0.0,
                           *rbm.get_cost_updates(lr=0.1, k=1, batch_size=10,
                                                 persistent=shared(np.zeros((10, 5)))),
                           display_fn=None, graph_output=False, verbose=False,
                           shuffle_rng=np.random.RandomState(0))
    assert len(hist) == 2 and all(np.isfinite(c) and c < 0 for c, _ in hist)   # pseudo-likelihood < 0
    assert rbm.bit_i_idx == 8 % 9


def test_rbm_training_cd(oracle_engine):
    data = (np.random.RandomState(0).uniform(size=(45, 9)) < 0.3).astype(np.float64)
    rbm = RBM(n_visible=9, n_hidden=5, numpy_rng=np.random.RandomState(1), theano_rng=RandomStreams(2))
    np.random.seed(0)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        hist = rbm.training(data, data[:5], training_epochs=3, batch_size=10, learning_rate=0.1, k=1,
                            initial_momentum=0.6, final_momentum=0.9, weightcost=2e-4, persistent=False)
    assert len(hist) == 3 and hist[-1][0] > 0 and hist[-1][1] is not None


def test_dbn_training_loop_and_cache(oracle_engine):
    DBN.verbose = False
    rs = np.random.RandomState(0)
    train = rs.normal(size=(60, 12)); val = rs.normal(size=(10, 12))
    dbn = DBN(numpy_rng=np.random.RandomState(123), n_ins=12, hidden_layers_sizes=[8], n_outs=4)
    dbn.shuffle_rng = np.random.RandomState(7)
    hist = dbn.training(train, batch_size=20, k=1, pretraining_epochs=[12, 12], pretrain_lr=[0.005, 0.1],
                        lambda_1=0.0, lambda_2=0.1, validation_set_x=val)
    assert len(hist) == 2 and all(len(h) > 0 for h in hist)
    # patience is compared with the iteration count (dbn.py:440,506): 12 "epochs" stop near 12 iters
    assert dbn.rbm_layers[0]._n_updates <= 2 * 12 + 3
    assert all(np.isfinite(r[1]) for h in hist for r in h)
    assert hist[0][0][2] is not None                          # FE gap recorded with a validation set
    out = dbn.get_output(train)
    assert out.shape == (60, 4) and np.all((out > 0) & (out < 1))
    # the cached lower activations were refreshed after layer 0 finished training
    ver, _, cached = dbn._lower_cache[1]
    assert ver[0][0] == dbn.rbm_layers[0]._n_updates
    np.testing.assert_allclose(cached.numpy(), dbn.get_output(train, 0), rtol=1e-6)


def test_dbn_upper_layer_step_matches_manual(oracle_engine):
    """Layer-1 step function consumes sigmoid(x W0 + b0) of the indexed rows (dbn.py:146,307)."""
    DBN.verbose = False
    train = np.random.RandomState(0).normal(size=(16, 10))
    dbn = DBN(numpy_rng=np.random.RandomState(1), n_ins=10, hidden_layers_sizes=[7], n_outs=4)
    fns, _ = dbn.training_functions(shared(train), batch_size=8, k=1)
    r1 = dbn.rbm_layers[1]
    st = rbm_np.RBMState(7, 4, W=r1.W.tensor.numpy()); st.freeze_W0()
    idx = np.arange(4, 12)
    got = float(fns[1](indexes=idx, momentum=0.6, lr=0.1))
    lower = rbm_np.mlp_forward([dbn.params[0].tensor.numpy()], [dbn.params[1].tensor.numpy()], train[idx], 0)
    want = rbm_np.cd_step(st, lower, PhiloxDraws(r1.theano_rng.seed, r1.stream_id, 0), lr=0.1, k=1,
                          weightcost=0.0002, batch_size=8, momentum=0.6)
    assert abs(got - want) < 1e-9
    np.testing.assert_allclose(r1.W_speed.tensor.numpy(), st.W_speed, rtol=1e-10)


def test_mdbn_glue(oracle_engine):
    from mdbn_amd import MDBN
    DBN.verbose = False
    np.random.seed(0)
    rs = np.random.RandomState(0)
    rng = np.random.RandomState(123)
    dbn, out_t, out_v = MDBN.train_bottom_layer(rs.normal(size=(40, 16)), rs.normal(size=(6, 16)), batch_size=20,
                                                k=1, layers_sizes=[8, 5], pretraining_epochs=[4, 4],
                                                pretrain_lr=[0.005, 0.1], rng=rng)
    assert out_t.shape == (40, 5) and out_v.shape == (6, 5)
    joint = np.concatenate([out_t, out_t], axis=1)
    top = DBN(numpy_rng=rng, n_ins=10, gauss=False, hidden_layers_sizes=[24], n_outs=3)
    assert top.number_of_nodes() == [10, 24, 3] and not isinstance(top.rbm_layers[0], GRBM)
    assert top.get_output(joint).shape == (40, 3)


def test_minibatches_mirror():
    from mdbn_amd.utils import get_minibatches_idx
    a = get_minibatches_idx(47, 10, shuffle=True, rng=np.random.RandomState(3))
    b = rbm_np.get_minibatches_idx(47, 10, shuffle=True, rng=np.random.RandomState(3))
    assert list(a[0]) == list(b[0]) and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))


def test_dbn_training_control_flow_scripted(oracle_engine, monkeypatch):
    """dbn.py:426-508 with scripted step costs: momentum 0.0 for the Gaussian layer, 0.6 -> 0.9 at
    epoch 6 for Bernoulli layers; lr per layer; validation every min(20 * n_batches, patience // 2)
    iterations; patience (an ITERATION budget, dbn.py:440,506) doubled to 2 * iter on a > 0.5 %
    improvement; stop when patience <= iter."""
    DBN.verbose = False
    dbn = DBN(numpy_rng=np.random.RandomState(1), n_ins=6, hidden_layers_sizes=[5], n_outs=3)
    dbn.shuffle_rng = np.random.RandomState(0)
    calls = {0: [], 1: []}

    class FakeFn(object):
        def __init__(self, layer, costs):
            self.layer, self.costs, self.n = layer, costs, 0
        def __call__(self, indexes=None, momentum=None, lr=None):
            calls[self.layer].append((len(indexes), momentum, lr))
            c = self.costs(self.n)
            self.n += 1
            return c
        def flush(self):
            pass

    # layer 0: cost improves by 10 % at every validation point -> patience keeps doubling until the
    # epoch budget ends; layer 1: cost flat -> stops when the initial patience runs out
    fns = [FakeFn(0, lambda n: 100.0 * 0.9 ** (n // 4)), FakeFn(1, lambda n: 5.0)]
    monkeypatch.setattr(dbn, "training_functions", lambda **kw: (fns, [dbn.rbm_layers[0].free_energies] * 2))
    train = np.random.RandomState(0).normal(size=(40, 6))
    hist = dbn.training(train, batch_size=10, k=1, pretraining_epochs=[8, 8], pretrain_lr=[0.005, 0.1])
    n_batches = 4
    # layer 0 (GRBM): validation_frequency = min(80, 8 // 2) = 4; improvements keep it alive for all 8 epochs
    assert len(calls[0]) == 8 * n_batches
    assert all(m == 0.0 and lr == 0.005 and n == 10 for n, m, lr in calls[0])
    assert [r[0] for r in hist[0]] == list(range(3, 32, 4))           # iterations of the validation points
    # layer 1 (RBM): flat cost -> first validation sets best, no significant improvement afterwards:
    # patience = max(8, 2 * 3) = 8 after iter 3 ... loop ends when patience <= iter, i.e. at iter 8
    assert len(calls[1]) == 9
    assert all(lr == 0.1 for _, _, lr in calls[1])
    assert [m for _, m, _ in calls[1]] == [0.6] * 9                   # epoch 6 never reached
    # momentum switch at epoch 6 (dbn.py:452-453) with a budget long enough to get there
    fns2 = [FakeFn(0, lambda n: 1.0), FakeFn(1, lambda n: 100.0 * 0.9 ** (n // 2))]
    calls[0].clear(); calls[1].clear()
    monkeypatch.setattr(dbn, "training_functions", lambda **kw: (fns2, [dbn.rbm_layers[0].free_energies] * 2))
    dbn.training(train, batch_size=10, k=1, pretraining_epochs=[1, 7], pretrain_lr=[0.005, 0.1])
    moms = [m for _, m, _ in calls[1]]
    assert moms[:5 * n_batches] == [0.6] * 20 and set(moms[5 * n_batches:]) == {0.9} and len(moms) == 7 * n_batches


@pytest.mark.parametrize("cls", [RBM, GRBM])
def test_symbolic_grad_step(oracle_engine, cls):
    """symbolic_grad=True (rbm.py:341-342,378-390): negative data = the chain's last visible SAMPLE,
    no weight cost, true means -- step function vs the oracle's closed-form free-energy gradient."""
    V, H, N, B = 10, 6, 24, 8
    rs = np.random.RandomState(0)
    data = rs.normal(size=(N, V)) if cls is GRBM else (rs.uniform(size=(N, V)) < 0.4).astype(np.float64)
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5), theano_rng=RandomStreams(99))
    st = rbm_np.RBMState(V, H, W=rbm.W.tensor.numpy(), gauss=rbm.gauss)
    _, updates = rbm.get_cost_updates(lr=0.05, k=2, weightcost=0.3, batch_size=1000, symbolic_grad=True)
    fn = function(updates, shared(data))
    idx = np.array([3, 1, 17, 9, 23, 0, 4, 12])
    for step in range(3):
        got = float(fn(indexes=idx, momentum=0.6))
        want = rbm_np.cd_step(st, data[idx], PhiloxDraws(99, rbm.stream_id, step), lr=0.05, k=2,
                              momentum=0.6, symbolic_grad=True)
        assert abs(got - want) < 1e-9
    np.testing.assert_allclose(rbm.W.tensor.numpy(), st.W, rtol=1e-12)
    np.testing.assert_allclose(rbm.W_speed.tensor.numpy(), st.W_speed, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(rbm.vbias_speed.tensor.numpy(), st.vbias_speed, rtol=1e-11, atol=1e-14)


def test_nan_guard_on_checker_engine(oracle_engine):
    V, H, B = 12, 7, 8
    rs = np.random.RandomState(0)
    data = 50 * rs.normal(size=(32, V))
    rbm = GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5))
    _, up = rbm.get_cost_updates(lr=5.0, k=1, lambda_2=0.0, batch_size=B)
    fn = function(up, shared(data))
    fn.nan_guard = True
    with np.errstate(all="ignore"), pytest.raises(FloatingPointError):
        for t in range(1000):                # |W| grows ~4x per step: float64 overflows after ~450
            fn(indexes=np.arange(B) + B * (t % 4), momentum=0.0)


def test_gibbs_chain_composes_eager_steps_on_checker_engine(oracle_engine):
    V, H, B = 9, 6, 4
    rs = np.random.RandomState(0)
    a = RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5), theano_rng=RandomStreams(3))
    b = RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5), theano_rng=RandomStreams(3))
    v0 = (rs.uniform(size=(B, V)) < 0.4).astype(np.float64)
    out = a.gibbs_vhv_chain(v0, 5)
    v = v0
    for _ in range(5):
        ref = b.gibbs_vhv(v)
        v = ref[5]
    assert a._rng_step == b._rng_step == 10
    for x, y in zip(out, ref):
        assert np.array_equal(x.get_value(), y.get_value())
    fn = a.make_sample_fn(v0, n_steps=3)
    mf, smp = fn()
    assert mf.shape == (B, V) and set(np.unique(smp)) <= {0.0, 1.0}


@pytest.mark.parametrize("resident,chunk", [("device", None), ("host", None), ("host", 16)])
def test_free_energy_gap_uses_the_reference_row_sets(oracle_engine, resident, chunk):
    """dbn.py:476-501: layer 0 compares the validation set with the WHOLE training set, the layers above with its
    first n_val rows through get_output(., i-1).  Every recorded gap equals the oracle's on exactly those rows."""
    import mdbn_amd
    import _fe_gap
    _fe_gap.run_and_check(mdbn_amd, oracle_engine, resident=resident, host_chunk_rows=chunk)
