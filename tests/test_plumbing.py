"""Rows f2-f4 of SURVEY 8f on the CPU checker engine: TSV loading / preprocessing, the npz
checkpoint schema (+ resume extension), class post-processing."""
import os

import numpy as np
import pytest

import mdbn_amd
from mdbn_amd import DBN, checkpoint, utils


def write_table(path, data):
    n_feat, n_person = data.shape
    with open(path, "w") as f:
        f.write("gene\t" + "\t".join("p%d" % i for i in range(n_person)) + "\n")
        for i, row in enumerate(data):
            f.write("g%d\t" % i + "\t".join("%.6f" % v for v in row) + "\n")


def test_tsv_loader_and_preprocessing(tmp_path, oracle_engine):
    rs = np.random.RandomState(0)
    raw = rs.normal(5, 2, size=(7, 12))
    raw[3] = 1.0                                   # zero variance -> NaN z-scores -> dropped (utils.py:97)
    write_table(str(tmp_path / "t.tsv"), raw)
    n_data, n_cols, data = utils.import_TCGA_data("t.tsv", str(tmp_path), "float64")
    assert (n_data, n_cols) == (12, 12) and data.shape == (7, 12)
    np.testing.assert_allclose(data, raw, atol=1e-6)
    train, val = utils.preprocess_table(data, holdout=0.25, repeats=1, shuffle=False)
    assert train.shape == (9, 6) and val.shape == (3, 6)      # one row per person, NaN feature gone
    keep = [0, 1, 2, 4, 5, 6]
    z = (raw[keep] - raw[keep].mean(1, keepdims=True)) / raw[keep].std(1, keepdims=True)
    np.testing.assert_allclose(np.concatenate([train, val]), z.T, atol=1e-5)
    # clip + the reference's repeats quirk: only the first n_persons replicated rows are used
    train, val = utils.preprocess_table(data, holdout=0.0, repeats=3, clip=(-1, 1), shuffle=False)
    assert val is None and train.shape == (12, 6) and np.abs(train).max() <= 1.0
    np.testing.assert_allclose(train[:3], np.repeat(np.clip(z.T[:1], -1, 1), 3, axis=0), atol=1e-5)
    # gz + device upload
    import gzip, shutil
    with open(str(tmp_path / "t.tsv"), "rb") as fi, gzip.open(str(tmp_path / "t.tsv.gz"), "wb") as fo:
        shutil.copyfileobj(fi, fo)
    tr, va = utils.load_n_preprocess_data("t.tsv.gz", holdout=0.25, repeats=1, shuffle=True, datadir=str(tmp_path),
                                          rng=np.random.RandomState(1))
    assert tr.shape == (9, 6) and va.shape == (3, 6)
    both = np.concatenate([tr.get_value(), va.get_value()])
    order = lambda a: a[np.lexsort(np.round(a, 3).T[::-1])]
    np.testing.assert_allclose(order(both.astype(np.float64)), order(z.T), atol=1e-4)


def test_mlp_output_from_datafile(tmp_path, oracle_engine):
    DBN.verbose = False
    raw = np.random.RandomState(1).normal(size=(8, 10))
    write_table(str(tmp_path / "ge.tsv"), raw)
    dbn = DBN(numpy_rng=np.random.RandomState(123), n_ins=8, hidden_layers_sizes=[5], n_outs=3)
    out_t, out_v = dbn.MLP_output_from_datafile("ge.tsv", holdout=0.2, datadir=str(tmp_path))
    assert out_t.shape == (8, 3) and out_v.shape == (2, 3)
    train, val = utils.preprocess_table(raw, holdout=0.2, repeats=1, shuffle=False)
    np.testing.assert_allclose(out_t, dbn.get_output(train), rtol=1e-5)


def test_checkpoint_schema_and_resume(tmp_path, oracle_engine):
    DBN.verbose = False
    rs = np.random.RandomState(0)
    ge = DBN(numpy_rng=np.random.RandomState(1), n_ins=10, hidden_layers_sizes=[6], n_outs=4)
    top = DBN(numpy_rng=np.random.RandomState(2), n_ins=4, gauss=False, hidden_layers_sizes=[5], n_outs=3)
    ge.shuffle_rng = np.random.RandomState(3)
    ge.training(rs.normal(size=(40, 10)), batch_size=10, k=1, pretraining_epochs=[6, 6], pretrain_lr=[0.005, 0.1])
    path = str(tmp_path / "parameters_and_classes.npz")
    classes = np.arange(5.0)
    checkpoint.save_network(path, {'ge': ge, 'top': top, 'dm': None}, classes=classes, holdout=0.1, repeats=10,
                            configs={'ge': {'epochs': [8000, 800], 'learning_rate': [0.005, 0.1],
                                            'batch_size': 20, 'k': 1}}, resume=True)
    # the reference's reader (AMLsm2.py:165-205) works on the file as is
    npz = np.load(path, allow_pickle=True)
    cfg = npz['ge_config'].tolist()
    params = npz['ge_params']
    assert cfg['number_of_nodes'] == [10, 6, 4] and cfg['k'] == 1
    assert params[0]['W'].shape == (10, 6) and params[1]['b'].shape == (6,) and params[2]['W'].shape == (6, 4)
    assert float(npz['holdout']) == 0.1 and int(npz['repeats']) == 10
    # round trip, including what the reference loses (vbias, speeds, RNG position)
    nets = checkpoint.load_network(path)
    assert set(nets) >= {'ge', 'top', 'classes'} and np.array_equal(nets['classes'], classes)
    g2 = nets['ge']
    assert not isinstance(nets['top'].rbm_layers[0], mdbn_amd.GRBM) and isinstance(g2.rbm_layers[0], mdbn_amd.GRBM)
    for a, b in zip(ge.params, g2.params):
        assert np.array_equal(a.get_value(), b.get_value()) and a.name == b.name
    for ra, rb in zip(ge.rbm_layers, g2.rbm_layers):
        assert np.array_equal(ra.vbias.get_value(), rb.vbias.get_value())
        assert np.array_equal(ra.W_speed.get_value(), rb.W_speed.get_value())
        assert (ra._rng_step, ra.stream_id, ra.theano_rng.seed) == (rb._rng_step, rb.stream_id, rb.theano_rng.seed)
    # resumed training continues exactly where the original would have
    x = rs.normal(size=(20, 10))
    for d in (ge, g2):
        _, up = d.rbm_layers[0].get_cost_updates(0.005, k=1, lambda_2=0.1, batch_size=10)
        fn = mdbn_amd.function(up, mdbn_amd.shared(x))
        d._c = [float(fn(indexes=np.arange(10) + 10 * t, momentum=0.0)) for t in range(2)]
    # (the CPU checker engine computes in float64 while checkpoints hold float32: compare to 1e-6;
    #  on the HIP engine, whose state IS float32, the continuation is exact -- tests/test_gpu_parity.py)
    np.testing.assert_allclose(ge._c, g2._c, rtol=1e-6)
    np.testing.assert_allclose(ge.params[0].get_value(), g2.params[0].get_value(), rtol=1e-5, atol=1e-7)


def test_class_postprocessing():
    out = np.array([[1, 0, 1], [0, 0, 0], [1, 0, 1], [1, 1, 1], [1, 0, 1], [0, 0, 0], [0, 1, 0]], dtype=np.float64)
    classes, D = utils.find_unique_classes(out)
    assert D.shape == (4, 4) and np.allclose(np.diag(D), 0) and np.allclose(D, D.T)
    assert len(set(classes)) == 4 and classes[0] == classes[2] == classes[4] and classes[1] == classes[5]
    pats = {int(c): tuple(out[i]) for i, c in enumerate(classes)}
    for a in pats:
        for b in pats:
            assert abs(D[a, b] - np.mean(np.array(pats[a]) != np.array(pats[b]))) < 1e-12
    merged = utils.remap_class(classes.astype(int), D, 2)
    assert set(merged) == {0, 1}
    assert merged[0] == merged[2] == merged[4] == 0          # most frequent class -> 0
    assert merged[1] == merged[5] == 1
    assert merged[3] in (0, 1) and merged[6] in (0, 1)


def test_remap_class_hand_case():
    """reference utils.py:124-159 on a hand-worked case: ranks by falling frequency with the higher ID
    first among ties; a dropped class takes the LAST admissible neighbour in rising-distance order;
    a neighbour is skipped when its RANK equals the dropped class's ID (the reference's comparison)."""
    labels = np.array([1, 1, 1, 2, 2, 3, 3, 0])
    # counts 1, 3, 2, 2 -> ranks: class 1 -> 0, class 3 -> 1 (tie with 2: higher ID first), class 2 -> 2, class 0 -> 3
    D = np.array([[0.0, 0.75, 0.5, 0.25],
                  [0.75, 0.0, 0.25, 1.0],
                  [0.5, 0.25, 0.0, 0.75],
                  [0.25, 1.0, 0.75, 0.0]])
    # class 2: neighbours by distance 2, 1, 0, 3 = ranks 2, 0, 3, 1 -> admissible 0 then 1 -> last wins: 1
    # class 0: neighbours 0, 3, 2, 1 = ranks 3, 1, 2, 0 -> rank 0 is skipped (equals the class ID 0) -> 1
    assert utils.remap_class(labels, D, 2).tolist() == [0, 0, 0, 1, 1, 1, 1, 1]
    assert utils.remap_class(labels.astype(np.float64), D, 2).tolist() == [0, 0, 0, 1, 1, 1, 1, 1]
    # keeping everything only relabels by rank
    assert utils.remap_class(labels, D, 4).tolist() == [0, 0, 0, 2, 2, 1, 1, 3]


def test_resume_keeps_frozen_weight_cost_constant(tmp_path, oracle_engine):
    """A Bernoulli layer's weight-cost term uses the W captured when its step function was BUILT
    (rbm.py:415).  A resumed run must keep that constant, not snapshot the resumed W."""
    DBN.verbose = False
    rs = np.random.RandomState(0)
    x = (rs.uniform(size=(40, 8)) < 0.4).astype(np.float64)
    net = DBN(numpy_rng=np.random.RandomState(4), n_ins=8, gauss=False, hidden_layers_sizes=[], n_outs=5)
    rbm = net.rbm_layers[0]
    _, up = rbm.get_cost_updates(0.1, k=1, weightcost=0.05, batch_size=10)      # large cost: the term matters
    fn = mdbn_amd.function(up, mdbn_amd.shared(x))
    for t in range(2):
        fn(indexes=np.arange(10) + 10 * t, momentum=0.5)
    path = str(tmp_path / "resume.npz")
    checkpoint.save_network(path, {'top': net}, resume=True)
    want = [float(fn(indexes=np.arange(10) + 10 * t, momentum=0.5)) for t in (2, 3)]
    net2 = checkpoint.load_network(path)['top']
    r2 = net2.rbm_layers[0]
    _, up2 = r2.get_cost_updates(0.1, k=1, weightcost=0.05, batch_size=10)
    np.testing.assert_allclose(up2.W0.get_value(), up.W0.get_value(), rtol=0, atol=1e-7)   # float32 round trip
    fn2 = mdbn_amd.function(up2, mdbn_amd.shared(x))
    got = [float(fn2(indexes=np.arange(10) + 10 * t, momentum=0.5)) for t in (2, 3)]
    np.testing.assert_allclose(got, want, rtol=1e-6)
    np.testing.assert_allclose(r2.W.get_value(), rbm.W.get_value(), rtol=1e-5, atol=1e-7)
    # without the stored constant the trajectories differ visibly
    net3 = checkpoint.load_network(path)['top']
    net3.rbm_layers[0]._resume_W0 = None
    _, up3 = net3.rbm_layers[0].get_cost_updates(0.1, k=1, weightcost=0.05, batch_size=10)
    fn3 = mdbn_amd.function(up3, mdbn_amd.shared(x))
    for t in (2, 3):
        fn3(indexes=np.arange(10) + 10 * t, momentum=0.5)
    assert np.abs(net3.rbm_layers[0].W.get_value() - rbm.W.get_value()).max() > 2e-5      # (the positive check above holds 1e-7)


def test_lower_layer_cache_sees_direct_parameter_writes(oracle_engine):
    """The cached lower-layer activations are dropped when a lower layer's W is replaced through
    set_value (weights loaded into an existing DBN), not only after training steps."""
    DBN.verbose = False
    rs = np.random.RandomState(0)
    net = DBN(numpy_rng=np.random.RandomState(1), n_ins=6, hidden_layers_sizes=[5], n_outs=3)
    x = mdbn_amd.shared(rs.normal(size=(12, 6)))
    provider = net._layer_input_fn(1, x)
    a = provider().clone()
    assert provider() is provider()                                   # cached
    net.rbm_layers[0].W.set_value(net.rbm_layers[0].W.get_value() * 0.5)
    b = provider()
    assert not np.allclose(a.numpy(), b.numpy())
    np.testing.assert_allclose(b.numpy(), net._forward(x, 0).numpy())


def test_bad_minibatch_index_raises(oracle_engine):
    from mdbn_amd.engine import HipEngine
    with pytest.raises(IndexError):
        HipEngine.index_tensor(oracle_engine, np.array([0, 5, 12]), 12)
    with pytest.raises(IndexError):
        HipEngine.index_tensor(oracle_engine, [-13], 12)
    assert HipEngine.index_tensor(oracle_engine, np.array([-12, 11]), 12).tolist() == [-12, 11]


def test_host_resident_table_on_the_checker_engine(tmp_path, oracle_engine):
    """shared(x, resident="host") (shared.HostTable): the matrix protocol the trainers use, row gathers, and
    load_n_preprocess_data -> DBN.training giving the same network as the device-resident default (on the CPU checker
    there is no PCIe path: this covers the host logic; tests/test_gpu_surface.py covers the streamed path)."""
    DBN.verbose = False
    rs = np.random.RandomState(3)
    x = rs.normal(size=(37, 10)).astype(np.float32)
    t = mdbn_amd.shared(x, resident="host")
    assert isinstance(t, mdbn_amd.HostTable) and t.shape == (37, 10) and len(t) == 37 and t.ndim == 2
    assert t.host.stride(0) == 12 and not t.host[:, 10:].any()           # padded leading dimension, zero padding
    np.testing.assert_array_equal(t.get_value(), x)
    np.testing.assert_array_equal(np.asarray(t[np.array([5, 0, 36])]), x[[5, 0, 36]])
    np.testing.assert_array_equal(oracle_engine.to_numpy(t.rows(slice(3, 9))), x[3:9])
    t.set_value(x[::-1].copy())
    np.testing.assert_array_equal(t.get_value(), x[::-1])
    assert mdbn_amd.shared(t) is t
    assert isinstance(mdbn_amd.shared(x, resident="auto"), mdbn_amd.SharedArray) and \
        not isinstance(mdbn_amd.shared(x, resident="auto"), mdbn_amd.HostTable)
    os.environ["MDBN_HOST_TABLE_BYTES"] = "100"
    try:
        assert isinstance(mdbn_amd.shared(x, resident="auto"), mdbn_amd.HostTable)
    finally:
        del os.environ["MDBN_HOST_TABLE_BYTES"]
    with pytest.raises(ValueError):
        mdbn_amd.shared(x, resident="nowhere")

    raw = rs.normal(3, 2, size=(14, 40))
    write_table(str(tmp_path / "ge.tsv"), raw)
    nets = []
    for resident in ("device", "host"):
        tr, va = utils.load_n_preprocess_data("ge.tsv", holdout=0.2, repeats=1, shuffle=True, datadir=str(tmp_path),
                                              rng=np.random.RandomState(1), resident=resident)
        assert isinstance(tr, mdbn_amd.HostTable) == (resident == "host")
        dbn = DBN(numpy_rng=np.random.RandomState(123), n_ins=14, hidden_layers_sizes=[9], n_outs=4)
        dbn.shuffle_rng = np.random.RandomState(5)
        dbn.host_chunk_rows = 8                         # several chunks through the lower-layer forward
        hist = dbn.training(tr, batch_size=8, k=1, pretraining_epochs=[6, 6], pretrain_lr=[0.005, 0.1],
                            validation_set_x=va)
        nets.append((hist, [p.get_value() for p in dbn.params], dbn.get_output(tr)))
    assert nets[0][0] == nets[1][0]
    for a, b in zip(nets[0][1], nets[1][1]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(nets[0][2], nets[1][2])


def test_host_row_gather_threads(built_lib):
    """mdbn_host_gather_rows: the CPU half of the row feeder (worker threads, row-sized copies) -- out[r] = table[idx[r]]
    for ragged row counts and thread counts, identity gather, pad columns untouched, an index outside the table refused."""
    import ctypes as C
    from mdbn_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(0)
    N, cols, ld, ld_out = 1000, 37, 40, 48
    table = rs.normal(size=(N, ld)).astype(np.float32)
    for n, threads in ((1, 1), (7, 3), (513, 8), (64, 64), (0, 4)):
        idx = rs.randint(0, N, size=n).astype(np.int64)
        out = np.full((max(n, 1), ld_out), 7.0, np.float32)
        rc = lib.mdbn_host_gather_rows(C.c_void_p(table.ctypes.data), N, cols, ld, C.c_void_p(idx.ctypes.data), n,
                                       C.c_void_p(out.ctypes.data), ld_out, threads)
        assert rc == 0, _lib.last_error()
        assert np.array_equal(out[:n, :cols], table[idx, :cols])
        assert (out[:, cols:] == 7.0).all() and (out[n:] == 7.0).all()
    out = np.zeros((5, ld_out), np.float32)
    assert lib.mdbn_host_gather_rows(C.c_void_p(table.ctypes.data), N, cols, ld, None, 5, C.c_void_p(out.ctypes.data), ld_out, 2) == 0
    assert np.array_equal(out[:, :cols], table[:5, :cols])
    bad = np.array([3, N, 5], np.int64)
    assert lib.mdbn_host_gather_rows(C.c_void_p(table.ctypes.data), N, cols, ld, C.c_void_p(bad.ctypes.data), 3,
                                     C.c_void_p(out.ctypes.data), ld_out, 2) == -1
    assert "out of range" in _lib.last_error()


def _interrupted_vs_straight(engine, tmp_path, exact):
    """DBN.training interrupted after a mid-epoch step of the SECOND layer, checkpointed, reloaded and resumed, against the
    same training run straight through (f2: AMLsm2.py:112-205 cannot resume at all; dbn.py:426-508 is the loop)."""
    DBN.verbose = False
    rs = np.random.RandomState(11)
    train = rs.normal(size=(70, 12)).astype(np.float32)
    val = rs.normal(size=(9, 12)).astype(np.float32)
    args = dict(batch_size=16, k=1, pretraining_epochs=[14, 14], pretrain_lr=[0.004, 0.05], lambda_2=0.1, validation_set_x=val)

    def fresh():
        d = DBN(numpy_rng=np.random.RandomState(5), n_ins=12, hidden_layers_sizes=[7], n_outs=4, engine=engine)
        d.shuffle_rng = np.random.RandomState(17)
        return d

    straight = fresh()
    want = straight.training(train, **args)

    class Interrupt(Exception):
        pass

    path = str(tmp_path / "mid.npz")
    seen = []

    def on_step(d):
        seen.append((d.trainer_state['layer'], d.trainer_state['epoch'], d.trainer_state['next_mb']))
        if d.trainer_state['layer'] == 1 and d.trainer_state['epoch'] == 2 and d.trainer_state['next_mb'] == 3:
            checkpoint.save_network(path, {'ge': d}, resume=True)
            raise Interrupt()

    first = fresh()
    with pytest.raises(Interrupt):
        first.training(train, on_step=on_step, **args)
    assert seen[0] == (0, 1, 1) and (1, None, 0) in seen           # every step reported; the layer boundary too
    second = checkpoint.load_network(path, engine=engine)['ge']
    second.shuffle_rng = np.random.RandomState(999)                # (its state comes from the checkpoint, not from here)
    state = second.trainer_state
    assert (state['layer'], state['epoch'], state['next_mb']) == (1, 2, 3)
    got = second.training(train, resume=state, **args)
    assert second.trainer_state is None
    if exact:
        assert got == want
    else:
        assert [[r[0] for r in h] for h in got] == [[r[0] for r in h] for h in want]
        for hg, hw in zip(got, want):
            np.testing.assert_allclose([r[1] for r in hg], [r[1] for r in hw], rtol=1e-5)
    for a, b in zip(second.params, straight.params):
        if exact:
            assert np.array_equal(a.get_value(), b.get_value()), a.name
        else:
            np.testing.assert_allclose(a.get_value(), b.get_value(), rtol=1e-5, atol=1e-7)
    for ra, rb in zip(second.rbm_layers, straight.rbm_layers):
        assert ra._n_updates == rb._n_updates and ra._rng_step == rb._rng_step
    return got


def test_trainer_loop_resume_on_the_checker_engine(tmp_path, oracle_engine):
    # (the CPU checker computes in float64 while a checkpoint holds float32: equal to rounding; exact on the HIP engine)
    _interrupted_vs_straight(oracle_engine, tmp_path, exact=False)
