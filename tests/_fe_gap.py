"""The free-energy-gap monitor of DBN.training against the oracle, on the row sets the reference uses
(dbn.py:476-501): layer 0 = ALL training rows vs the validation rows; layers above = the first n_val training
rows vs the validation rows, both through ``get_output(., i-1)``.  Shared by the CPU-checker test
(tests/test_host_logic.py) and the GPU test (tests/test_gpu_surface.py)."""
import numpy as np

from oracle import rbm_np


def make_tables(n_train=75, n_val=11, n_ins=20, seed=4):
    """Training rows beyond the first n_val are drawn three times as wide, so that the mean free energy over the
    whole set and over its first n_val rows differ by far more than any tolerance: a wrong row set cannot pass."""
    rs = np.random.RandomState(seed)
    train = rs.normal(size=(n_train, n_ins))
    train[n_val:] *= 3.0
    val = rs.normal(size=(n_val, n_ins))
    return train.astype(np.float32), val.astype(np.float32)


def run_and_check(mdbn_amd, engine, resident="device", host_chunk_rows=None, hidden=(12,), n_outs=5, tol=2e-5):
    """Train a len(hidden)+1 layer DBN with a validation set; every recorded gap must equal the oracle's
    ``free_energy_gap`` on the reference's row sets, evaluated with the parameters the network had at that
    moment.  Returns the number of gaps checked per layer."""
    train, val = make_tables()
    n_val = val.shape[0]
    dbn = mdbn_amd.DBN(numpy_rng=np.random.RandomState(123), n_ins=train.shape[1], hidden_layers_sizes=list(hidden),
                       n_outs=n_outs, engine=engine)
    dbn.verbose = False
    dbn.shuffle_rng = np.random.RandomState(9)
    if host_chunk_rows is not None:
        dbn.host_chunk_rows = host_chunk_rows
    seen = []
    inner = dbn._free_energy_gap

    def spy(i, energy_fn, data, held_out):
        gap = inner(i, energy_fn, data, held_out)
        seen.append((i, gap, [p.get_value().astype(np.float64) for p in dbn.params],
                     dbn.rbm_layers[i].vbias.get_value().astype(np.float64)))
        return gap

    dbn._free_energy_gap = spy
    table = mdbn_amd.shared(train, resident=resident, engine=engine)
    n_layers = len(hidden) + 1
    hist = dbn.training(table, batch_size=15, k=1, pretraining_epochs=[30] * n_layers,
                        pretrain_lr=[0.002] + [0.05] * (n_layers - 1), lambda_2=0.1, validation_set_x=val)
    recorded = [[r[2] for r in h if r[2] is not None] for h in hist]
    assert [g for i, g, _, _ in seen] == [g for layer in recorded for g in layer]
    counts = [0] * n_layers
    t64, v64 = train.astype(np.float64), val.astype(np.float64)
    for i, gap, params, vbias in seen:
        Ws, bs = params[0::2], params[1::2]
        s = rbm_np.RBMState(Ws[i].shape[0], Ws[i].shape[1], W=Ws[i], hbias=bs[i], vbias=vbias, gauss=(i == 0))
        if i == 0:
            rows_t, rows_v = t64, v64                                       # dbn.py:477-479: the whole t_set
            wrong = rbm_np.free_energy_gap(s, t64[:n_val], v64)
        else:
            rows_t = rbm_np.mlp_forward(Ws, bs, t64[:n_val], layer=i - 1)   # dbn.py:481-483
            rows_v = rbm_np.mlp_forward(Ws, bs, v64, layer=i - 1)
            wrong = rbm_np.free_energy_gap(s, rbm_np.mlp_forward(Ws, bs, t64, layer=i - 1), rows_v)
        want = rbm_np.free_energy_gap(s, rows_t, rows_v)
        scale = max(1.0, np.abs(rbm_np.free_energy(s, rows_t)).max(), np.abs(rbm_np.free_energy(s, rows_v)).max())
        assert abs(gap - want) <= tol * scale, (i, gap, want, scale)
        # the other row set must be told apart by this test
        assert abs(wrong - want) > 100 * tol * scale, (i, wrong, want, scale)
        counts[i] += 1
    assert all(c > 0 for c in counts), counts
    return counts
