"""Parity tests proper: the HIP path (through the C-ABI) against the oracle on the same
seeded inputs, plus size-independent properties at BASELINE's full size.  Tolerances are
fp32-device vs float64-oracle: probabilities 2e-6 abs, free energy 1e-4 rel
(north star), Bernoulli samples exact outside the near-tie mask |u - p| < 1e-6.  Every
tolerance goes through tests/_margins.check, which records the worst value measured on the
box next to it (profiles/r03zf_tolerance_margins.json)."""
import numpy as np
import pytest
import torch

from oracle import philox_np, rbm_np
from oracle.philox_np import PhiloxDraws
from _margins import check

pytestmark = pytest.mark.gpu

SHAPES = [(6, 4, 3), (64, 32, 8), (130, 70, 37), (784, 500, 20), (1000, 333, 129), (4096, 1024, 512)]


def make(V, H, B, gauss, seed=0, dtype=np.float32):
    rs = np.random.RandomState(seed)
    W = rbm_np.init_W(rs, V, H, dtype)
    hb = rs.normal(0, 0.2, H).astype(dtype)
    vb = rs.normal(0, 0.2, V).astype(dtype)
    x = rs.normal(size=(B, V)).astype(dtype) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(dtype)
    return W, hb, vb, x


def state64(W, hb, vb, gauss):
    return rbm_np.RBMState(W.shape[0], W.shape[1], W=W, hbias=hb, vbias=vb, dtype=np.float64, gauss=gauss)


def ptol(K):
    """Probability tolerance: SURVEY 8d's 2e-6 abs at EVERY reduction length (round 4).  Rounds 2-3 let it grow as sqrt(K)
    beyond K = 1024 (4e-6 at V = 4096: measured 2.2e-6); the error was the accumulator's: six roundings at its own magnitude
    per 32-deep stage.  With the stage-local accumulators of the propup layout (one such rounding per stage) V = 4096
    measures 8e-7."""
    return 2e-6


def gtol(K):
    """One f32 GEMM against ANOTHER f32 GEMM (the bf16-split kernels beside the exact-f32 MFMA kernel, whose K-long chain
    of f32 additions has no stage-local accumulators): 2e-6 of the scale up to K = 1024, growing as sqrt(K) beyond."""
    return 2e-6 * max(1.0, (K / 1024.0) ** 0.5)


def dev(eng, *arrays):
    return [eng.to_device(a) for a in arrays]


@pytest.mark.parametrize("rows,cols,off", [(1, 1, 0), (7, 5, 0), (9, 6, 3), (64, 33, 1022), (512, 1024, 512)])
def test_device_rng_is_bit_exact(hip_engine, rows, cols, off):
    from mdbn_amd import RngAddr
    addr = RngAddr(0x1234ABCD5678, 3, 41, 2, off)
    got = hip_engine.rng_uniform(rows, cols, addr).cpu().numpy()
    assert np.array_equal(got, philox_np.uniform(rows, cols, addr.seed, 3, 41, 2, off))
    z = hip_engine.rng_uniform(rows, cols, addr, normal=True).cpu().numpy()
    np.testing.assert_allclose(z, philox_np.normal(rows, cols, addr.seed, 3, 41, 2, off), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("V,H,B", SHAPES)
def test_propup_sample(hip_engine, V, H, B):
    from mdbn_amd import RngAddr
    W, hb, vb, x = make(V, H, B, True, seed=V)
    dW, dhb, dx = dev(hip_engine, W, hb, x)
    addr = RngAddr(77, 1, 5, 0, 8)
    pre, mean, sample = [t.cpu().numpy() for t in hip_engine.propup(dx, dW, dhb, rng=addr)]
    s = state64(W, hb, vb, True)
    pre_o, mean_o = rbm_np.propup(s, x.astype(np.float64))
    check("propup p [V=%d]" % V, np.abs(mean - mean_o).max(), ptol(V), "prob")
    check("propup pre / max|pre| [V=%d]" % V, np.abs(pre - pre_o).max() / max(1.0, np.abs(pre_o).max()), ptol(V))
    u = philox_np.uniform(B, H, 77, 1, 5, 0, 8).astype(np.float64)
    want = (u < mean_o).astype(np.float32)
    bad = sample != want
    assert np.all(np.abs(u - mean_o)[bad] < 1e-6), "sample mismatch outside the near-tie mask"
    assert bad.sum() <= 2
    assert set(np.unique(sample)) <= {0.0, 1.0}


@pytest.mark.parametrize("V,H,B", SHAPES)
@pytest.mark.parametrize("gauss", [False, True])
def test_propdown_sample(hip_engine, V, H, B, gauss):
    from mdbn_amd import RngAddr
    W, hb, vb, x = make(V, H, B, gauss, seed=H)
    h = (np.random.RandomState(1).uniform(size=(B, H)) < 0.5).astype(np.float32)
    dW, dvb, dh, dx = dev(hip_engine, W, vb, h, x)
    addr = RngAddr(78, 2, 6, 3, 0)
    pre, mean, sample, cost = hip_engine.propdown(dh, dW, dvb, gauss=gauss, add_noise=gauss, rng=addr, v0=dx)
    pre, mean, sample, cost = pre.cpu().numpy(), mean.cpu().numpy(), sample.cpu().numpy(), float(cost)
    s = state64(W, hb, vb, gauss)
    s.error_free = False
    h64 = h.astype(np.float64)
    if gauss:
        z = philox_np.normal(B, V, 78, 2, 6, 3, 0)
        pre_o, mean_o, samp_o = rbm_np.sample_v_given_h(s, h64, z)
        check("propdown GRBM nv_mean / max|nv| [H=%d]" % H, np.abs(mean - mean_o).max() / max(1.0, np.abs(mean_o).max()),
              2e-6, "nv_mean")
        check("propdown GRBM noisy sample [H=%d]" % H, np.abs(sample - samp_o).max(), 1e-5)
        want_cost = ((rbm_np.sigmoid(mean_o) - x) ** 2).sum()
    else:
        u = philox_np.uniform(B, V, 78, 2, 6, 3, 0).astype(np.float64)
        pre_o, mean_o, samp_o = rbm_np.sample_v_given_h(s, h64, u)
        check("propdown RBM p [H=%d]" % H, np.abs(mean - mean_o).max(), ptol(H), "prob")
        bad = sample != samp_o
        assert np.all(np.abs(u - mean_o)[bad] < 1e-6) and bad.sum() <= 2
        want_cost = (x * rbm_np.softplus(-pre_o) + (1 - x) * rbm_np.softplus(pre_o)).sum()
    check("propdown recon cost rel [H=%d]" % H, abs(cost - want_cost) / (abs(want_cost) + 0.05), 2e-6)


@pytest.mark.parametrize("V,H,B", SHAPES)
@pytest.mark.parametrize("gauss", [False, True])
def test_free_energy(hip_engine, V, H, B, gauss):
    W, hb, vb, x = make(V, H, B, gauss, seed=B)
    dW, dhb, dvb, dx = dev(hip_engine, W, hb, vb, x)
    F = hip_engine.free_energy(dx, dW, dhb, dvb, gauss).cpu().numpy()
    s = state64(W, hb, vb, gauss)
    F_o = rbm_np.free_energy(s, x.astype(np.float64))
    rel = np.abs(F - F_o) / np.maximum(np.abs(F_o), 1.0)
    check("free energy rel [V=%d]" % V, rel.max(), 1e-4, "free_energy")            # north-star bound
    # F is a difference of two O(V) terms: fp32-level accuracy is relative to their magnitude
    hid = rbm_np.softplus(x.astype(np.float64) @ s.W + s.hbias).sum(axis=1)
    scale = np.maximum(hid + np.abs(F_o + hid), 1.0)
    check("free energy / term magnitude [V=%d]" % V, (np.abs(F - F_o) / scale).max(), 5e-7)


@pytest.mark.parametrize("V,H,B,k", [(6, 4, 3, 1), (64, 32, 8, 3), (130, 70, 37, 2), (784, 500, 20, 1),
                                     (4096, 1024, 512, 1),
                                     # small layers at B > 64: register-streaming GEMM with row tiles
                                     (1024, 256, 512, 1), (400, 40, 512, 1), (100, 128, 300, 2), (256, 200, 129, 1)])
@pytest.mark.parametrize("gauss", [False, True])
def test_cd_step_statistics(hip_engine, V, H, B, k, gauss):
    """Chain + statistics of one CD-k step vs the oracle.  The oracle's chain is started
    from the DEVICE's positive-phase sample (checked against the twin first) so that a
    near-tie flip cannot fork the two chains (teacher forcing)."""
    from mdbn_amd import RngAddr
    W, hb, vb, x = make(V, H, B, gauss, seed=V + k)
    N = 3 * B
    data = np.concatenate([x, x[::-1], x])[:N]
    idx = np.random.RandomState(2).permutation(N)[:B].astype(np.int64)
    dW, dhb, dvb, ddata = dev(hip_engine, W, hb, vb, data)
    addr = RngAddr(4242, 1, 9, 0, 0)
    stats, sc = hip_engine.cd_step(ddata, idx, dW, dhb, dvb, gauss, k, addr)
    hip_engine.synchronize()
    ldh, ldv = sc.P2.stride(0), sc.V2.stride(0)
    st = stats.cpu().numpy()
    S = st[:V * ldh].reshape(V, ldh)
    assert not S[:, H:].any(), "pad columns of S must stay zero"
    S = S[:, :H]
    s_h, s_v = st[V * ldh:V * ldh + H], st[V * ldh + ldh:V * ldh + ldh + V]
    cost = st[V * ldh + ldh + ldv]
    v0_d = sc.V2[:B].cpu().numpy()
    assert np.array_equal(v0_d, data[idx]), "gather"

    s = state64(W, hb, vb, gauss)
    v0 = data[idx].astype(np.float64)
    draws = PhiloxDraws(4242, 1, 9, 0)
    ph_mean, ph_sample, out = rbm_np.cd_chain(s, v0, draws, k)
    check("cd_step ph_mean [V=%d]" % V, np.abs(sc.P2[:B].cpu().numpy() - ph_mean).max(), ptol(V), "prob")
    if k == 1:
        hs = sc.hs.cpu().numpy()
        bad = hs != ph_sample
        assert np.all(np.abs(draws.u(0, B, H) - ph_mean)[bad] < 1e-6)
        if bad.any():                       # re-run the oracle chain from the device's sample
            dv = None if gauss else draws.u(1, B, V)
            out = rbm_np.gibbs_hvh(s, hs.astype(np.float64), dv, draws.u(2, B, H))
    pre_nv, nv_mean, nv_sample, pre_nh, nh_mean, nh_sample = out
    scale_v = max(1.0, np.abs(nv_mean).max())
    tag = "[V=%d %s]" % (V, "GRBM" if gauss else "RBM")
    check("cd_step nv_mean / max|nv| " + tag, np.abs(sc.V2[B:].cpu().numpy() - nv_mean).max() / scale_v, 2e-6, "nv_mean")
    check("cd_step nh_mean " + tag, np.abs(-sc.P2[B:].cpu().numpy() - nh_mean).max(), 2 * ptol(V), "prob")      # two passes deep
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph_mean, nv_mean, nh_mean)
    check("cd_step S / max|S| " + tag, np.abs(S - S_o).max() / max(1.0, np.abs(S_o).max()), 1e-5, "stats")
    check("cd_step s_h / max " + tag, np.abs(s_h - s_h_o).max() / max(1.0, np.abs(s_h_o).max()), 1e-5, "stats")
    check("cd_step s_v / max " + tag, np.abs(s_v - s_v_o).max() / max(1.0, np.abs(s_v_o).max()), 1e-5, "stats")
    if gauss:
        cost_o = ((rbm_np.sigmoid(pre_nv) - v0) ** 2).sum()
    else:
        cost_o = (v0 * rbm_np.softplus(-pre_nv) + (1 - v0) * rbm_np.softplus(pre_nv)).sum()
    check("cd_step cost rel " + tag, abs(cost - cost_o) / abs(cost_o), 2e-6)


@pytest.mark.parametrize("V,H", [(6, 4), (130, 70), (784, 500), (4096, 1024)])
@pytest.mark.parametrize("l1,l2,wc,mu,frozen", [(0.0, 0.1, 0.0, 0.0, False), (0.01, 0.01, 0.0, 0.0, False),
                                                 (0.0, 0.0, 2e-4, 0.9, True), (0.01, 0.1, 2e-4, 0.6, False)])
def test_apply_update(hip_engine, V, H, l1, l2, wc, mu, frozen):
    rs = np.random.RandomState(V)
    from mdbn_amd.engine import padded_ld
    ldh, ldv = padded_ld(H), padded_ld(V)
    W = rbm_np.init_W(rs, V, H, np.float32)
    W[0, 0] = 0.0                                             # exercises the epsilon in the shrink
    Ws = rs.normal(0, 0.01, (V, H)).astype(np.float32)
    W0 = rbm_np.init_W(rs, V, H, np.float32)
    hb, hbs = rs.normal(size=H).astype(np.float32), rs.normal(size=H).astype(np.float32)
    vb, vbs = rs.normal(size=V).astype(np.float32), rs.normal(size=V).astype(np.float32)
    S = rs.normal(0, 5, (V, H)).astype(np.float32)
    s_h, s_v = rs.normal(size=H).astype(np.float32), rs.normal(size=V).astype(np.float32)
    stats = np.zeros(V * ldh + ldh + ldv + 4, np.float32)
    stats[:V * ldh].reshape(V, ldh)[:, :H] = S
    stats[V * ldh:V * ldh + H] = s_h
    stats[V * ldh + ldh:V * ldh + ldh + V] = s_v
    stats[V * ldh + ldh + ldv] = 123.0
    e = hip_engine
    dW, dWs, dW0 = dev(e, W, Ws, W0)
    dhb, dhbs, dvb, dvbs = dev(e, hb, hbs, vb, vbs)
    dstats = torch.from_numpy(stats).to(e.device)
    cost = e.apply_update(dW, dWs, dW0 if frozen else None, dhb, dhbs, dvb, dvbs, dstats,
                          0.05, l1, l2, wc, mu, 20.0, 17.0, 0.5)
    s = rbm_np.RBMState(V, H, W=W, hbias=hb, vbias=vb)
    s.W_speed, s.hbias_speed, s.vbias_speed = Ws.astype(np.float64), hbs.astype(np.float64), vbs.astype(np.float64)
    s.W0 = W0.astype(np.float64) if frozen else None
    g = rbm_np.rbm_grad(s, S.astype(np.float64), s_h.astype(np.float64), s_v.astype(np.float64), 20, 17, wc,
                        strict_reference=frozen)
    rbm_np.apply_update(s, g[0], g[1], g[2], 0.05, l1, l2, mu)
    assert abs(float(cost) - 61.5) < 1e-5
    for name, t in (("W", dW), ("W_speed", dWs), ("hbias", dhb), ("hbias_speed", dhbs), ("vbias", dvb),
                    ("vbias_speed", dvbs)):
        got, want = t.cpu().numpy(), getattr(s, name)
        check("apply_update %s / max [V=%d]" % (name, V), np.abs(got - want).max() / max(1.0, np.abs(want).max()), 1e-6,
              "update")


def test_classes_same_host_code_both_engines(hip_engine):
    """The reference-shaped host code (DBN.training) run on the HIP engine and on the CPU
    checker engine: costs, learned W and outputs agree (fp32 vs f64) after a short run."""
    import mdbn_amd
    from _oracle_engine import OracleEngine
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(0)
    train, val = rs.normal(size=(96, 40)).astype(np.float32), rs.normal(size=(16, 40)).astype(np.float32)
    results = []
    for eng in (hip_engine, OracleEngine()):
        dbn = mdbn_amd.DBN(numpy_rng=np.random.RandomState(123), n_ins=40, hidden_layers_sizes=[24], n_outs=8,
                           engine=eng)
        dbn.shuffle_rng = np.random.RandomState(5)
        hist = dbn.training(mdbn_amd.shared(train, engine=eng), batch_size=16, k=1, pretraining_epochs=[30, 30],
                            pretrain_lr=[0.005, 0.1], lambda_1=0.01, lambda_2=0.1,
                            validation_set_x=mdbn_amd.shared(val, engine=eng))
        results.append((hist, [p.get_value() for p in dbn.params], dbn.get_output(train)))
    (h_a, p_a, o_a), (h_b, p_b, o_b) = results
    assert [len(x) for x in h_a] == [len(x) for x in h_b]
    for ra, rb in zip(sum(h_a, []), sum(h_b, [])):
        assert ra[0] == rb[0] and abs(ra[1] - rb[1]) <= 1e-4 * abs(rb[1]) + 1e-6
        assert (ra[2] is None) == (rb[2] is None)
        if ra[2] is not None:
            assert abs(ra[2] - rb[2]) <= 1e-3 * max(1.0, abs(rb[2]))
    for a, b in zip(p_a, p_b):
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())
    assert np.abs(o_a - o_b).max() <= 1e-4


def test_full_size_row_linearity(hip_engine):
    """Size-independent property at BASELINE's c2 shape (GRBM 4096->1024, B=512): statistics
    are additive over row shards when the Philox counters are keyed by GLOBAL row -- the
    identity the data-parallel path relies on.  Two half-batches with row
    offsets 0 and 256 must sum to the full-batch statistics (fp32 summation-order tolerance),
    and the positive-phase samples must be bit-identical."""
    from mdbn_amd import RngAddr
    V, H, B = 4096, 1024, 512
    W, hb, vb, x = make(V, H, B, True, seed=11)
    e = hip_engine
    dW, dhb, dvb, dx = dev(e, W, hb, vb, x)
    full, sc = e.cd_step(dx, None, dW, dhb, dvb, True, 1, RngAddr(9, 0, 3, 0, 0))
    full = full.clone()
    hs_full = sc.hs.clone()
    parts = torch.zeros_like(full)
    for lo in (0, 256):
        idx = torch.arange(lo, lo + 256, device=e.device)
        st, sc2 = e.cd_step(dx, idx, dW, dhb, dvb, True, 1, RngAddr(9, 0, 3, 0, lo))
        assert torch.equal(sc2.hs, hs_full[lo:lo + 256]), "samples depend on the sharding"
        parts += st
    scale = float(full.abs().max())
    assert float((parts - full).abs().max()) <= 2e-6 * scale
    # and the full-size statistics agree with the float64 oracle
    s = state64(W, hb, vb, True)
    ph, _, out = rbm_np.cd_chain(s, x.astype(np.float64), PhiloxDraws(9, 0, 3, 0), 1)
    S_o, _, _ = rbm_np.cd_statistics(x.astype(np.float64), ph, out[1], out[4])
    S = full[:V * dW.stride(0)].reshape(V, dW.stride(0))[:, :H].cpu().numpy()
    hs = hs_full.cpu().numpy()
    if np.array_equal(hs, (PhiloxDraws(9, 0, 3, 0).u(0, B, H) < ph).astype(np.float32)):
        assert np.abs(S - S_o).max() <= 1e-5 * np.abs(S_o).max()


def test_multi_step_weights_track_oracle(hip_engine):
    """100 CD-1 steps of the c2-shaped update on a smaller GRBM: learned W stays within 1e-4
    relative of the float64 oracle driven by the same Philox stream."""
    import mdbn_amd
    V, H, B, N = 256, 128, 64, 512
    rs = np.random.RandomState(3)
    data = rs.normal(size=(N, V)).astype(np.float32)
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123),
                        theano_rng=mdbn_amd.RandomStreams(17), engine=hip_engine)
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), gauss=True)
    _, up = rbm.get_cost_updates(lr=0.005, k=1, lambda_1=0.0, lambda_2=0.1, batch_size=B)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=hip_engine), data_parallel=None)
    forks = 0
    for t in range(100):
        idx = rs.permutation(N)[:B]
        c = fn(indexes=idx, momentum=0.0)
        c_o = rbm_np.cd_step(st, data[idx], PhiloxDraws(17, rbm.stream_id, t), lr=0.005, k=1, lambda_2=0.1,
                             batch_size=B)
        if abs(float(c) - c_o) > 1e-4 * abs(c_o):
            forks += 1
    W, W_o = rbm.W.get_value(), st.W
    assert forks <= 1
    check("100-step W drift rel (256->128)", np.abs(W - W_o).max() / np.abs(W_o).max(), 1e-4, "drift100")


def test_checkpoint_resume_is_exact_on_device(hip_engine, tmp_path):
    """Save with the resume extension mid-training, reload, continue: identical to not stopping."""
    import mdbn_amd
    from mdbn_amd import checkpoint
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(0)
    x = rs.normal(size=(64, 40)).astype(np.float32)
    a = mdbn_amd.DBN(numpy_rng=np.random.RandomState(1), n_ins=40, hidden_layers_sizes=[24], n_outs=8, engine=hip_engine)
    def steps(d, n, first):
        r = d.rbm_layers[0]
        _, up = r.get_cost_updates(0.005, k=1, lambda_2=0.1, batch_size=16)
        fn = mdbn_amd.function(up, mdbn_amd.shared(x, engine=hip_engine), data_parallel=None)
        return [float(fn(indexes=np.arange(16) + 16 * ((first + t) % 4), momentum=0.0)) for t in range(n)]
    steps(a, 3, 0)
    path = str(tmp_path / "ck.npz")
    checkpoint.save_network(path, {'ge': a}, resume=True)
    b = checkpoint.load_network(path, engine=hip_engine)['ge']
    ca, cb = steps(a, 3, 3), steps(b, 3, 3)
    assert ca == cb
    for pa, pb in zip(a.params, b.params):
        assert np.array_equal(pa.get_value(), pb.get_value())
    assert np.array_equal(a.rbm_layers[0].W_speed.get_value(), b.rbm_layers[0].W_speed.get_value())


def test_large_n_forward_is_chunked_consistently(hip_engine):
    """get_output / sampling over more rows than the workspace holds at once: the C-ABI chunks
    the rows; means, samples (Philox keyed by absolute row) and free energies must not notice."""
    from mdbn_amd import RngAddr
    V, H, N = 1024, 1024, 9000
    W, hb, vb, x = make(V, H, N, True, seed=3)
    e = hip_engine
    dW, dhb, dvb, dx = dev(e, W, hb, vb, x)
    addr = RngAddr(5, 0, 1, 0, 4)
    _, mean, sample = e.propup(dx, dW, dhb, rng=addr, want_pre=False)
    s = state64(W, hb, vb, True)
    mean_o = rbm_np.propup(s, x.astype(np.float64))[1]
    # 9.2 M probabilities from unsplit 1024-long fp32 chains: the maximum sits further out in the tail
    assert np.abs(mean.cpu().numpy() - mean_o).max() <= 2 * ptol(V)
    u = philox_np.uniform(N, H, 5, 0, 1, 0, 4).astype(np.float64)
    bad = sample.cpu().numpy() != (u < mean_o)
    assert np.all(np.abs(u - mean_o)[bad] < 1e-6) and bad.sum() <= 4
    F = e.free_energy(dx, dW, dhb, dvb, True).cpu().numpy()
    F_o = rbm_np.free_energy(s, x.astype(np.float64))
    assert (np.abs(F - F_o) / np.maximum(np.abs(F_o), 1.0)).max() <= 1e-4


@pytest.mark.parametrize("V,H1,H2", [(4096, 1024, 256), (2048, 1000, 40)])
def test_config4_two_layer_dbn_full_width(hip_engine, V, H1, H2, tmp_path):
    """BASELINE configs[3]: DBN 4096 -> 1024 -> 256, CD-1, layer-wise, B = 512: both layers'
    step functions (layer 1 consuming the cached sigmoid activations of layer 0) against the
    oracle driven with the same Philox streams.  Second case: a RAGGED first layer, 2048 -> 1000: its weights (shared by the
    HiddenLayer and the RBM) live on rows padded to 1024 (Engine.weight_ld) and the step runs on the plane path; the class
    surface, the forward pass, the free energy and a checkpoint round trip see [2048, 1000]."""
    import mdbn_amd
    mdbn_amd.DBN.verbose = False
    N, B = 1024, 512
    x = np.random.RandomState(0).normal(size=(N, V)).astype(np.float32)
    dbn = mdbn_amd.DBN(numpy_rng=np.random.RandomState(123), n_ins=V, hidden_layers_sizes=[H1], n_outs=H2,
                       engine=hip_engine)
    ld1 = (H1 + 127) // 128 * 128
    assert dbn.rbm_layers[0].W.tensor.stride(0) == ld1 and dbn.sigmoid_layers[0].W is dbn.rbm_layers[0].W
    assert dbn.params[0].get_value().shape == (V, H1)
    W0, W1 = dbn.params[0].get_value(), dbn.params[2].get_value()
    fns, fe_fns = dbn.training_functions(mdbn_amd.shared(x, engine=hip_engine), batch_size=B, k=1,
                                         lambda_1=0.0, lambda_2=0.1)
    r0, r1 = dbn.rbm_layers
    s0 = rbm_np.RBMState(V, H1, W=W0, gauss=True)
    s1 = rbm_np.RBMState(H1, H2, W=W1); s1.freeze_W0()
    idx = [np.arange(B), np.arange(B) + B]
    for t in range(2):                                   # layer 0: GRBM, lr 0.001 (stable), momentum 0
        c = float(fns[0](indexes=idx[t], momentum=0.0, lr=0.001))
        # teacher forcing: the oracle's chain starts from the device's positive-phase sample, which must
        # equal the oracle's own wherever the uniform is not within fp32 rounding of the probability
        hs_dev = hip_engine.cd_scratch(B, V, H1, False, fns[0]._data().stride(0), r0.W.tensor.stride(0)).hs
        assert not hs_dev.cpu().numpy()[:, H1:].any()
        hs_dev = hs_dev.cpu().numpy()[:, :H1].astype(np.float64)
        draws = PhiloxDraws(r0.theano_rng.seed, r0.stream_id, t)
        p_o = rbm_np.propup(s0, x[idx[t]].astype(np.float64))[1]
        u = draws.u(0, B, H1)
        flips = hs_dev != (u < p_o)
        assert np.all(np.abs(u - p_o)[flips] < 1e-6) and flips.sum() <= 4, "sample differs away from a tie"
        c_o = rbm_np.cd_step(s0, x[idx[t]], draws, lr=0.001, k=1, lambda_2=0.1, batch_size=B, chain_start=hs_dev)
        assert abs(c - c_o) <= 1e-4 * abs(c_o)
    assert np.abs(r0.W.get_value() - s0.W).max() <= 2e-6
    lower = rbm_np.mlp_forward([s0.W], [s0.hbias], x.astype(np.float64), 0)
    for t in range(2):                                   # layer 1: Bernoulli RBM on sigmoid(x W0 + b0)
        c = float(fns[1](indexes=idx[t], momentum=0.6, lr=0.1))
        c_o = rbm_np.cd_step(s1, lower[idx[t]], PhiloxDraws(r1.theano_rng.seed, r1.stream_id, t), lr=0.1, k=1,
                             weightcost=0.0002, batch_size=B, momentum=0.6)
        assert abs(c - c_o) <= 2e-4 * abs(c_o), (t, c, c_o)
    assert np.abs(r1.W_speed.get_value() - s1.W_speed).max() <= 1e-4 * np.abs(s1.W_speed).max() + 1e-6
    out = dbn.get_output(x[:64])
    out_o = rbm_np.mlp_forward([s0.W, s1.W], [s0.hbias, s1.hbias], x[:64].astype(np.float64))
    assert np.abs(out - out_o).max() <= 1e-4
    ft, fv = fe_fns[0](x[:64], x[64:128])
    np.testing.assert_allclose(ft, rbm_np.free_energy(s0, x[:64].astype(np.float64)), rtol=1e-4, atol=1e-2)
    # checkpoint round trip (AMLsm2.py:112-205 schema): logical shapes on disk, the padded rows again after loading
    from mdbn_amd import checkpoint
    path = str(tmp_path / "net.npz")
    checkpoint.save_network(path, {"dbn": dbn})
    back = checkpoint.load_network(path, engine=hip_engine)["dbn"]
    assert back.rbm_layers[0].W.tensor.stride(0) == ld1
    for p, q in zip(dbn.params, back.params):
        assert np.array_equal(p.get_value(), q.get_value())
    assert np.array_equal(back.get_output(x[:64]), out)


def test_config5_three_modality_mdbn(hip_engine):
    """BASELINE configs[4] at reduced row count: GE 2048 -> 400 -> 40, miRNA 512 -> 40 (CD-5),
    SM 256 -> 200 -> 20, concatenated 100 -> joint Bernoulli layer 128; one numpy RandomState
    threaded ME -> GE -> SM -> top as AMLsm2.py:38-62.  Same host code on both engines."""
    import mdbn_amd
    from _oracle_engine import OracleEngine
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(0)
    N = 128
    ge = rs.normal(size=(N, 2048)).astype(np.float32)
    me = rs.normal(size=(N, 512)).astype(np.float32)
    sm = (rs.uniform(size=(N, 256)) < 0.02).astype(np.float32)
    sm = ((sm - sm.mean(0)) / (sm.std(0) + 1e-3)).astype(np.float32)
    outs = []
    for eng in (hip_engine, OracleEngine()):
        rng = np.random.RandomState(123)
        res = []
        for data, sizes, k, lr in ((me, [40], 5, [0.002]), (ge, [400, 40], 1, [0.001, 0.1]), (sm, [200, 20], 1, [0.002, 0.1])):
            d = mdbn_amd.DBN(numpy_rng=rng, n_ins=data.shape[1], hidden_layers_sizes=sizes[:-1], n_outs=sizes[-1], engine=eng)
            d.shuffle_rng = np.random.RandomState(9)
            d.training(mdbn_amd.shared(data, engine=eng), batch_size=32, k=k, pretraining_epochs=[8] * len(sizes),
                       pretrain_lr=lr, lambda_1=0.0, lambda_2=0.1)
            res.append(d.get_output(data))
        joint = np.concatenate(res, axis=1)
        top = mdbn_amd.DBN(numpy_rng=rng, n_ins=100, gauss=False, hidden_layers_sizes=[128], n_outs=3, engine=eng)
        top.shuffle_rng = np.random.RandomState(10)
        top.training(mdbn_amd.shared(joint, engine=eng), batch_size=32, k=1, pretraining_epochs=[8, 8], pretrain_lr=[0.1, 0.1])
        outs.append((joint, top.get_output(joint), top.params[0].get_value()))
    (ja, ta, wa), (jb, tb, wb) = outs
    assert ja.shape == (N, 100) and ta.shape == (N, 3)
    assert np.abs(ja - jb).max() <= 2e-4
    assert np.abs(wa - wb).max() <= 2e-4 * max(1.0, np.abs(wb).max())
    # fp32 vs float64 after four trained, stacked layers (sampling included): drift, not error
    assert np.abs(ta - tb).max() <= 3e-3


def _sweep_shapes():
    rs = np.random.RandomState(2024)
    shapes = [(1, 1, 1), (2, 3, 1), (5, 7, 2), (31, 33, 3), (32, 32, 4), (33, 31, 5), (63, 65, 7), (64, 64, 1),
              (127, 129, 9), (129, 127, 130), (257, 61, 33), (61, 257, 65), (500, 784, 20), (1021, 509, 131),
              (96, 2053, 17), (2053, 96, 260), (1024, 256, 512), (256, 200, 512), (100, 128, 512), (40, 400, 97)]
    for _ in range(8):
        shapes.append((int(rs.randint(1, 700)), int(rs.randint(1, 700)), int(rs.randint(1, 300))))
    return shapes


@pytest.mark.parametrize("V,H,B", _sweep_shapes())
def test_shape_sweep_chain_and_update(hip_engine, V, H, B):
    """Tile edges everywhere: sizes below one tile / one slice, primes, ragged tails, single rows.
    One CD-2 step of a Bernoulli RBM and one CD-1 step of a GRBM through the classes vs the oracle."""
    import mdbn_amd
    rs = np.random.RandomState(V * 7 + H * 3 + B)
    N = B + 3
    # The oracle follows the device's recorded chain (mdbn_cd_args.trace_*, rbm_np.cd_chain_forced), so a
    # Bernoulli draw within fp32 rounding of its probability cannot fork the two -- whatever the seed.
    seed = 77
    hip_engine.trace_chain = True
    for cls, gauss, hp, k in ((mdbn_amd.RBM, False, dict(lr=0.1, weightcost=2e-4), 2),
                              (mdbn_amd.GRBM, True, dict(lr=0.002, lambda_1=0.01, lambda_2=0.1), 1)):
        data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.4).astype(np.float32)
        rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(V + H), theano_rng=mdbn_amd.RandomStreams(seed),
                  engine=hip_engine)
        st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), gauss=gauss)
        if hp.get("weightcost"):
            st.freeze_W0()
        _, up = rbm.get_cost_updates(k=k, batch_size=B, **hp)
        fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=hip_engine), data_parallel=None)
        idx = rs.permutation(N)[:B]
        for t in range(2):
            c = float(fn(indexes=idx, momentum=0.5))
            sc = hip_engine.last_scratch
            forced = (sc.trace_h.cpu().numpy()[:, :, :H], None if gauss else sc.trace_v.cpu().numpy()[:, :, :V])
            c_o = rbm_np.cd_step(st, data[idx], PhiloxDraws(seed, rbm.stream_id, t), k=k, batch_size=B, momentum=0.5,
                                 forced=forced, **hp)
            assert abs(c - c_o) <= 2e-4 * abs(c_o) + 1e-6, (cls.__name__, t, c, c_o)
        for name in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed"):
            got, want = getattr(rbm, name).get_value(), getattr(st, name)
            # two steps of fp32 chains (k up to 2053 long) against float64
            assert np.abs(got - want).max() <= 5e-5 * max(1.0, np.abs(want).max()), (cls.__name__, name)
        F = rbm.free_energy(data).get_value()
        F_o = rbm_np.free_energy(st, data.astype(np.float64))
        assert np.abs(F - F_o).max() <= 1e-4 * max(1.0, np.abs(F_o).max())
    hip_engine.trace_chain = False


def test_pcd_on_device(hip_engine):
    """PCD-k (rbm.py:308-311,367-371): persistent chain as chain start, replaced by nh_sample,
    pseudo-likelihood cost with the rotating bit index -- against the oracle, 4 steps."""
    import mdbn_amd
    V, H, B, N = 96, 40, 16, 64
    rs = np.random.RandomState(5)
    data = (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    rbm = mdbn_amd.RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(1), theano_rng=mdbn_amd.RandomStreams(9),
                       engine=hip_engine)
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value())
    st.persistent = np.zeros((B, H))
    chain = mdbn_amd.shared(np.zeros((B, H), dtype=np.float32), engine=hip_engine)
    _, up = rbm.get_cost_updates(lr=0.1, k=2, batch_size=B, persistent=chain)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=hip_engine), data_parallel=None)
    for t in range(4):
        idx = rs.permutation(N)[:B]
        c = float(fn(indexes=idx, momentum=0.0))
        c_o = rbm_np.cd_step(st, data[idx], PhiloxDraws(9, rbm.stream_id, t), lr=0.1, k=2, batch_size=B,
                             persistent=True)
        assert abs(c - c_o) <= 1e-4 * abs(c_o), (t, c, c_o)
        assert np.array_equal(up.persistent.get_value(), st.persistent.astype(np.float32)), "chain diverged"
    assert rbm.bit_i_idx == st.bit_i_idx == 4
    assert np.abs(rbm.W.get_value() - st.W).max() <= 1e-5
    with pytest.raises(ValueError):                       # ragged minibatch vs fixed chain (rbm.py:416)
        fn(indexes=np.arange(B - 1), momentum=0.0)


def test_empty_and_degenerate_inputs(hip_engine):
    import mdbn_amd
    mdbn_amd.DBN.verbose = False
    rbm = mdbn_amd.RBM(n_visible=12, n_hidden=5, engine=hip_engine)
    empty = np.zeros((0, 12), dtype=np.float32)
    pre, mean, sample = rbm.sample_h_given_v(empty)
    assert pre.shape == mean.shape == sample.shape == (0, 5)
    assert rbm.free_energy(empty).shape == (0,)
    dbn = mdbn_amd.DBN(n_ins=12, hidden_layers_sizes=[5], n_outs=3, engine=hip_engine)
    assert dbn.get_output(empty).shape == (0, 3)
    with pytest.raises(AssertionError):
        rbm.sample_h_given_v(np.zeros((3, 11), dtype=np.float32))      # wrong width
    one = rbm.sample_h_given_v(np.ones((1, 12), dtype=np.float32))     # a single row
    assert one[1].shape == (1, 5)


@pytest.mark.parametrize("gauss", [False, True])
def test_symbolic_grad_on_device(hip_engine, gauss):
    """compute_symbolic_grad path (rbm.py:378-390) through the C-ABI: sample statistics."""
    import mdbn_amd
    V, H, B, N = 130, 70, 24, 96
    rs = np.random.RandomState(3)
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.4).astype(np.float32)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(1), theano_rng=mdbn_amd.RandomStreams(21),
              engine=hip_engine)
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), gauss=gauss)
    _, up = rbm.get_cost_updates(lr=0.01, k=2, batch_size=B, symbolic_grad=True)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=hip_engine), data_parallel=None)
    for t in range(3):
        idx = rs.permutation(N)[:B]
        c = float(fn(indexes=idx, momentum=0.5))
        c_o = rbm_np.cd_step(st, data[idx], PhiloxDraws(21, rbm.stream_id, t), lr=0.01, k=2, momentum=0.5,
                             symbolic_grad=True)
        assert abs(c - c_o) <= 1e-4 * abs(c_o)
    for name in ("W", "W_speed", "hbias_speed", "vbias_speed"):
        got, want = getattr(rbm, name).get_value(), getattr(st, name)
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), name


def test_many_rows_gather_and_forward(hip_engine):
    """More rows than one grid dimension holds (65 535): chunked gather (indexed and identity)
    and a forward pass / free energy over 70 001 rows."""
    import mdbn_amd
    e = hip_engine
    N, V, H = 70001, 12, 20
    rs = np.random.RandomState(0)
    x = rs.normal(size=(N, V)).astype(np.float32)
    sx = mdbn_amd.shared(x, engine=e)
    idx = rs.permutation(N).astype(np.int64)
    got = sx[idx].get_value()
    assert np.array_equal(got, x[idx])
    assert np.array_equal(sx[idx.astype(np.int32)[:66000]].get_value(), x[idx[:66000]])
    W, hb, vb, _ = make(V, H, 1, True, seed=1)
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, W=W, hbias=hb, vbias=vb, engine=e)
    mean = np.asarray(rbm.propup(sx)[1])
    s = state64(W, hb, vb, True)
    assert np.abs(mean - rbm_np.propup(s, x.astype(np.float64))[1]).max() <= 2e-6
    F = rbm.free_energy(sx).get_value()
    assert np.abs(F - rbm_np.free_energy(s, x.astype(np.float64))).max() <= 1e-4
    # identity minibatch (indexes=None) over all rows through the step function
    _, up = rbm.get_cost_updates(lr=1e-4, k=1, lambda_2=0.1)
    c = float(mdbn_amd.function(up, sx, data_parallel=None)())
    c_o = rbm_np.cd_step(s, x, PhiloxDraws(rbm.theano_rng.seed, rbm.stream_id, 0), lr=1e-4, k=1, lambda_2=0.1)
    assert abs(c - c_o) <= 1e-4 * abs(c_o)


def test_bitwise_determinism_and_race_screen(hip_engine):
    """No atomics anywhere, fixed summation orders: the same 30 steps run three times give
    bit-identical parameters (also a race screen for the producer / consumer GEMM, whose LDS
    double buffer would show up here as run-to-run differences).  Both headline-sized and ragged."""
    import mdbn_amd
    for V, H, B, k, gauss in ((4096, 1024, 512, 1, True), (777, 333, 50, 2, False)):
        rs = np.random.RandomState(1)
        N = 4 * B
        data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
        sx = mdbn_amd.shared(data, engine=hip_engine)
        finals = []
        for rep in range(3):
            cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
            rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5),
                      engine=hip_engine)
            hp = dict(lr=0.001, lambda_2=0.1) if gauss else dict(lr=0.05, weightcost=2e-4)
            _, up = rbm.get_cost_updates(k=k, batch_size=B, **hp)
            fn = mdbn_amd.function(up, sx, data_parallel=None)
            perm = np.random.RandomState(2).permutation(N)
            costs = [float(fn(indexes=perm[(t % 4) * B:(t % 4 + 1) * B], momentum=0.5)) for t in range(30)]
            finals.append((rbm.W.get_value(), rbm.W_speed.get_value(), rbm.vbias.get_value(), costs))
        for other in finals[1:]:
            assert np.array_equal(finals[0][0], other[0]) and np.array_equal(finals[0][1], other[1])
            assert np.array_equal(finals[0][2], other[2]) and finals[0][3] == other[3]
        assert np.isfinite(finals[0][0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("V,H,B", [(100, 260, 37), (200, 1024, 512), (37, 19, 5), (4096, 1024, 4096),
                                   (130, 2050, 1030)])
def test_fused_epilogue_equals_unfused(hip_engine, V, H, B):
    """GEMMs that need no split-K apply bias + activation + Philox sampling on the MFMA
    accumulators (fused_act_epilogue); the result must be BITWISE what the slab + epilogue pair
    produces (same accumulation order, same Philox addressing), the cost equal up to summation
    order, and a whole CD step's statistics bitwise too."""
    eng = hip_engine
    rs = np.random.RandomState(V + H + B)
    W = eng.to_device((0.05 * rs.randn(V, H)).astype(np.float32))
    hb = eng.to_device((0.1 * rs.randn(H)).astype(np.float32))
    vb = eng.to_device((0.1 * rs.randn(V)).astype(np.float32))
    v = eng.to_device(rs.randn(B, V).astype(np.float32))
    hsrc = eng.to_device((rs.rand(B, H) < 0.5).astype(np.float32))
    from mdbn_amd.engine import RngAddr
    out = {}
    try:
        for fused in (1, 0):
            eng.set_option("fused_epilogue", fused)
            up = eng.propup(v, W, hb, rng=RngAddr(7, 1, 3, 0))
            dn = eng.propdown(hsrc, W, vb, gauss=False, rng=RngAddr(7, 1, 3, 1), v0=(v > 0).float())
            dg = eng.propdown(hsrc, W, vb, gauss=True, add_noise=True, rng=RngAddr(7, 1, 3, 1), v0=v)
            out[fused] = [t.cpu().numpy() for t in up] + [t.cpu().numpy() for t in dn] + \
                         [t.cpu().numpy() for t in dg[1:]]
    finally:
        eng.set_option("fused_epilogue", 1)
    names = ["up.pre", "up.mean", "up.sample", "dn.pre", "dn.mean", "dn.sample", "dn.cost",
             "dg.mean", "dg.sample", "dg.cost"]
    for name, a, b in zip(names, out[1], out[0]):
        if name.endswith("cost"):
            assert abs(float(a) - float(b)) <= 2e-6 * abs(float(b)) + 1e-6, name
        else:
            assert np.array_equal(a, b), name


@pytest.mark.gpu
@pytest.mark.parametrize("V,H,B", [(784, 500, 20), (37, 19, 5), (1000, 130, 33), (2050, 70, 64), (300, 2049, 1),
                                   (16384, 400, 20), (8, 8, 32),
                                   (1024, 256, 512), (400, 40, 512), (100, 128, 300), (256, 200, 129)])
def test_skinny_gemm_matches_oracle_and_tile_kernel(hip_engine, V, H, B):
    """Forward passes of <= 64 rows, and of small layers at any batch size (operands L2-resident),
    run on skinny_gemm_kernel (no LDS staging, 8 waves split K, in-block reduction; 32- or 64-row
    tiles).  Pre-activations must match the float64
    oracle to the f32 accumulation tolerance and the 128-row tile kernel to the same; samples may
    differ from the tile kernel's only where u is within that tolerance of the mean."""
    eng = hip_engine
    rs = np.random.RandomState(V * 7 + H * 3 + B)
    Wn = (0.05 * rs.randn(V, H)).astype(np.float32)
    hbn, vbn = (0.1 * rs.randn(H)).astype(np.float32), (0.1 * rs.randn(V)).astype(np.float32)
    vn = rs.randn(B, V).astype(np.float32)
    hn = (rs.rand(B, H) < 0.5).astype(np.float32)
    W, hb, vb, v, hsrc = [eng.to_device(a) for a in (Wn, hbn, vbn, vn, hn)]
    from mdbn_amd.engine import RngAddr
    out = {}
    try:
        for skinny in (1, 0):
            eng.set_option("skinny_gemm", skinny)
            up = eng.propup(v, W, hb, rng=RngAddr(11, 2, 9, 0))
            dn = eng.propdown(hsrc, W, vb, gauss=False, rng=RngAddr(11, 2, 9, 1), v0=(v > 0).float())
            out[skinny] = [t.cpu().numpy() for t in up] + [t.cpu().numpy() for t in dn]
    finally:
        eng.set_option("skinny_gemm", 1)
    pre_up = vn.astype(np.float64) @ Wn.astype(np.float64) + hbn
    pre_dn = hn.astype(np.float64) @ Wn.astype(np.float64).T + vbn
    tol_up = 4 * gtol(V) * max(1.0, np.abs(pre_up).max())     # pre-activation, not probability
    tol_dn = 4 * gtol(H) * max(1.0, np.abs(pre_dn).max())
    for kern in (1, 0):
        assert np.abs(out[kern][0] - pre_up).max() <= tol_up
        assert np.abs(out[kern][3] - pre_dn).max() <= tol_dn
    # samples: identical Philox words; a flip needs u within the rounding band of the mean
    for (m_i, s_i, tol) in ((1, 2, tol_up), (4, 5, tol_dn)):
        flips = out[1][s_i] != out[0][s_i]
        assert flips.mean() <= 1e-3
        assert np.abs(out[1][m_i] - out[0][m_i]).max() <= tol
    assert abs(float(out[1][6]) - float(out[0][6])) <= 1e-5 * abs(float(out[0][6])) + 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("V,H,B", [(4096, 1024, 512), (130, 70, 37), (784, 500, 20), (2050, 258, 64),
                                   (1024, 256, 512), (100, 128, 300), (2048, 400, 512)])
@pytest.mark.parametrize("hp", [dict(lambda_2=0.1), dict(lambda_1=0.01, lambda_2=0.01), dict(weightcost=2e-4, momentum=0.9)],
                         ids=["l2", "l1l2", "wc_mu"])
def test_fused_update_is_bitwise_the_separate_update(hip_engine, V, H, B, hp):
    """mdbn_cd_train_step applies the weight update inside the statistics GEMM (tile parked in LDS) and
    the bias / cost half in the finalize kernel; 5 steps must leave every parameter, speed and cost
    BITWISE equal to the statistics GEMM -> S -> update_kernel sequence."""
    import mdbn_amd
    eng = hip_engine
    rs = np.random.RandomState(5)
    data = rs.randn(4 * B, V).astype(np.float32)
    hp = dict(hp)
    momentum = hp.pop("momentum", 0.5)
    res = {}
    try:
        for fused in (1, 0):
            eng.set_option("fused_update", fused)
            rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123),
                                theano_rng=mdbn_amd.RandomStreams(77), engine=eng)
            _, up = rbm.get_cost_updates(lr=0.001, k=1, batch_size=B, **hp)
            fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel=None)
            costs = [float(fn(indexes=np.arange(B) + (t % 4) * B, momentum=momentum)) for t in range(5)]
            res[fused] = [getattr(rbm, k).get_value() for k in
                          ("W", "W_speed", "hbias", "hbias_speed", "vbias", "vbias_speed")] + [np.array(costs)]
    finally:
        eng.set_option("fused_update", 1)
    for a, b in zip(res[1], res[0]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("V,H,B", [(20000, 4000, 1024), (4097, 1031, 777), (16384, 400, 20), (3000, 3000, 4096),
                                   (70000, 96, 40), (96, 70000, 40)])
def test_large_shapes_against_rocblas(hip_engine, V, H, B):
    """Sizes the CPU oracle cannot reach in seconds: pre-activations of both passes against
    torch.matmul (rocBLAS, an independent f32 GEMM) on the same device, every kernel family
    (LDS-tiled split / unsplit + fused epilogue, one and two MFMA waves per SIMD, register-streaming),
    and the sampled values against the pre-activations they must derive from."""
    import torch
    eng = hip_engine
    g = torch.Generator().manual_seed(V + H + B)
    W = (0.02 * torch.randn((V, H), generator=g)).to(eng.device)
    Wd = eng.alloc_matrix(V, H); Wd.copy_(W)
    hb = (0.1 * torch.randn(H, generator=g)).to(eng.device)
    vb = (0.1 * torch.randn(V, generator=g)).to(eng.device)
    x = torch.randn((B, V), generator=g).to(eng.device)
    h = (torch.rand((B, H), generator=g) < 0.5).float().to(eng.device)
    from mdbn_amd.engine import RngAddr
    pre, mean, sample = eng.propup(x, Wd, hb, rng=RngAddr(3, 0, 1, 0))
    ref = x.double() @ W.double() + hb.double()
    tol = 4 * gtol(V) * max(1.0, float(ref.abs().max()))
    assert float((pre[:, :H].double() - ref).abs().max()) <= tol
    assert float((mean[:, :H] - torch.sigmoid(pre[:, :H])).abs().max()) <= 2e-6
    u = eng.rng_uniform(B, H, RngAddr(3, 0, 1, 0))
    expect = (u[:, :H] < mean[:, :H]).float()
    assert float((sample[:, :H] != expect).float().mean()) == 0.0
    dpre, dmean, dsample = eng.propdown(h, Wd, vb, gauss=False, rng=RngAddr(3, 0, 1, 1))
    dref = h.double() @ W.double().t() + vb.double()
    dtol = 4 * gtol(H) * max(1.0, float(dref.abs().max()))
    assert float((dpre[:, :V].double() - dref).abs().max()) <= dtol
    assert float((dmean[:, :V] - torch.sigmoid(dpre[:, :V])).abs().max()) <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("V,H", [(130, 70), (4096, 1024)])
@pytest.mark.parametrize("wc", [0.0, 2e-4])
def test_update_phase3_is_phase1_then_phase2(hip_engine, V, H, wc):
    """mdbn_update_args.phase = 3 (the overlapped data-parallel step: speeds of step t-1 from the reduced
    statistics, then the parameters of step t from those NEW speeds, one pass) must equal phase 1
    followed by phase 2 bit for bit."""
    eng = hip_engine
    rs = np.random.RandomState(V + H)
    def fresh():
        W = eng.to_device((0.05 * rs.randn(V, H)).astype(np.float32))
        return W
    W_init = (0.05 * rs.randn(V, H)).astype(np.float32)
    Ws_init = (0.01 * rs.randn(V, H)).astype(np.float32)
    hb_i, hbs_i = (0.1 * rs.randn(H)).astype(np.float32), (0.01 * rs.randn(H)).astype(np.float32)
    vb_i, vbs_i = (0.1 * rs.randn(V)).astype(np.float32), (0.01 * rs.randn(V)).astype(np.float32)
    ldv, ldh = (V + 3) // 4 * 4, (H + 3) // 4 * 4
    stats_np = np.zeros(V * ldh + ldh + ldv + 4, np.float32)
    stats_np[:V * ldh].reshape(V, ldh)[:, :H] = rs.randn(V, H) * 30
    stats_np[V * ldh:V * ldh + H] = rs.randn(H) * 10
    stats_np[V * ldh + ldh:V * ldh + ldh + V] = rs.randn(V) * 10
    stats_np[V * ldh + ldh + ldv] = 123.0
    out = {}
    for mode in ("split", "fused"):
        W, Ws = eng.to_device(W_init), eng.to_device(Ws_init)
        W0 = eng.to_device(W_init) if wc else None
        hb, hbs, vb, vbs = [eng.to_device(a) for a in (hb_i, hbs_i, vb_i, vbs_i)]
        stats = eng.to_device(stats_np)
        args = (W, Ws, W0, hb, hbs, vb, vbs, stats, 0.01, 0.0, 0.1, wc, 0.9, 512.0, 500.0, 0.5)
        if mode == "split":
            c = eng.apply_update(*args, phase=1, ldv=ldv)
            eng.apply_update(*args, phase=2, ldv=ldv)
        else:
            c = eng.apply_update(*args, phase=3, ldv=ldv)
        out[mode] = [t.cpu().numpy() for t in (W, Ws, hb, hbs, vb, vbs)] + [np.float32(float(c))]
    for a, b in zip(out["split"], out["fused"]):
        assert np.array_equal(a, b)
    assert not np.array_equal(out["fused"][0], W_init)


@pytest.mark.gpu
@pytest.mark.parametrize("V,H,B", [(4096, 1024, 512), (2048, 400, 512), (1031, 4097, 777)])
def test_bf16x6_gemm_is_f32_grade(hip_engine, V, H, B):
    """The default GEMM path (gemm_bf16x6_kernel: exact three-way bf16 split of the f32 operands, six
    products on the bf16 MFMA, f32 accumulation) against float64, next to the exact-f32 MFMA kernel
    on the same inputs: pre-activations of both passes and the CD statistics.  The bar is the f32
    tolerance of the other parity tests; the split path may not be more than 4x worse than the exact
    kernel's own error (measured: about equal; rocBLAS sgemm sits at the same level)."""
    import torch
    eng = hip_engine
    g = torch.Generator().manual_seed(V + 3 * H + B)
    W = (0.03 * torch.randn((V, H), generator=g)).to(eng.device)
    Wd = eng.alloc_matrix(V, H); Wd.copy_(W)
    hb = (0.1 * torch.randn(H, generator=g)).to(eng.device)
    vb = (0.1 * torch.randn(V, generator=g)).to(eng.device)
    x = torch.randn((B, V), generator=g).to(eng.device)
    h = torch.rand((B, H), generator=g).to(eng.device)                  # probabilities: three pieces
    ref_up = x.double() @ W.double() + hb.double()
    ref_dn = h.double() @ W.double().t() + vb.double()
    V2 = torch.randn((2 * B, V), generator=g).to(eng.device)
    P2 = torch.rand((2 * B, H), generator=g).to(eng.device)
    ref_S = V2.double().t() @ P2.double()
    err = {}
    try:
        for mode in (3, 0):
            eng.set_option("gemm_bf16x6", mode)
            up = eng.propup(x, Wd, hb, want_mean=False, want_sample=False)[0][:, :H].double()
            dn = eng.propdown(h, Wd, vb, gauss=True)[0][:, :V].double()
            ldv, ldh = (V + 3) // 4 * 4, (H + 3) // 4 * 4
            V2d, P2d = eng.alloc_matrix(2 * B, V, ldv), eng.alloc_matrix(2 * B, H, ldh)
            V2d.copy_(V2); P2d.copy_(P2)
            stats = torch.zeros(V * ldh + ldh + ldv + 4, device=eng.device)
            ws = eng.workspace(B, V, H)
            from mdbn_amd import _lib
            _lib.check(eng.lib.mdbn_cd_stats(eng.ctx, eng._stream(), eng._p(V2d), eng._p(P2d), B, V, H, ldv, ldh,
                                             eng._p(stats), eng._p(ws), ws.numel() * 4), "mdbn_cd_stats")
            S = stats[:V * ldh].reshape(V, ldh)[:, :H].double()
            err[mode] = [float((up - ref_up).abs().max() / ref_up.abs().max()),
                         float((dn - ref_dn).abs().max() / ref_dn.abs().max()),
                         float((S - ref_S).abs().max() / ref_S.abs().max())]
    finally:
        eng.set_option("gemm_bf16x6", 3)
    for e6, e0, K in zip(err[3], err[0], (V, H, 2 * B)):
        assert e6 <= 4 * gtol(K)                       # the f32 bar
        assert e6 <= 4 * e0 + 1e-7                     # and not materially worse than the exact-f32 MFMA kernel


@pytest.mark.gpu
def test_bf16x6_split_is_exact(hip_engine):
    """Multiplying by an identity matrix through the bf16x6 GEMM must return the other operand BIT FOR BIT:
    that holds only if the in-kernel three-way bf16 split reproduces every f32 value exactly (x = x1 + x2 +
    x3) and the MFMA accumulation adds the pieces without loss -- for values across many magnitudes."""
    import torch
    eng = hip_engine
    n, B = 1024, 512
    g = torch.Generator().manual_seed(11)
    eye = torch.eye(n)
    Wd = eng.alloc_matrix(n, n); Wd.copy_(eye.to(eng.device))
    zero = eng.alloc_vector(n)
    # values over ~30 binades, both signs, plus exact zeros and powers of two
    vals = torch.randn((B, n), generator=g) * torch.exp2(torch.randint(-20, 10, (B, n), generator=g).float())
    vals[0, :8] = torch.tensor([0.0, 1.0, -1.0, 2.0 ** -20, 3.0, 1.0 + 2.0 ** -23, -(1.0 + 2.0 ** -12), 255.99998])
    h = vals.to(eng.device)
    try:
        eng.set_option("gemm_bf16x6", 3)
        pre_dn = eng.propdown(h, Wd, zero, gauss=True)[0]           # A operand (K-contiguous) split
        assert torch.equal(pre_dn[:, :n], h)
        W2 = eng.alloc_matrix(n, n); W2.copy_(vals[:n // 2].repeat(2, 1).to(eng.device))
        x = eng.alloc_matrix(B, n); x.copy_(eye[:B].to(eng.device))
        pre_up = eng.propup(x, W2, zero, want_mean=False, want_sample=False)[0]   # B operand (row-contiguous) split
        assert torch.equal(pre_up[:, :n], W2[:B, :n])
    finally:
        eng.set_option("gemm_bf16x6", 3)


@pytest.mark.gpu
def test_bf16x6_split_value_range_and_non_finite(hip_engine, planes=0):
    """The whole f32 range through an identity multiply on the in-kernel split of gemm_bf16x6_kernel (the plane path's
    split, the same truncation, is covered in test_gpu_planes.py): exponents -110 ... 127 come back BIT FOR BIT; below 2^-110 the lowest significand bits weigh less
    than the smallest bf16 subnormal (2^-133) and are truncated -- error < 2^-132 absolute; a +-Inf or NaN operand
    makes its output row/column NaN (Inf - Inf in the remainder), never a finite wrong number."""
    import torch
    eng = hip_engine
    n, B = 1024, 512
    eng.set_option("gemm_planes", planes)
    try:
        g = torch.Generator().manual_seed(5)
        mant = 1.0 + torch.rand((B, n), generator=g)
        expo = torch.randint(-110, 128, (B, n), generator=g).float()
        vals = mant * torch.exp2(expo) * (torch.randint(0, 2, (B, n), generator=g).float() * 2 - 1)
        tiny = (1.0 + torch.rand((B, n), generator=g)) * torch.exp2(torch.randint(-126, -110, (B, n), generator=g).float())
        Wd = eng.alloc_matrix(n, n); Wd.copy_(torch.eye(n).to(eng.device))
        zero = eng.alloc_vector(n)
        for src, exact in ((vals, True), (tiny, False)):
            h = eng.alloc_matrix(B, n); h.copy_(src.to(eng.device))
            out = eng.propdown(h, Wd, zero, gauss=True)[0][:, :n]
            if exact:
                assert torch.equal(out, h)
            else:
                assert float((out.double() - h.double()).abs().max()) < 2.0 ** -132
        bad = vals.clone().clamp(-1e3, 1e3)
        bad[3, 7], bad[5, 9], bad[8, 11] = float("inf"), float("-inf"), float("nan")
        h = eng.alloc_matrix(B, n); h.copy_(bad.to(eng.device))
        out = eng.propdown(h, Wd, zero, gauss=True)[0][:, :n].cpu()
        # identity weights: column j of the output only sees column j of h, times 0 elsewhere -- 0 * Inf = NaN spreads
        # along the ROW of the non-finite entry, nowhere else
        rows_bad = torch.isnan(out).any(dim=1) | torch.isinf(out).any(dim=1)
        assert rows_bad.nonzero().flatten().tolist() == [3, 5, 8]
        assert not torch.isfinite(out[3, 7]) and not torch.isfinite(out[5, 9]) and torch.isnan(out[8, 11])
    finally:
        eng.set_option("gemm_planes", 1)


@pytest.mark.gpu
def test_bf16x6_ragged_shapes_match_exact_kernel(hip_engine):
    """Random ragged shapes (edge tiles in both dimensions, K tails, K not a multiple of 4, split and
    unsplit K): the bf16x6 path against the exact-f32 kernel on both passes -- same results to the f32
    accumulation tolerance, pad columns exactly zero."""
    import torch
    eng = hip_engine
    rs = np.random.RandomState(20261004)
    shapes = [(int(rs.randint(130, 1500)), int(rs.randint(130, 1500)), int(rs.randint(65, 900))) for _ in range(14)]
    shapes += [(257, 1025, 129), (1281, 129, 640), (999, 1001, 1003)]
    for V, H, B in shapes:
        g = torch.Generator().manual_seed(V * 3 + H * 5 + B)
        Wd = eng.alloc_matrix(V, H); Wd.copy_((0.05 * torch.randn((V, H), generator=g)).to(eng.device))
        hb = (0.1 * torch.randn(H, generator=g)).to(eng.device)
        vb = (0.1 * torch.randn(V, generator=g)).to(eng.device)
        x = eng.alloc_matrix(B, V); x.copy_(torch.randn((B, V), generator=g).to(eng.device))
        h = eng.alloc_matrix(B, H); h.copy_(torch.rand((B, H), generator=g).to(eng.device))
        out = {}
        try:
            for mode in (3, 0):
                eng.set_option("gemm_bf16x6", mode)
                up = eng.propup(x, Wd, hb, want_mean=False, want_sample=False)[0]
                dn = eng.propdown(h, Wd, vb, gauss=True)[0]
                out[mode] = (up, dn)
        finally:
            eng.set_option("gemm_bf16x6", 3)
        for a, b, K in zip(out[3], out[0], (V, H)):
            scale = max(1.0, float(b.abs().max()))
            assert float((a - b).abs().max()) <= 4 * gtol(K) * scale, (V, H, B)
            base_a = a._base if a._base is not None else a
            if base_a.shape[1] > a.shape[1]:
                assert float(base_a[:, a.shape[1]:].abs().max()) == 0.0, "pad columns must stay zero"


def test_trainer_loop_resume_is_exact_on_device(hip_engine, tmp_path):
    """f2 "epoch": DBN.training interrupted after a mid-epoch step of its second layer, written with
    save_network(resume=True), reloaded into a fresh DBN and resumed -- records, parameters, update and RNG counters equal the
    uninterrupted run's bit for bit (the device state is float32, as the checkpoint)."""
    from test_plumbing import _interrupted_vs_straight
    _interrupted_vs_straight(hip_engine, tmp_path, exact=True)
