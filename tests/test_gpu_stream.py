"""The streaming bf16x6 GEMM (csrc/mdbn_stream.hip: mid-size layers at B > 64 -- one UNSPLIT launch per pass on 32 x 32 /
64 x 32 / 64 x 64 tiles, f32 operands split in registers, activation / update epilogue on the tile) against the float64
oracle, teacher-forced along the device's own chain, at the layer sizes of BASELINE configs 4 / 5 (reference:
AMLsm2.py:242-340, MDBN.py:31-35) and at ragged shapes; the arithmetic is that of rbm.py:303 (positive phase), :242-248 /
:662-671 (gibbs_hvh), :392-419 (statistics) and :347-365 (update)."""
import numpy as np
import pytest

from oracle import rbm_np
from oracle.philox_np import PhiloxDraws
from _margins import check

pytestmark = pytest.mark.gpu

SHAPES = [  # V, H, B, k, gauss, through an index list, stream_x6 (2: the small-layer passes too), (mi, ni) override or None
    (2048, 400, 512, 5, True, True, 1, None),      # c5 GE first layer, CD-5
    (1024, 256, 512, 1, False, True, 1, None),     # c4 second layer
    (256, 200, 512, 5, True, True, 2, None),       # c5 SM first layer: a small-layer shape, served under stream_x6 = 2
    (784, 500, 512, 1, False, True, 1, None),      # MNIST layer at the benchmark batch
    (1000, 300, 200, 2, False, False, 2, None),    # K tails in both directions (1000 = 62.5 steps, 300 = 18.75), ragged row tile
    (530, 77, 97, 1, True, True, 2, (2, 2)),       # nothing a multiple of anything, 64 x 64 tiles forced: half-empty last tiles
    (530, 77, 97, 3, False, True, 2, (2, 1)),      # ... 64 x 32 tiles, Bernoulli chain (0/1 row operand: three products)
    (96, 640, 130, 1, True, False, 2, (1, 1)),     # wide hidden layer: K = 96 (6 steps: fewer than the waves of a tile)
]


def _step(eng, V, H, B, k, gauss, indexed, mode, tiles, seed):
    from mdbn_amd import RngAddr
    rs = np.random.RandomState(seed)
    W = rbm_np.init_W(rs, V, H, np.float32)
    hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
    N = B + 13 if indexed else B
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    idx = None
    if indexed:
        idx = rs.permutation(N)[:B].astype(np.int64)
        idx[::7] -= N
    dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, hb, vb, data)]
    eng.set_option("small_fused", 0)
    eng.set_option("stream_x6", mode)
    if tiles:
        eng.set_option("stream_mi", tiles[0]); eng.set_option("stream_ni", tiles[1])
    eng.set_planes_min_work(1 << 30)          # the product rule (the test fixture serves every whole-tile shape on planes)
    eng.trace_chain = True
    eng.kernel_timing(True)
    try:
        stats, sc = eng.cd_step(dx, idx, dW, dhb, dvb, gauss, k, RngAddr(5, 3, 11, 0, 0))
        eng.synchronize()
        kinds = [kd for _, _, _, kd in eng.kernel_timing_detail()]
        th = sc.trace_h.cpu().numpy()[:, :, :H]
        tv = None if gauss else sc.trace_v.cpu().numpy()[:, :, :V]
    finally:
        eng.kernel_timing(False)
        eng.trace_chain = False
        eng.set_option("small_fused", 1)
        eng.set_option("stream_x6", 2)
        eng.set_option("stream_mi", 0); eng.set_option("stream_ni", 0)
        eng.set_planes_min_work(0)
    x = data[idx] if idx is not None else data
    return dict(W=W, hb=hb, vb=vb, x=x, stats=stats.cpu().numpy(), sc=sc, th=th, tv=tv, kinds=kinds)


@pytest.mark.parametrize("V,H,B,k,gauss,indexed,mode,tiles", SHAPES)
def test_stream_step_against_forced_oracle(hip_engine, V, H, B, k, gauss, indexed, mode, tiles):
    r = _step(hip_engine, V, H, B, k, gauss, indexed, mode, tiles, seed=V + H + k)
    # every GEMM launch of the step: the streaming kernel (1000) on the bf16 pipe (100 = six products, 200 = three)
    assert len(r["kinds"]) == 2 * k + 2 and all(kd >= 1100 for kd in r["kinds"]), "not the streaming bf16x6 path: %r" % r["kinds"]
    if not gauss:
        assert sum(1 for kd in r["kinds"] if kd >= 1200) == 2 * k, "0/1 row operands must take three products: %r" % r["kinds"]
    st = rbm_np.RBMState(V, H, W=r["W"], hbias=r["hb"], vbias=r["vb"], gauss=gauss)
    v0 = r["x"].astype(np.float64)
    ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(5, 3, 11, 0), k, r["th"], r["tv"])
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
    sc = r["sc"]
    ldh, ldv = sc.P2.stride(0), sc.V2.stride(0)
    d = r["stats"]
    S, s_h, s_v = d[:V * ldh].reshape(V, ldh), d[V * ldh:V * ldh + H], d[V * ldh + ldh:V * ldh + ldh + V]
    cost = d[V * ldh + ldh + ldv]
    tag = "stream CD-%d %d->%d B=%d %s" % (k, V, H, B, "GRBM" if gauss else "RBM")
    assert not S[:, H:].any(), "pad columns of S must stay zero"
    check(tag + ": S / max|S|", np.abs(S[:, :H] - S_o).max() / max(1.0, np.abs(S_o).max()), 1e-5, "stats")
    check(tag + ": s_h / max", np.abs(s_h - s_h_o).max() / max(1.0, np.abs(s_h_o).max()), 1e-5, "stats")
    check(tag + ": s_v / max", np.abs(s_v - s_v_o).max() / max(1.0, np.abs(s_v_o).max()), 1e-5, "stats")
    check(tag + ": ph_mean", np.abs(sc.P2[:B].cpu().numpy()[:, :H] - ph).max(), 2e-6, "prob")
    check(tag + ": nh_mean", np.abs(-sc.P2[B:2 * B].cpu().numpy()[:, :H] - out[4]).max(), 4e-6, "prob")
    check(tag + ": nv_mean / max|nv|", np.abs(sc.V2[B:2 * B].cpu().numpy()[:, :V] - out[1]).max() / max(1.0, np.abs(out[1]).max()),
          2e-6, "nv_mean")
    np.testing.assert_array_equal(sc.V2[:B].cpu().numpy()[:, :V], r["x"])
    assert not sc.P2.cpu().numpy()[:, H:].any() and not sc.V2.cpu().numpy()[:, V:].any(), "pad columns must stay zero"
    pre = out[0]
    if gauss:
        want = ((rbm_np.sigmoid(pre) - v0) ** 2).sum()
    else:
        want = (v0 * rbm_np.softplus(-pre) + (1 - v0) * rbm_np.softplus(pre)).sum()
    check(tag + ": cost sum rel", abs(cost - want) / abs(want), 2e-6)
    assert flips <= 3


@pytest.mark.parametrize("V,H,B,k,gauss,indexed,mode,tiles", [SHAPES[1], SHAPES[4], SHAPES[5]])
def test_stream_step_draws_the_uniforms_of_the_tiled_path(hip_engine, V, H, B, k, gauss, indexed, mode, tiles):
    """Same Philox addressing as the LDS-tiled split-K path (stream_x6 = 0): the positive-phase samples agree except within
    rounding of a uniform, the statistics to fp32 summation order."""
    a = _step(hip_engine, V, H, B, k, gauss, indexed, mode, tiles, seed=3)
    b = _step(hip_engine, V, H, B, k, gauss, indexed, 0, None, seed=3)
    assert all(kd >= 1100 for kd in a["kinds"]) and not any(kd >= 1100 for kd in b["kinds"]), (a["kinds"], b["kinds"])
    differ = int((a["th"][0] != b["th"][0]).sum())
    assert differ <= 2, differ
    if differ == 0 and k == 1:
        scale = max(1.0, np.abs(b["stats"]).max())
        assert np.abs(a["stats"] - b["stats"]).max() <= 2e-5 * scale


def test_stream_step_repeats_bit_for_bit(hip_engine):
    """The 8 partial tiles are reduced in wave order: the step is deterministic from run to run."""
    a = _step(hip_engine, 2048, 400, 512, 2, True, True, 1, None, seed=9)
    b = _step(hip_engine, 2048, 400, 512, 2, True, True, 1, None, seed=9)
    assert np.array_equal(a["stats"], b["stats"]) and np.array_equal(a["th"], b["th"])


@pytest.mark.parametrize("V,H,B,cls_gauss,hp", [(2048, 400, 512, True, dict(lr=0.005, lambda_2=0.1)),
                                                (1024, 256, 512, False, dict(lr=0.1, weightcost=2e-4)),
                                                (530, 77, 97, True, dict(lr=0.002, lambda_1=0.01, lambda_2=0.05))],
                         ids=["ge_grbm_l2", "c4_rbm_weightcost", "ragged_grbm_l1_l2"])
def test_stream_training_steps_follow_the_oracle(hip_engine, V, H, B, cls_gauss, hp):
    """Four calls of the compiled step function (rbm.py:258-376) through the RBM classes: parameters, speeds and costs
    against the float64 oracle that follows the device's recorded chain -- the update applied to each tile of S inside the
    statistics launch of the streaming kernel."""
    import mdbn_amd
    eng = hip_engine
    N = 2 * B + 5
    rs = np.random.RandomState(7)
    data = rs.normal(size=(N, V)).astype(np.float32) if cls_gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    cls = mdbn_amd.GRBM if cls_gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(7), engine=eng)
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), gauss=cls_gauss)
    if hp.get("weightcost"):
        st.freeze_W0()
    eng.set_planes_min_work(1 << 30)
    eng.set_option("stream_x6", 2)            # (the ragged shape's passes are small-layer passes: exact-f32 kernel under 1)
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
            check("stream training: cost rel", abs(cost - want) / abs(want), 1e-5)
        eng.synchronize()
        kinds = [kd for _, _, _, kd in eng.kernel_timing_detail()]
        assert kinds and all(kd >= 1100 for kd in kinds), "not the streaming path: %r" % kinds
        assert any(kd % 100 == 23 for kd in kinds), "the update was not fused into the statistics launch: %r" % kinds
    finally:
        eng.kernel_timing(False)
        eng.trace_chain = False
        eng.set_planes_min_work(0)
        eng.set_option("stream_x6", 2)
    for name in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed"):
        got, ref = getattr(rbm, name).get_value(), getattr(st, name)
        check("stream training: %s after 4 steps / max" % name, np.abs(got - ref).max() / max(1.0, np.abs(ref).max()), 2e-6, "update")

