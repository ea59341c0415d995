"""The group-chain CD-k step (csrc/mdbn_gchain.hip: positive phase + whole Gibbs chain of a mid-size layer in one launch, W held
in the LDS of a group of workgroups, one in-launch exchange per pass) against the float64 oracle, teacher-forced along the
device's own chain, and beside the multi-launch path it replaces.  Reference: the scan of rbm.py:318-336 over gibbs_hvh
(:242-248, :662-671) at the layer sizes of AMLsm2.py:242-340 / MDBN.py:31-35."""
import numpy as np
import pytest

from oracle import rbm_np
from oracle.philox_np import PhiloxDraws
from _margins import check

pytestmark = pytest.mark.gpu

SHAPES = [  # V, H, B, k, gauss, through an index list
    (256, 200, 512, 5, True, True),       # c5 SM first layer: groups of 2
    (1024, 256, 512, 1, False, True),     # c4 second layer: groups of 16
    (256, 200, 100, 2, False, False),     # ragged last slab (4 rows), Bernoulli chain
    (512, 300, 64, 1, True, True),        # groups of 8, two slabs
    (300, 130, 40, 3, False, True),       # nothing a multiple of 16; last member short of rows
    (1000, 72, 1100, 1, True, False),     # more slabs than groups: a group loops
]


def _step(eng, V, H, B, k, gauss, indexed, seed, on):
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
    eng.set_option("gchain", int(on))
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
        eng.set_option("gchain", 0)               # (the library default: opt-in, see csrc/mdbn_capi.hip)
        eng.set_planes_min_work(0)
    x = data[idx] if idx is not None else data
    return dict(W=W, hb=hb, vb=vb, x=x, stats=stats.cpu().numpy(), sc=sc, th=th, tv=tv, kinds=kinds)


@pytest.mark.parametrize("V,H,B,k,gauss,indexed", SHAPES)
def test_group_chain_step_against_forced_oracle(hip_engine, V, H, B, k, gauss, indexed):
    r = _step(hip_engine, V, H, B, k, gauss, indexed, seed=V + H + k, on=True)
    # only the statistics GEMM may have been launched as a GEMM (kind % 10 = 2 la + lb = 3): no forward pass was
    assert r["kinds"] and all(kd % 10 == 3 for kd in r["kinds"]), "forward GEMM launches: not the group-chain path: %r" % r["kinds"]
    st = rbm_np.RBMState(V, H, W=r["W"], hbias=r["hb"], vbias=r["vb"], gauss=gauss)
    v0 = r["x"].astype(np.float64)
    ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(5, 3, 11, 0), k, r["th"], r["tv"])
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
    sc = r["sc"]
    ldh, ldv = sc.P2.stride(0), sc.V2.stride(0)
    d = r["stats"]
    S, s_h, s_v = d[:V * ldh].reshape(V, ldh), d[V * ldh:V * ldh + H], d[V * ldh + ldh:V * ldh + ldh + V]
    cost = d[V * ldh + ldh + ldv]
    tag = "group chain CD-%d %d->%d B=%d %s" % (k, V, H, B, "GRBM" if gauss else "RBM")
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


@pytest.mark.parametrize("V,H,B,k,gauss,indexed", SHAPES[:3])
def test_group_chain_draws_the_uniforms_of_the_multi_launch_path(hip_engine, V, H, B, k, gauss, indexed):
    a = _step(hip_engine, V, H, B, k, gauss, indexed, seed=3, on=True)
    b = _step(hip_engine, V, H, B, k, gauss, indexed, seed=3, on=False)
    assert any(kd % 10 != 3 for kd in b["kinds"]), "the comparison path took no forward GEMM launches"
    differ = int((a["th"][0] != b["th"][0]).sum())
    assert differ <= 2, differ
    if differ == 0 and k == 1:
        scale = max(1.0, np.abs(b["stats"]).max())
        assert np.abs(a["stats"] - b["stats"]).max() <= 2e-5 * scale


def test_group_chain_repeats_bit_for_bit_and_reports_no_stalled_exchange(hip_engine):
    """The exchange sums the members' partials in member order on every member: the step is deterministic from run to run."""
    V, H, B, k = 256, 200, 512, 5
    a = _step(hip_engine, V, H, B, k, True, True, seed=9, on=True)
    b = _step(hip_engine, V, H, B, k, True, True, seed=9, on=True)
    assert np.array_equal(a["stats"], b["stats"]) and np.array_equal(a["th"], b["th"])
