"""The bf16 plane path (csrc/mdbn_planes.hip): every tensor of the CD step split once into three bf16 planes,
GEMMs fed by LDS-DMA with row / transposed LDS reads.  Checked: the split is exact (and what it does with
non-finite values), the plane step equals the f32-operand step BIT FOR BIT (same products, same order), the
planes of W stay in step with W through every update path, and the oracle comparison at the c2 / c4 shapes."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import rbm_np
from oracle.philox_np import PhiloxDraws

pytestmark = pytest.mark.gpu


def planes_to_f32(P):
    """[3, rows, ld] int16 planes -> the three f32 pieces."""
    return [((P[i].to(torch.int32) & 0xffff) << 16).view(torch.float32) for i in range(3)]


def split(eng, x):
    from mdbn_amd import _lib
    x = eng.as_matrix(x)
    rows, ld = x.shape[0], x.stride(0)
    P = torch.empty((3, rows, ld), dtype=torch.int16, device=eng.device)
    base = x._base if x._base is not None else x
    _lib.check(eng.lib.mdbn_split_planes(eng.ctx, eng._stream(), C.c_void_p(base.data_ptr()), rows, ld,
                                         C.c_void_p(P.data_ptr())), "mdbn_split_planes")
    return P


def test_split_is_exact_over_the_whole_exponent_range(hip_engine):
    """x = p1 + p2 + p3 bit for bit for every |x| >= 2^-110 (exponents -110 ... 127, random 24-bit significands, both
    signs) and for +-0.  Below 2^-110 the lowest of the 24 significand bits weighs less than 2^-133, the smallest bf16
    subnormal, and is truncated: |x - (p1 + p2 + p3)| < 2^-133 (about 1e-40) for tiny normals and f32 subnormals --
    an absolute error no sum of products can notice."""
    rs = np.random.RandomState(0)
    mant = rs.uniform(1, 2, (254, 64)).astype(np.float32)
    expo = np.arange(-126, 128, dtype=np.float32)[:, None]
    x = (mant * np.exp2(expo) * rs.choice([-1.0, 1.0], (254, 64))).astype(np.float32)
    denorm = (rs.randint(1, 1 << 23, (2, 64)).astype(np.uint32)).view(np.float32)        # f32 subnormals
    x = np.concatenate([x, denorm, np.zeros((1, 64), np.float32), -np.zeros((1, 64), np.float32)])
    xd = hip_engine.to_device(x)
    p1, p2, p3 = planes_to_f32(split(hip_engine, xd))
    back = ((p3 + p2) + p1).cpu().numpy()       # the residuals are exact f32 values: two exact additions
    exact_rows = np.concatenate([expo[:, 0] >= -110, [False, False, False, False]])
    assert np.array_equal(back[exact_rows].view(np.uint32), x[exact_rows].view(np.uint32)), "x != p1 + p2 + p3"
    assert not back[-2:].any() and np.signbit(p1.cpu().numpy()[-1]).all()      # +-0: p1 keeps the sign, the sum is a zero
    err = np.abs(back.astype(np.float64) - x.astype(np.float64))
    assert err[~exact_rows][:-2].max() < 2.0 ** -133
    # the pieces shrink by at least 2^-7 each (bf16: 8 significant bits)
    a1, a2, a3 = [t.abs().cpu().numpy().astype(np.float64) for t in (p1, p2, p3)]
    assert np.all(a2 <= a1 * 2.0 ** -7) and np.all(a3 <= a1 * 2.0 ** -15)


def test_split_of_non_finite_values(hip_engine):
    """+-Inf: p1 = +-Inf and the remainder Inf - Inf = NaN, so a product with such an operand comes out NaN where
    an f32 GEMM would give +-Inf (or NaN); NaN stays NaN.  Non-finite operands are therefore never silently
    turned into finite results; the NaN guard (StepFunction.nan_guard) is the tool to locate them."""
    x = np.array([[np.inf, -np.inf, np.nan, 1.0]], dtype=np.float32)
    p1, p2, p3 = [t.cpu().numpy()[0] for t in planes_to_f32(split(hip_engine, hip_engine.to_device(x)))]
    assert p1[0] == np.inf and p1[1] == -np.inf and np.isnan(p1[2]) and p1[3] == 1.0
    assert np.isnan(p2[0]) and np.isnan(p2[1]) and np.isnan(p2[2]) and p2[3] == 0.0 and p3[3] == 0.0


def _run_steps(eng, gauss, planes, V, H, B, k, steps=3, seed=0, mfma=16):
    import mdbn_amd
    eng.set_option("gemm_planes", int(planes))
    eng.set_option("planes_mfma", mfma)
    eng.set_option("stream_x6", 0)      # (the f32-operand side of these comparisons: the LDS-tiled kernels whose plans the planes follow)
    if planes:          # a shape the library does not serve on planes would compare the f32-operand path with itself
        assert eng.plane_shape(B, V, H, V, H), "test shape is not on the plane path: %r" % ((B, V, H),)
    keep, eng.keep_f32 = eng.keep_f32, bool(seed & 1)        # the product default (no float32 copies) on even seeds
    rs = np.random.RandomState(seed)
    N = 4 * B
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5), engine=eng)
    hp = dict(lr=0.001, lambda_2=0.1) if gauss else dict(lr=0.05, weightcost=2e-4)
    _, up = rbm.get_cost_updates(k=k, batch_size=B, **hp)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel=None)
    costs = [float(fn(indexes=rs.permutation(N)[:B], momentum=0.5)) for _ in range(steps)]
    out = dict(costs=np.array(costs), W=rbm.W.get_value(), Ws=rbm.W_speed.get_value(), hb=rbm.hbias.get_value(),
               vbs=rbm.vbias_speed.get_value())
    wp, valid = eng.w_planes(rbm.W.tensor)
    eng.set_option("gemm_planes", 1)
    eng.set_option("planes_mfma", 16)
    eng.set_option("stream_x6", 2)
    eng.keep_f32 = keep
    return out, rbm, wp, valid


@pytest.mark.parametrize("gauss,V,H,B,k", [(True, 4096, 1024, 512, 1), (False, 1024, 512, 512, 1), (True, 1024, 768, 256, 3),
                                           (False, 2048, 1024, 256, 2)])
def test_plane_step_equals_f32_operand_step_bit_for_bit(hip_engine, gauss, V, H, B, k):
    """With the 32x32x16 MFMA shape the plane GEMMs issue the products of gemm_bf16x6_kernel in the same order and follow
    the same split-K plans: every parameter, speed and sample equals the f32-operand step bit for bit.  The default
    16x16x32 shape sums each 32-deep stage in one MFMA instead of two: same products, another fp32 summation order."""
    a, _, _, _ = _run_steps(hip_engine, gauss, False, V, H, B, k)
    b, rbm, wp, valid = _run_steps(hip_engine, gauss, True, V, H, B, k, mfma=32)
    c, _, _, _ = _run_steps(hip_engine, gauss, True, V, H, B, k, mfma=16)
    for key in ("W", "Ws", "hb", "vbs"):
        scale = max(1e-3, np.abs(a[key]).max())
        assert np.abs(c[key] - a[key]).max() <= 2e-5 * scale, key
    np.testing.assert_allclose(c["costs"], a["costs"], rtol=1e-4)
    for key in a:
        if key == "costs":       # the monitoring cost is summed from per-tile instead of per-epilogue-block partials
            np.testing.assert_allclose(a[key], b[key], rtol=2e-6)
        else:
            assert np.array_equal(a[key], b[key]), key
    # the planes of W followed every update
    assert wp is not None and valid
    p1, p2, p3 = planes_to_f32(wp)
    assert torch.equal((p3 + p2) + p1, rbm.W.tensor._base if rbm.W.tensor._base is not None else rbm.W.tensor)


def test_w_planes_follow_every_way_w_changes(hip_engine):
    """set_value (a torch-side write) invalidates the planes and the next step re-splits; the data-parallel update
    phases (mdbn_apply_update 0 / 2 / 3) rewrite them; an in-library update on the f32 path does too."""
    import mdbn_amd
    eng = hip_engine
    V, H, B = 1024, 512, 256
    out, rbm, wp, valid = _run_steps(eng, True, True, V, H, B, 1, steps=2)
    assert valid
    rbm.W.set_value(rbm.W.get_value() * 0.5)
    assert not eng.w_planes(rbm.W.tensor)[1]
    data = mdbn_amd.shared(np.random.RandomState(1).normal(size=(2 * B, V)).astype(np.float32), engine=eng)
    _, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
    fn = mdbn_amd.function(up, data, data_parallel=None)
    fn(indexes=np.arange(B), momentum=0.0)

    def in_step():
        wp, valid = eng.w_planes(rbm.W.tensor)
        p1, p2, p3 = planes_to_f32(wp)
        return valid and torch.equal((p3 + p2) + p1, rbm.W.tensor)
    assert in_step()
    # the statistics + separate update path (what a data-parallel rank runs), every phase that writes W
    from mdbn_amd import RngAddr
    stats, _ = eng.cd_step(data.tensor, torch.arange(B, device=eng.device), rbm.W.tensor, rbm.hbias.tensor, rbm.vbias.tensor,
                           True, 1, RngAddr(1, 0, 7, 0, 0))
    for phase in (0, 2, 1, 3):
        eng.apply_update(rbm.W.tensor, rbm.W_speed.tensor, None, rbm.hbias.tensor, rbm.hbias_speed.tensor, rbm.vbias.tensor,
                         rbm.vbias_speed.tensor, stats, 0.001, 0.0, 0.1, 0.0, 0.5, B, B, 1.0, phase=phase, ldv=V)
        assert in_step(), phase
    # f32-operand path with planes present: the fused update still rewrites them
    eng.set_option("gemm_planes", 0)
    fn(indexes=np.arange(B) + B, momentum=0.0)
    eng.set_option("gemm_planes", 1)
    assert in_step()


@pytest.mark.parametrize("V,H,B,k,narrow", [(1024, 512, 256, 2, None), (4096, 1024, 512, 2, 1), (4096, 1024, 512, 2, 0)])
def test_plane_step_against_oracle_teacher_forced(hip_engine, V, H, B, k, narrow):
    """One CD-2 step of a Bernoulli RBM on the plane path with the chain taps on, oracle following the device.  At the
    headline shape propdown runs unsplit on 128 x 64 tiles with the activation on the parked tile (option
    "narrow_tiles", default on: the timing records must show that kernel) or split two ways + epilogue launch."""
    from mdbn_amd import RngAddr
    if narrow is not None:
        hip_engine.set_option("narrow_tiles", narrow)
        hip_engine.kernel_timing(True)
    rs = np.random.RandomState(3)
    W = rbm_np.init_W(rs, V, H, np.float32)
    hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
    x = (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
    eng = hip_engine
    dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, hb, vb, x)]
    eng.trace_chain = True
    try:
        stats, sc = eng.cd_step(dx, None, dW, dhb, dvb, False, k, RngAddr(11, 1, 2, 0, 0))
        assert sc.planes is not None
        th, tv = sc.trace_h.cpu().numpy(), sc.trace_v.cpu().numpy()
        if narrow is not None:
            eng.synchronize()
            kinds = [kind for _, _, _, kind in eng.kernel_timing_detail()]
            assert (kinds.count(2210) == k) == bool(narrow), kinds      # 2210: fused propdown (128 x 64 tiles)
    finally:
        eng.trace_chain = False
        if narrow is not None:
            eng.kernel_timing(False)
            eng.set_option("narrow_tiles", 1)
    st = rbm_np.RBMState(V, H, W=W, hbias=hb, vbias=vb, gauss=False)
    v0 = x.astype(np.float64)
    ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(11, 1, 2, 0), k, th, tv)
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
    d = stats.cpu().numpy()
    S, s_h, s_v = d[:V * H].reshape(V, H), d[V * H:V * H + H], d[V * H + H:V * H + H + V]
    assert np.abs(S - S_o).max() <= 1e-5 * np.abs(S_o).max()
    assert np.abs(s_h - s_h_o).max() <= 1e-5 * max(1.0, np.abs(s_h_o).max())
    assert np.abs(s_v - s_v_o).max() <= 1e-5 * max(1.0, np.abs(s_v_o).max())
    assert np.abs(sc.P2[:B].cpu().numpy() - ph).max() <= 2e-6
    assert flips <= 3


@pytest.mark.parametrize("gauss,V,H,B,k,comm_cus", [(True, 4096, 1024, 512, 1, 32), (True, 4096, 1024, 512, 1, 1),
                                                    (False, 1024, 512, 512, 2, 192), (True, 2048, 1024, 256, 1, 100),
                                                    # forward passes with MORE tiles than workgroups: whole tiles + shared rest
                                                    (False, 1024, 2048, 2048, 1, 32), (True, 2048, 2048, 1152, 1, 57)])
def test_balanced_launches_match_and_repeat(hip_engine, gauss, V, H, B, k, comm_cus):
    """mdbn_set_option("comm_cus", n): the plane GEMMs run on (CUs - n) workgroups that share tiles x stages evenly
    (data-parallel mode: the CUs left over belong to the collective).  A tile's stages are then summed in another
    grouping -- forward passes through per-segment slabs, the statistics GEMM through parked partials and per-wave
    flags -- so the step agrees with the one-workgroup-per-tile step to fp32 summation order, and with ITSELF bit for
    bit from run to run (a race in the flag protocol would show as run-to-run differences)."""
    eng = hip_engine
    ref, _, _, _ = _run_steps(eng, gauss, True, V, H, B, k, steps=3, seed=2)
    runs = []
    for _ in range(3):
        eng.set_option("comm_cus", comm_cus)
        try:
            out, rbm, wp, valid = _run_steps(eng, gauss, True, V, H, B, k, steps=3, seed=2)
        finally:
            eng.set_option("comm_cus", 0)
        runs.append(out)
    for key in ("W", "Ws", "hb", "vbs"):
        scale = max(1e-3, np.abs(ref[key]).max())
        assert np.isfinite(runs[0][key]).all(), key
        # fp32 summation order only; widest measured 2.3e-5 (hb of the 1024 x 2048 layer, whose reference step runs
        # propdown unsplit on 128 x 64 tiles while the balanced one sums per-segment slabs).  These are FREE-RUNNING
        # chains of up to 2.4 M Bernoulli draws per step: a draw whose uniform lies within the two launches' rounding
        # difference of its probability falls the other way in one of them (about one step in four at the largest shape)
        # and moves that row's whole chain -- a rank-2 change of S of relative size ~1 / B.  That case is told apart by
        # its signature: the RMS deviation stays at the summation-order level while single entries move up to 1e-3.
        dev = np.abs(runs[0][key] - ref[key])
        assert dev.max() <= 4e-5 * scale or (np.sqrt((dev.astype(np.float64) ** 2).mean()) <= 4e-5 * scale and
                                             dev.max() <= 1e-3 * scale), (key, dev.max(), scale)
    np.testing.assert_allclose(runs[0]["costs"], ref["costs"], rtol=1e-4)
    for other in runs[1:]:
        for key in runs[0]:
            assert np.array_equal(runs[0][key], other[key]), key
    p1, p2, p3 = planes_to_f32(wp)             # the planes of W followed the (unfused) update
    assert valid and torch.equal((p3 + p2) + p1, rbm.W.tensor._base if rbm.W.tensor._base is not None else rbm.W.tensor)


def test_bf16_input_reporting_mode(hip_engine):
    """mdbn_set_option("bf16_inputs", 1): one product on the leading bf16 pieces.  A reporting mode (BASELINE configs[1]
    names "bf16/fp32"), not a parity path: probabilities within 3e-2 of the oracle, far outside the 2e-6 of the default
    path -- and switching it off restores the f32-grade results exactly."""
    from mdbn_amd import RngAddr
    eng = hip_engine
    V, H, B = 1024, 512, 256
    rs = np.random.RandomState(4)
    W = rbm_np.init_W(rs, V, H, np.float32)
    hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
    x = rs.normal(size=(B, V)).astype(np.float32)
    dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, hb, vb, x)]
    st = rbm_np.RBMState(V, H, W=W, hbias=hb, vbias=vb, gauss=True)
    _, ph = rbm_np.propup(st, x.astype(np.float64))
    ref, _ = eng.cd_step(dx, None, dW, dhb, dvb, True, 1, RngAddr(2, 0, 0, 0, 0))
    ref_stats = ref.clone()
    sc = eng.last_scratch
    good = sc.P2[:B].cpu().numpy().copy()
    assert np.abs(good - ph).max() <= 2e-6
    eng.set_option("bf16_inputs", 1)
    try:
        eng.cd_step(dx, None, dW, dhb, dvb, True, 1, RngAddr(2, 0, 0, 0, 0))
        rough = eng.last_scratch.P2[:B].cpu().numpy().copy()
    finally:
        eng.set_option("bf16_inputs", 0)
    err = np.abs(rough - ph).max()
    assert 1e-5 < err <= 3e-2, err
    again, _ = eng.cd_step(dx, None, dW, dhb, dvb, True, 1, RngAddr(2, 0, 0, 0, 0))
    assert torch.equal(again, ref_stats) and np.array_equal(eng.last_scratch.P2[:B].cpu().numpy(), good)


def test_plane_path_serves_big_layers_only_by_default(built_lib):
    """Product default (no fixture overrides): whole-tile layers below B * V * H = 2^30 (or V * H < 2^21) stay on the
    f32-operand kernels, where they are faster (c4's 1024 -> 256: 69 vs 77 us); the headline shape takes the planes."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdbn_amd
    eng = mdbn_amd.HipEngine()
    try:
        eng.set_planes_min_work(1 << 30)
        assert eng.plane_shape(512, 4096, 1024, 4096, 1024)
        assert eng.plane_shape(512, 2048, 1024, 2048, 1024)
        assert not eng.plane_shape(512, 1024, 256, 1024, 256)          # c4 L1
        assert not eng.plane_shape(1024, 1024, 1024, 1024, 1024)       # V * H < 2^21
        assert not eng.plane_shape(512, 4096, 1000, 4096, 1000)        # not whole tiles
        assert eng.cd_scratch(512, 1024, 256, True, 1024, 256).planes is None
        assert eng.cd_scratch(512, 4096, 1024, False, 4096, 1024).planes is not None
    finally:
        eng.set_planes_min_work(0)          # what the other tests of this process expect from the library option


@pytest.mark.parametrize("gauss", [True, False])
def test_product_default_c2_statistics_against_forced_oracle(built_lib, gauss):
    """The path bench.py times, untouched: a fresh HipEngine (keep_f32 = 0: no float32 copies of ph / nh / nv / samples,
    cost target read through the minibatch index; default planes_min_work) at the c2 shape, mdbn_cd_step.  Its packed
    statistics S / s_h / s_v and cost are compared with the float64 oracle FOLLOWING the device's chain: without the
    float32 taps the positive-phase sample (and the visible sample of the Bernoulli RBM) is read from the step's own
    bf16 sample planes -- the very operands its next GEMM consumed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdbn_amd
    from mdbn_amd import RngAddr
    from _margins import check
    V, H, B, N = 4096, 1024, 512, 2048
    eng = mdbn_amd.HipEngine()
    assert eng.keep_f32 is False and eng.trace_chain is False
    try:
        eng.set_planes_min_work(1 << 30)                        # the library default (another test's fixture sets 0)
        rs = np.random.RandomState(5)
        W = rbm_np.init_W(rs, V, H, np.float32)
        hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
        data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
        idx = rs.permutation(N)[:B].astype(np.int64)
        dW, dhb, dvb, ddata = [eng.to_device(a) for a in (W, hb, vb, data)]
        stats, sc = eng.cd_step(ddata, idx, dW, dhb, dvb, gauss, 1, RngAddr(31, 2, 11, 0, 0))
        eng.synchronize()
        assert sc.planes is not None, "c2 must be on the plane path by default"
        # sample planes inside mdbn_cd_args.planes (mdbn_planes_bytes): [6 B V | 6 B H | hs: B H | vs: B V] bf16
        pl = sc.planes
        hs = ((pl[6 * B * V + 6 * B * H:][:B * H].to(torch.int32) & 0xffff) << 16).view(torch.float32).view(B, H).cpu().numpy()
        vs = ((pl[6 * B * V + 7 * B * H:][:B * V].to(torch.int32) & 0xffff) << 16).view(torch.float32).view(B, V).cpu().numpy()
        assert set(np.unique(hs)) <= {0.0, 1.0}
        st = rbm_np.RBMState(V, H, W=W, hbias=hb, vbias=vb, gauss=gauss)
        v0 = data[idx].astype(np.float64)
        th = np.stack([hs, np.zeros_like(hs)])
        tv = None if gauss else vs[None]
        rbm_np.FLIP_GAP["max"] = 0.0
        ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(31, 2, 11, 0), 1, th, tv)
        S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
        d = stats.cpu().numpy()
        S, s_h, s_v, cost = d[:V * H].reshape(V, H), d[V * H:V * H + H], d[V * H + H:V * H + H + V], d[V * H + H + V]
        tag = "product-default c2 %s" % ("GRBM" if gauss else "RBM")
        check(tag + ": S / max|S|", np.abs(S - S_o).max() / np.abs(S_o).max(), 1e-5, "stats")
        check(tag + ": s_h / max", np.abs(s_h - s_h_o).max() / max(1.0, np.abs(s_h_o).max()), 1e-5, "stats")
        check(tag + ": s_v / max", np.abs(s_v - s_v_o).max() / max(1.0, np.abs(s_v_o).max()), 1e-5, "stats")
        check(tag + ": |u - p| of a flipped draw", rbm_np.FLIP_GAP["max"], 1e-6, "tie")
        pre_nv = out[0]
        cost_o = ((rbm_np.sigmoid(pre_nv) - v0) ** 2).sum() if gauss else \
            (v0 * rbm_np.softplus(-pre_nv) + (1 - v0) * rbm_np.softplus(pre_nv)).sum()
        check(tag + ": cost rel", abs(cost - cost_o) / abs(cost_o), 2e-6)
        assert flips <= 3
    finally:
        eng.set_planes_min_work(0)


@pytest.mark.parametrize("gauss,V,H,B", [(True, 4096, 1024, 512), (False, 2048, 1024, 512), (True, 1024, 1024, 256)])
def test_early_parameter_half_is_bitwise_the_epilogue_update(hip_engine, gauss, V, H, B):
    """mdbn_set_option("early_w"): the statistics GEMM's loader waves apply W' = W * decay + speed_old * lr (and rewrite W's
    planes) DURING the main loop, the epilogue only forms the new speed -- the same functions on the same operands as the
    whole rule in the epilogue: every parameter, speed, cost and W plane bit for bit, for the GRBM (lambda_2 decay) and
    the Bernoulli RBM (frozen weight-cost snapshot, momentum); B = 256 runs two chunks per stage."""
    eng = hip_engine
    runs = []
    for early in (1, 0, 1):
        eng.set_option("early_w", early)
        try:
            out, rbm, wp, valid = _run_steps(eng, gauss, True, V, H, B, 1, steps=6, seed=4)
        finally:
            eng.set_option("early_w", 1)
        assert valid
        p1, p2, p3 = planes_to_f32(wp)
        assert torch.equal((p3 + p2) + p1, rbm.W.tensor), "W planes out of step with W"
        runs.append(out)
    for other in runs[1:]:
        for key in runs[0]:
            assert np.array_equal(runs[0][key], other[key]), key


@pytest.mark.parametrize("gauss,V,H,B", [(True, 4096, 1024, 512), (False, 2048, 2048, 512)])
def test_gather_ahead_is_bitwise_the_gather_launch(built_lib, gauss, V, H, B):
    """fn(indexes=, next_indexes=): the statistics kernel's loader waves gather the next minibatch into the other X2-plane
    buffer, the next step starts without a gather launch.  A product-default engine, 10 steps: with the hint every step,
    with a WRONG hint every third step (the step must then gather for itself), and without hints -- with ONE index buffer
    that the caller refills in place and announces as its own successor (the gathered rows are then stale and must be
    dropped) -- identical parameters, speeds and costs bit for bit; and the hinted run really took the ahead path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdbn_amd
    eng = mdbn_amd.HipEngine()
    eng.set_planes_min_work(1 << 30)
    try:
        runs, took = [], []
        for mode in ("hint", "wrong", "refill", "none"):
            rs = np.random.RandomState(11)
            N = 4 * B
            data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
            cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
            rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5), engine=eng)
            hp = dict(lr=0.001, lambda_2=0.1) if gauss else dict(lr=0.05, weightcost=2e-4)
            _, up = rbm.get_cost_updates(k=1, batch_size=B, **hp)
            fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel=None)
            batches = [eng.index_tensor(rs.permutation(N)[:B]) for _ in range(11)]
            costs, ahead = [], 0
            if mode == "refill":
                # ONE index buffer refilled in place before every call: the rows gathered ahead belong to the
                # buffer's previous contents and must not be used -- the version counter of the tensor tells
                values, buf = batches, batches[0].clone()
            for t in range(10):
                if mode == "refill":
                    buf.copy_(values[t])
                    costs.append(float(fn(indexes=buf, momentum=0.5, next_indexes=buf)))
                    ahead += int(eng.last_scratch.ahead is not None)
                    continue
                nxt = None if mode == "none" else (batches[(t + 5) % 11] if (mode == "wrong" and t % 3 == 1) else batches[t + 1])
                costs.append(float(fn(indexes=batches[t], momentum=0.5, next_indexes=nxt)))
                ahead += int(eng.last_scratch.ahead is not None)
            took.append(ahead)
            runs.append(dict(costs=np.array(costs), W=rbm.W.get_value(), Ws=rbm.W_speed.get_value(), vb=rbm.vbias.get_value()))
        assert took[0] == 10 and took[1] == 10 and took[3] == 0, took       # (a wrong hint is still gathered, just not used)
        for other in runs[1:]:
            for key in runs[0]:
                assert np.array_equal(runs[0][key], other[key]), key
    finally:
        eng.set_planes_min_work(0)


@pytest.mark.parametrize("gauss", [True, False])
def test_product_default_train_step_is_bitwise_statistics_plus_update(built_lib, gauss):
    """The path bench.py times -- a FRESH HipEngine (keep_f32 = 0, default planes_min_work), mdbn_cd_train_step with the
    update inside the statistics GEMM (parameter half early in the loader waves) and the next minibatch gathered there too --
    against the same three steps as mdbn_cd_step + mdbn_apply_update on the same engine: every parameter, speed and cost bit
    for bit.  The statistics of that second form are what test_product_default_c2_statistics_against_forced_oracle holds
    against the float64 oracle, so the chain oracle -> statistics -> update -> fused step has no link on a fixture engine
    any more."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdbn_amd
    from mdbn_amd import RngAddr
    V, H, B, N = 4096, 1024, 512, 2048
    eng = mdbn_amd.HipEngine()
    eng.set_planes_min_work(1 << 30)
    try:
        rs = np.random.RandomState(8)
        data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
        batches = [eng.index_tensor(rs.permutation(N)[:B]) for _ in range(4)]
        hp = dict(lr=0.001, lambda_2=0.1) if gauss else dict(lr=0.05, weightcost=2e-4)
        cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
        table = mdbn_amd.shared(data, engine=eng)
        # A: the step function (one library call per step, hints passed)
        rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5), engine=eng)
        _, up = rbm.get_cost_updates(k=1, batch_size=B, **hp)
        fn = mdbn_amd.function(up, table, data_parallel=None)
        costs_a = [float(fn(indexes=batches[t], momentum=0.5, next_indexes=batches[t + 1])) for t in range(3)]
        assert eng.last_scratch.planes is not None and eng.last_scratch.ahead is not None, "headline path: planes + gather-ahead"
        # B: statistics, then the update launch
        ref = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5), engine=eng)
        W0 = ref.W.tensor.clone() if not gauss else None                      # frozen weight-cost snapshot (rbm.py:415)
        costs_b = []
        for t in range(3):
            stats, _ = eng.cd_step(table.tensor, batches[t], ref.W.tensor, ref.hbias.tensor, ref.vbias.tensor, gauss, 1,
                                   RngAddr(5, ref.stream_id, t, 0, 0))
            cost_scale = 1.0 / (B * V) if gauss else 1.0 / B
            costs_b.append(float(eng.apply_update(ref.W.tensor, ref.W_speed.tensor, W0, ref.hbias.tensor, ref.hbias_speed.tensor,
                                                  ref.vbias.tensor, ref.vbias_speed.tensor, stats, hp["lr"], 0.0,
                                                  hp.get("lambda_2", 0.0), hp.get("weightcost", 0.0), 0.5, B, B, cost_scale,
                                                  ldv=table.tensor.stride(0))))
        assert costs_a == costs_b, (costs_a, costs_b)
        for name in ("W", "W_speed", "hbias", "hbias_speed", "vbias", "vbias_speed"):
            assert torch.equal(getattr(rbm, name).tensor, getattr(ref, name).tensor), name
    finally:
        eng.set_planes_min_work(0)


@pytest.mark.parametrize("gauss,V,H,B,k", [(False, 1024, 400, 256, 2), (True, 2048, 400, 512, 2), (True, 1024, 449, 256, 1)])
def test_ragged_hidden_width_on_a_padded_leading_dimension_against_forced_oracle(hip_engine, gauss, V, H, B, k):
    """A hidden width that is not a multiple of 128 (the reference's 2048 -> 400 gene-expression layer, MDBN.py:45-52) rides
    the plane path on a leading dimension padded to the next multiple: the GEMMs run on the padded width, the pad columns
    of W / activations / statistics hold exact zeros, the activation epilogues treat them as dead.  One CD-k step with the
    chain taps on, the oracle following the device; the launches must be plane kernels; the pads must come back zero."""
    from mdbn_amd import RngAddr
    eng = hip_engine
    ldh = (H + 127) // 128 * 128
    assert eng.plane_shape(B, V, H, V, ldh), "not on the plane path"
    assert not eng.plane_shape(B, V, H, V, (H + 3) // 4 * 4), "a dense ragged width cannot be"
    rs = np.random.RandomState(V + H)
    W = rbm_np.init_W(rs, V, H, np.float32)
    hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
    x = rs.normal(size=(B, V)).astype(np.float32) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
    dW = eng.alloc_matrix(V, H, ld=ldh)
    dW.copy_(torch.from_numpy(W))
    dhb, dvb, dx = [eng.to_device(a) for a in (hb, vb, x)]
    eng.trace_chain = True
    eng.kernel_timing(True)
    try:
        stats, sc = eng.cd_step(dx, None, dW, dhb, dvb, gauss, k, RngAddr(11, 1, 2, 0, 0))
        eng.synchronize()
        assert sc.planes is not None
        kinds = [kind for _, _, _, kind in eng.kernel_timing_detail()]
        assert kinds and all(kd >= 2000 for kd in kinds), kinds             # plane GEMMs only
        th = sc.trace_h.cpu().numpy()
        tv = None if gauss else sc.trace_v.cpu().numpy()
    finally:
        eng.trace_chain = False
        eng.kernel_timing(False)
    assert not th[:, :, H:].any(), "pad columns of the hidden samples"
    st = rbm_np.RBMState(V, H, W=W, hbias=hb, vbias=vb, gauss=gauss)
    v0 = x.astype(np.float64)
    ph, _, out, flips = rbm_np.cd_chain_forced(st, v0, PhiloxDraws(11, 1, 2, 0), k, th[:, :, :H], tv)
    S_o, s_h_o, s_v_o = rbm_np.cd_statistics(v0, ph, out[1], out[4])
    d = stats.cpu().numpy()
    S, s_h, s_v = d[:V * ldh].reshape(V, ldh), d[V * ldh:V * ldh + ldh], d[V * ldh + ldh:V * ldh + ldh + V]
    assert not S[:, H:].any() and not s_h[H:].any(), "pad columns of the statistics"
    assert np.abs(S[:, :H] - S_o).max() <= 1e-5 * max(1.0, np.abs(S_o).max())
    assert np.abs(s_h[:H] - s_h_o).max() <= 1e-5 * max(1.0, np.abs(s_h_o).max())
    assert np.abs(s_v - s_v_o).max() <= 1e-5 * max(1.0, np.abs(s_v_o).max())
    P2 = sc.P2.cpu().numpy()
    assert not P2[:, H:].any()
    assert np.abs(P2[:B, :H] - ph).max() <= 2e-6
    assert np.abs(-P2[B:2 * B, :H] - out[4]).max() <= 4e-6
    assert flips <= 3


@pytest.mark.parametrize("gauss", [True, False])
def test_a_ragged_layer_trains_on_padded_planes_as_on_the_exact_path(built_lib, gauss):
    """RBM / GRBM 4096 -> 1000 through the class surface on product defaults: Engine.weight_ld pads W's rows to 1024, the step
    function lands on the plane path (its launches say so: 140 us per step instead of 172, profiles/r04zg_ragged_planes_ab.log);
    four training steps agree with the same steps on the f32-operand kernels (option gemm_planes = 0) to fp32 summation
    order, and the pad columns of W stay zero.  (A layer too small for the plane path under the default rule -- 2048 -> 400:
    5 % -- keeps its dense rows.)"""
    import mdbn_amd
    eng = mdbn_amd.HipEngine()
    assert eng.weight_ld(2048, 400) is None and eng.weight_ld(4096, 1024) is None and eng.weight_ld(100, 1000) is None
    V, H, B, LD = 4096, 1000, 512, 1024
    runs = []
    for planes in (1, 0):
        eng.set_option("gemm_planes", planes)
        rs = np.random.RandomState(7)
        N = 4 * B
        data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
        cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
        rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(5), engine=eng)
        assert rbm.W.tensor.stride(0) == LD and rbm.W_speed.tensor.stride(0) == LD and rbm.W.shape == (V, H)
        hp = dict(lr=0.001, lambda_2=0.1) if gauss else dict(lr=0.05, weightcost=2e-4)
        _, up = rbm.get_cost_updates(k=1, batch_size=B, **hp)
        fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel=None)
        eng.kernel_timing(True)
        costs = [float(fn(indexes=rs.permutation(N)[:B], momentum=0.5)) for _ in range(4)]
        eng.synchronize()
        kinds = [kind for _, _, _, kind in eng.kernel_timing_detail()]
        eng.kernel_timing(False)
        assert all(kd >= 2000 for kd in kinds) == bool(planes) and kinds, kinds
        base = rbm.W.tensor._base if rbm.W.tensor._base is not None else rbm.W.tensor
        assert not base.reshape(-1)[:V * LD].reshape(V, LD)[:, H:].any().item(), "pad columns of W"
        runs.append(dict(costs=np.array(costs), W=rbm.W.get_value(), Ws=rbm.W_speed.get_value(), hb=rbm.hbias.get_value(),
                         vbs=rbm.vbias_speed.get_value()))
    eng.set_option("gemm_planes", 1)
    a, b = runs
    # (free-running chains: a hidden sample whose probability lies within rounding of its uniform falls the other way on one
    #  of the two paths about once per step at this size -- 2 M draws -- and moves its column (and, less, its row's) by ~1e-5,
    #  scripts/experiments/ragged_edge_probe2.py; every other column agrees to summation order)
    for key in ("W", "Ws", "hb"):
        assert a[key].shape == b[key].shape
        scale = max(1e-3, np.abs(b[key]).max())
        d = np.abs(a[key] - b[key])
        d = d.max(axis=0) if d.ndim == 2 else d
        assert (d > 2e-5 * scale).sum() <= 64 and np.median(d) <= 2e-6 * scale and d.max() <= 2e-3 * scale, (key, d.max(), np.median(d))
    assert np.abs(a["vbs"] - b["vbs"]).max() <= 1e-4 * max(1e-3, np.abs(b["vbs"]).max())
    np.testing.assert_allclose(a["costs"], b["costs"], rtol=1e-4)
