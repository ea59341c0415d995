"""Pins for the oracle (PARITY UNPINNED against the reference itself, which has no tests and
needs Theano): algebraic known answers that hold for ANY correct implementation of the
reference's formulas, hand-computed update-rule cases, and numpy RandomState known answers."""
import itertools

import numpy as np
import pytest

from oracle import rbm_np
from oracle.rbm_np import ArrayDraws, RBMState


def small_state(V, H, seed, gauss=False):
    rng = np.random.RandomState(seed)
    s = RBMState(V, H, numpy_rng=rng, gauss=gauss)
    s.W *= 0.5
    s.hbias = rng.normal(0, 0.3, H)
    s.vbias = rng.normal(0, 0.3, V)
    return s


def test_init_known_answers():
    # numpy legacy RandomState stream is frozen
    rs = np.random.RandomState(123)
    assert rs.randint(2 ** 30) == 843828734
    W = rbm_np.init_W(rs, 784, 500, np.float64)
    np.testing.assert_allclose(W[0, :4], [0.11645861, -0.03911701, 0.10438896, 0.11984645], atol=1e-8)
    assert abs(W[783, 499] - 0.23165458854559234) < 1e-15
    assert abs(4 * np.sqrt(6.0 / 1284) - 0.27343437080986527) < 1e-15
    assert np.random.RandomState(1234).randint(2 ** 30) == 822569775


def test_free_energy_bruteforce_rbm():
    """F(v) = -log sum_h exp(-E(v,h)),  E = -v'Wh - b'h - c'v   (rbm.py:166-171)."""
    V, H = 6, 4
    s = small_state(V, H, 0)
    vs = np.array(list(itertools.product([0, 1], repeat=V)), dtype=np.float64)
    hs = np.array(list(itertools.product([0, 1], repeat=H)), dtype=np.float64)
    E = -(vs @ s.W @ hs.T) - (hs @ s.hbias)[None, :] - (vs @ s.vbias)[:, None]
    F_brute = -np.log(np.exp(-E).sum(axis=1))
    np.testing.assert_allclose(rbm_np.free_energy(s, vs), F_brute, rtol=1e-12, atol=1e-12)


def test_free_energy_bruteforce_grbm():
    """GRBM energy E = 0.5|v-c|^2 - v'Wh - b'h  (rbm.py:684-688)."""
    V, H = 5, 4
    s = small_state(V, H, 1, gauss=True)
    v = np.random.RandomState(5).normal(size=(7, V))
    hs = np.array(list(itertools.product([0, 1], repeat=H)), dtype=np.float64)
    E = 0.5 * ((v - s.vbias) ** 2).sum(1)[:, None] - v @ s.W @ hs.T - (hs @ s.hbias)[None, :]
    np.testing.assert_allclose(rbm_np.free_energy(s, v), -np.log(np.exp(-E).sum(axis=1)), rtol=1e-12)


def test_conditionals_match_energy():
    """propup/propdown are the exact conditionals of the RBM energy (rbm.py:187-227)."""
    V, H = 5, 3
    s = small_state(V, H, 2)
    rng = np.random.RandomState(3)
    v = (rng.uniform(size=(4, V)) < 0.5).astype(np.float64)
    hs = np.array(list(itertools.product([0, 1], repeat=H)), dtype=np.float64)
    E = -(v @ s.W @ hs.T) - (hs @ s.hbias)[None, :]
    p = np.exp(-E)
    p /= p.sum(1, keepdims=True)
    np.testing.assert_allclose(rbm_np.propup(s, v)[1], p @ hs, rtol=1e-12)


@pytest.mark.parametrize("gauss", [False, True])
def test_rbm_grad_is_free_energy_gradient(gauss):
    """compute_rbm_grad (rbm.py:392-419) == d/dtheta [mean F(neg) - mean F(pos)] with the
    negative data held constant (the identity rbm.py:386-389 encodes), when the negative
    'means' are the data the free energy is evaluated on.  Checked by finite differences."""
    V, H, B = 6, 4, 5
    s = small_state(V, H, 4, gauss=gauss)
    rng = np.random.RandomState(7)
    pos = rng.normal(size=(B, V)) if gauss else (rng.uniform(size=(B, V)) < 0.5).astype(np.float64)
    neg = rng.normal(size=(B, V)) if gauss else (rng.uniform(size=(B, V)) < 0.5).astype(np.float64)

    def cost(st):
        return rbm_np.free_energy(st, neg).mean() - rbm_np.free_energy(st, pos).mean()

    ph, nh = rbm_np.propup(s, pos)[1], rbm_np.propup(s, neg)[1]
    S, s_h, s_v = rbm_np.cd_statistics(pos, ph, neg, nh)
    g_W, g_hb, g_vb = rbm_np.rbm_grad(s, S, s_h, s_v, B, B, 0.0)
    sym = rbm_np.symbolic_grad(s, pos, neg)
    for a, b in zip((g_W, g_hb, g_vb), sym):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)
    eps = 1e-6
    for name, g in (("W", g_W), ("hbias", g_hb), ("vbias", g_vb)):
        arr = getattr(s, name)
        it = np.nditer(arr, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            old = arr[i]
            arr[i] = old + eps; cp = cost(s)
            arr[i] = old - eps; cm = cost(s)
            arr[i] = old
            assert abs((cp - cm) / (2 * eps) - g[i]) < 1e-6, (name, i)


def test_update_rule_hand_computed():
    """rbm.py:347-365 on a 2x2 case: lambda shrink, EMA speed, lagged apply."""
    s = RBMState(2, 2, W=[[0.5, -0.25], [0.0, 1.0]])
    s.W_speed = np.array([[0.1, 0.2], [-0.3, 0.4]])
    s.hbias_speed = np.array([0.5, -0.5]); s.vbias_speed = np.array([1.0, 2.0])
    W, sp = s.W.copy(), s.W_speed.copy()
    g = np.array([[1.0, 2.0], [3.0, 4.0]]); ghb = np.array([0.1, 0.2]); gvb = np.array([0.3, 0.4])
    lr, l1, l2, mu = 0.1, 0.01, 0.1, 0.6
    rbm_np.apply_update(s, g.copy(), ghb, gvb, lr, l1, l2, mu)
    shrink = 1 + 2 * lr * l1 / (np.abs(W) + 0.001)
    gs = g / shrink
    np.testing.assert_allclose(s.W_speed, gs + (sp - gs) * mu, rtol=1e-15)          # (1-mu) g + mu s
    np.testing.assert_allclose(s.W_speed, (1 - mu) * gs + mu * sp, rtol=1e-14)
    np.testing.assert_allclose(s.W, W * (1 - 2 * lr * l2) / shrink + sp * lr, rtol=1e-15)   # OLD speed
    np.testing.assert_allclose(s.hbias, [0.05, -0.05], rtol=1e-15)                  # 0 + old speed * lr
    np.testing.assert_allclose(s.hbias_speed, ghb + (np.array([0.5, -0.5]) - ghb) * mu)
    np.testing.assert_allclose(s.vbias, [0.1, 0.2], rtol=1e-15)
    # shrink at W = 0: 1 + 2 lr l1 / eps
    assert abs(shrink[1, 0] - (1 + 2 * 0.1 * 0.01 / 0.001)) < 1e-15


def test_first_step_only_shrinks():
    """The applied step lags one iteration: with zero speeds the first call
    changes W only through the multiplier."""
    s = small_state(6, 4, 8, gauss=True)
    W0, hb0 = s.W.copy(), s.hbias.copy()
    v0 = np.random.RandomState(1).normal(size=(3, 6))
    U = {0: np.random.RandomState(2).uniform(size=(3, 4)), 2: np.random.RandomState(3).uniform(size=(3, 4))}
    rbm_np.cd_step(s, v0, ArrayDraws(U), lr=0.005, k=1, lambda_2=0.1, batch_size=3)
    np.testing.assert_allclose(s.W, W0 * (1 - 2 * 0.005 * 0.1), rtol=1e-15)
    assert np.abs(s.W_speed).max() > 0
    assert np.all(s.hbias == hb0) and np.abs(s.hbias_speed).max() > 0


def test_frozen_weightcost_snapshot():
    """rbm.py:415: the weight-cost term uses the W captured at graph-build time."""
    s = small_state(6, 4, 9)
    s.freeze_W0()
    frozen = s.W0.copy()
    S = np.zeros((6, 4)); z4, z6 = np.zeros(4), np.zeros(6)
    s.W += 1.0                                   # live W moves on
    g_strict = rbm_np.rbm_grad(s, S, z4, z6, 3, 3, 2e-4, strict_reference=True)[0]
    g_live = rbm_np.rbm_grad(s, S, z4, z6, 3, 3, 2e-4, strict_reference=False)[0]
    np.testing.assert_allclose(g_strict, -2e-4 * frozen)
    np.testing.assert_allclose(g_live, -2e-4 * s.W)


def test_grad_divides_by_batch_size_argument():
    """rbm.py:413 divides by the argument; the bias means use the rows present (:416-417)."""
    s = small_state(6, 4, 10)
    rng = np.random.RandomState(0)
    S = rng.normal(size=(6, 4)); s_h = rng.normal(size=4); s_v = rng.normal(size=6)
    g = rbm_np.rbm_grad(s, S, s_h, s_v, batch_size=20, n_rows=7, weightcost=0.0)
    np.testing.assert_allclose(g[0], S / 20); np.testing.assert_allclose(g[1], s_h / 7)
    np.testing.assert_allclose(g[2], s_v / 7)


def test_grbm_chain_is_mean_field():
    """GRBM gibbs_hvh feeds the visible MEAN upward (rbm.py:669); noise never reaches h1."""
    s = small_state(6, 4, 11, gauss=True)
    s.error_free = False
    h0 = (np.random.RandomState(0).uniform(size=(3, 4)) < 0.5).astype(np.float64)
    Z = np.random.RandomState(1).normal(size=(3, 6)); U = np.random.RandomState(2).uniform(size=(3, 4))
    out = rbm_np.gibbs_hvh(s, h0, Z, U)
    mean = h0 @ s.W.T + s.vbias
    np.testing.assert_allclose(out[1], mean); np.testing.assert_allclose(out[2], mean + Z)
    np.testing.assert_allclose(out[4], rbm_np.sigmoid(mean @ s.W + s.hbias))


def test_reconstruction_costs():
    s = small_state(6, 4, 12)
    x = np.random.RandomState(0).normal(size=(3, 6)) * 3
    t = (np.random.RandomState(1).uniform(size=(3, 6)) < 0.5).astype(np.float64)
    o = 1 / (1 + np.exp(-x))
    want = (-(t * np.log(o) + (1 - t) * np.log(1 - o))).sum(1).mean()      # rbm.py:479-480
    assert abs(rbm_np.reconstruction_cost(s, x, t) - want) < 1e-12
    s.gauss = True
    assert abs(rbm_np.reconstruction_cost(s, x, t) - ((o - t) ** 2).mean()) < 1e-15   # rbm.py:697


def test_pseudo_likelihood_direct():
    """rbm.py:421-447: cost = mean(V * log p(x_i | x_-i)) for the flipped bit."""
    s = small_state(6, 4, 13)
    s.bit_i_idx = 2
    v = (np.random.RandomState(0).uniform(size=(5, 6)) < 0.5).astype(np.float64)
    flip = v.copy(); flip[:, 2] = 1 - flip[:, 2]
    Fa, Fb = rbm_np.free_energy(s, v), rbm_np.free_energy(s, flip)
    logp = -Fa - np.logaddexp(-Fa, -Fb)
    assert abs(rbm_np.pseudo_likelihood_cost(s, v) - (6 * logp).mean()) < 1e-12


def test_minibatches_idx():
    r, mbs = rbm_np.get_minibatches_idx(47, 10)
    assert len(r) == 5 and [len(m) for m in mbs] == [10, 10, 10, 10, 7]
    assert mbs[0].dtype == np.int32 and np.array_equal(np.concatenate(mbs), np.arange(47))
    _, mbs = rbm_np.get_minibatches_idx(40, 10, shuffle=True, rng=np.random.RandomState(0))
    assert len(mbs) == 4 and sorted(np.concatenate(mbs).tolist()) == list(range(40))


def test_float32_close_to_float64():
    s64 = small_state(64, 32, 14, gauss=True)
    s32 = RBMState(64, 32, W=s64.W, hbias=s64.hbias, vbias=s64.vbias, dtype=np.float32, gauss=True)
    v0 = np.random.RandomState(0).normal(size=(8, 64))
    U = {0: np.random.RandomState(1).uniform(size=(8, 32)), 2: np.random.RandomState(2).uniform(size=(8, 32))}
    c64 = rbm_np.cd_step(s64, v0, ArrayDraws(U), lr=0.005, lambda_2=0.1, batch_size=8)
    c32 = rbm_np.cd_step(s32, v0, ArrayDraws(U), lr=0.005, lambda_2=0.1, batch_size=8)
    assert abs(c64 - c32) < 1e-5 and np.abs(s64.W_speed - s32.W_speed).max() < 1e-5


def test_forced_chain_follows_a_recording():
    """cd_chain_forced (teacher forcing for the device tests): following its own recording reproduces
    cd_chain; a recorded sample that differs away from a tie is refused, one within a tie is followed."""
    from oracle.philox_np import PhiloxDraws
    for gauss in (False, True):
        V, H, B, k = 9, 6, 5, 3
        rs = np.random.RandomState(2)
        s = rbm_np.RBMState(V, H, numpy_rng=rs, gauss=gauss)
        v0 = rs.normal(size=(B, V)) if gauss else (rs.uniform(size=(B, V)) < 0.4).astype(np.float64)
        draws = PhiloxDraws(3, 1, 4)
        ph, ph_s, out = rbm_np.cd_chain(s, v0, draws, k)
        # record: hidden samples h0..h_{k-1}, visible samples v1..vk (Bernoulli)
        th, tv, chain = [ph_s], [], ph_s
        for t in range(1, k + 1):
            dv = None if gauss else draws.u(2 * t - 1, B, V)
            o = rbm_np.gibbs_hvh(s, chain, dv, draws.u(2 * t, B, H))
            tv.append(o[2]); th.append(o[5]); chain = o[5]
        ph2, ph_s2, out2, flips = rbm_np.cd_chain_forced(s, v0, draws, k, th, None if gauss else tv)
        assert flips == 0 and np.array_equal(ph, ph2) and all(np.array_equal(a, b) for a, b in zip(out, out2))
        bad = [t.copy() for t in th]
        u0 = draws.u(0, B, H)
        far = np.unravel_index(np.argmax(np.abs(u0 - ph)), ph.shape)
        bad[0][far] = 1 - bad[0][far]
        with pytest.raises(AssertionError):
            rbm_np.cd_chain_forced(s, v0, draws, k, bad, None if gauss else tv)
        # a flip within the tie width is followed, and the rest of the chain is computed from it
        # (k = 1: the step after the flipped h0 is the one returned)
        _, _, out1, _ = rbm_np.cd_chain_forced(s, v0, draws, 1, th, None if gauss else tv)
        _, _, out3, flips = rbm_np.cd_chain_forced(s, v0, draws, 1, bad, None if gauss else tv, tie=1.0)
        assert flips >= 1 and not np.array_equal(out3[1], out1[1])
