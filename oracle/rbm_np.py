"""numpy restatement of the reference's RBM / GRBM arithmetic (CPU oracle).

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED.

Every function cites the reference lines (``/root/reference/src/...``) it follows.
Randomness is *injected*: callers pass a ``draws`` object with ``u(draw, rows,
cols)`` / ``z(draw, rows, cols)`` (``oracle.philox_np.PhiloxDraws`` for the device
twin, ``ArrayDraws`` for literal arrays), numbered as in oracle/philox_np.py.

All arithmetic runs in ``dtype`` (float64 for parity checks, float32 for the timed
CPU baseline, mirroring theano.config.floatX).
"""
import numpy as np

EPSILON = 0.001          # rbm.py:347


class ArrayDraws(object):
    """Literal uniforms/normals: ``{draw_index: ndarray}``."""

    def __init__(self, arrays):
        self.arrays = arrays

    def u(self, draw, rows, cols):
        a = np.asarray(self.arrays[draw])
        assert a.shape == (rows, cols), (a.shape, rows, cols)
        return a

    z = u


def sigmoid(x):
    """theano.tensor.nnet.sigmoid; evaluated without overflow."""
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e)).astype(x.dtype)


def softplus(x):
    """theano.tensor.nnet.softplus = log(1+exp(x)); evaluated without overflow."""
    return (np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))).astype(x.dtype)


def init_W(numpy_rng, n_visible, n_hidden, dtype):
    """rbm.py:100-107 / dbn.py:155-159: U(+-4 sqrt(6/(n_in+n_out))) drawn in f64, cast."""
    b = 4.0 * np.sqrt(6.0 / (n_hidden + n_visible))
    return np.asarray(numpy_rng.uniform(low=-b, high=b, size=(n_visible, n_hidden)), dtype=dtype)


class RBMState(object):
    """Parameters + momentum buffers of one RBM (rbm.py:84-164).

    ``gauss`` selects GRBM (rbm.py:631-699); ``error_free`` is rbm.py:642.
    ``W0`` is the frozen snapshot ``self.W.get_value(borrow=True)`` captured when
    the update graph is built (rbm.py:415): set by ``freeze_W0``.
    """

    def __init__(self, n_visible, n_hidden, W=None, hbias=None, vbias=None,
                 numpy_rng=None, dtype=np.float64, gauss=False, error_free=True):
        self.n_visible, self.n_hidden = n_visible, n_hidden
        self.dtype = np.dtype(dtype)
        self.gauss, self.error_free = gauss, error_free
        if numpy_rng is None:
            numpy_rng = np.random.RandomState(1234)          # rbm.py:87-89
        if W is None:
            W = init_W(numpy_rng, n_visible, n_hidden, dtype)
        self.W = np.array(W, dtype=dtype)
        self.hbias = np.zeros(n_hidden, dtype) if hbias is None else np.array(hbias, dtype=dtype)
        self.vbias = np.zeros(n_visible, dtype) if vbias is None else np.array(vbias, dtype=dtype)
        self.W_speed = np.zeros((n_visible, n_hidden), dtype)    # rbm.py:153-162
        self.hbias_speed = np.zeros(n_hidden, dtype)
        self.vbias_speed = np.zeros(n_visible, dtype)
        self.W0 = None
        self.persistent = None                                   # PCD chain (rbm.py:496)
        self.bit_i_idx = 0                                       # rbm.py:425

    def freeze_W0(self):
        self.W0 = self.W.copy()

    def copy(self):
        import copy
        return copy.deepcopy(self)


# ---------------------------------------------------------------- propagation

def propup(s, vis):
    """rbm.py:187-199."""
    pre = vis @ s.W + s.hbias
    return pre, sigmoid(pre)


def sample_h_given_v(s, v0, U):
    """rbm.py:201-213; binomial(n=1,p) == (uniform < p) cast to floatX."""
    pre, mean = propup(s, v0)
    sample = (U.astype(s.dtype) < mean).astype(s.dtype)
    return pre, mean, sample


def propdown(s, hid):
    """rbm.py:215-227 (Wt is a view of W, rbm.py:139,301)."""
    pre = hid @ s.W.T + s.vbias
    return pre, sigmoid(pre)


def sample_v_given_h(s, h0, draw):
    """RBM rbm.py:229-240 (``draw`` = uniforms); GRBM rbm.py:647-660 (``draw`` =
    N(0,1) noise, ignored when error_free).  Returns [pre, mean, sample]."""
    if s.gauss:
        v1_mean = h0 @ s.W.T + s.vbias
        v1_sample = v1_mean if s.error_free else v1_mean + draw.astype(s.dtype)
        return v1_mean, v1_mean, v1_sample
    pre, mean = propdown(s, h0)
    sample = (draw.astype(s.dtype) < mean).astype(s.dtype)
    return pre, mean, sample


def gibbs_hvh(s, h0, draw_v, U_h):
    """rbm.py:242-248; GRBM rbm.py:662-671 (h1 from v1_mean, mean-field)."""
    pre_v1, v1_mean, v1_sample = sample_v_given_h(s, h0, draw_v)
    pre_h1, h1_mean, h1_sample = sample_h_given_v(s, v1_mean if s.gauss else v1_sample, U_h)
    return [pre_v1, v1_mean, v1_sample, pre_h1, h1_mean, h1_sample]


def gibbs_vhv(s, v0, U_h, draw_v):
    """rbm.py:250-256; GRBM rbm.py:673-682 (v1 from h1_mean)."""
    pre_h1, h1_mean, h1_sample = sample_h_given_v(s, v0, U_h)
    pre_v1, v1_mean, v1_sample = sample_v_given_h(s, h1_mean if s.gauss else h1_sample, draw_v)
    return [pre_h1, h1_mean, h1_sample, pre_v1, v1_mean, v1_sample]


# ---------------------------------------------------------------- energies / costs

def free_energy(s, v):
    """RBM rbm.py:166-171; GRBM rbm.py:684-688."""
    wx_b = v @ s.W + s.hbias
    hidden_term = softplus(wx_b).sum(axis=1)
    if s.gauss:
        return -hidden_term + 0.5 * ((v - s.vbias) ** 2).sum(axis=1)
    return -hidden_term - v @ s.vbias


def free_energy_gap(s, train, test):
    """rbm.py:173-180."""
    return free_energy(s, test).mean() - free_energy(s, train).mean()


def reconstruction_cost(s, pre_sigmoid_nv, v0):
    """RBM rbm.py:479-480: mean_rows(sum_cols BCE(sigmoid(pre), v0)), with
    log(sigmoid(x)) = -softplus(-x) (the rewrite the reference relies on, rbm.py:456-475).
    GRBM rbm.py:697: mean over ALL elements of (sigmoid(v1_mean) - v0)^2."""
    if s.gauss:
        return ((sigmoid(pre_sigmoid_nv) - v0) ** 2).mean()
    bce = v0 * softplus(-pre_sigmoid_nv) + (1.0 - v0) * softplus(pre_sigmoid_nv)
    return bce.sum(axis=1).mean()


def pseudo_likelihood_cost(s, v0):
    """rbm.py:421-447 (reads bit_i_idx; caller advances it as part of the updates).
    tensor.round is round-half-away-from-zero."""
    xi = np.sign(v0) * np.floor(np.abs(v0) + 0.5)
    fe_xi = free_energy(s, xi)
    xi_flip = xi.copy()
    xi_flip[:, s.bit_i_idx] = 1 - xi[:, s.bit_i_idx]
    fe_xi_flip = free_energy(s, xi_flip)
    return -np.mean(s.n_visible * softplus(fe_xi - fe_xi_flip))


# ---------------------------------------------------------------- CD-k statistics / update

def cd_chain(s, v0, draws, k, persistent=None):
    """Positive phase + k x gibbs_hvh (rbm.py:303-336).  Draw numbering: 0 = U_h0,
    2t-1 = visible draw of step t, 2t = hidden draw of step t."""
    B = v0.shape[0]
    pre_ph, ph_mean, ph_sample = sample_h_given_v(s, v0, draws.u(0, B, s.n_hidden))
    chain = ph_sample if persistent is None else persistent       # rbm.py:308-311
    Bc = chain.shape[0]
    out = None
    for t in range(1, k + 1):
        if s.gauss:
            dv = None if s.error_free else draws.z(2 * t - 1, Bc, s.n_visible)
        else:
            dv = draws.u(2 * t - 1, Bc, s.n_visible)
        out = gibbs_hvh(s, chain, dv, draws.u(2 * t, Bc, s.n_hidden))
        chain = out[5]
    return ph_mean, ph_sample, out


# largest |u - p| at which a recorded draw has differed from the oracle's own since the last reset (test bookkeeping:
# the margin left under the allowed tie width)
FLIP_GAP = {"max": 0.0}


def _follow(own, recorded, u, mean, tie, what):
    """Teacher forcing: the recorded sample must equal the oracle's own except where the uniform lies
    within ``tie`` of the probability (a draw that fp32 rounding can push either way)."""
    recorded = np.asarray(recorded, dtype=own.dtype)
    differ = own != recorded
    if differ.any():
        gap = np.abs(np.asarray(u, dtype=np.float64) - mean)[differ]
        FLIP_GAP["max"] = max(FLIP_GAP["max"], float(gap.max()))
        assert gap.max() < tie, "%s: recorded sample differs from the oracle's away from a tie (|u - p| = %g)" % (what, gap.max())
    return recorded, int(differ.sum())


def cd_chain_forced(s, v0, draws, k, trace_h, trace_v=None, persistent=None, tie=1e-6):
    """``cd_chain`` (rbm.py:303-336) that FOLLOWS a recorded chain (tests only): every sample the chain
    feeds onward is taken from ``trace_h[t]`` / ``trace_v[t-1]`` (the device's, mdbn_cd_args.trace_*)
    after checking it against the oracle's own draw outside near-ties, so probabilities are always
    compared on identical chain states -- for any k.  Returns (ph_mean, ph_sample, out, n_tie_flips)."""
    B = v0.shape[0]
    flips = 0
    U0 = draws.u(0, B, s.n_hidden)
    pre_ph, ph_mean, own = sample_h_given_v(s, v0, U0)
    ph_sample, n = _follow(own, trace_h[0], U0, ph_mean, tie, "h0")
    flips += n
    chain = ph_sample if persistent is None else persistent
    Bc = chain.shape[0]
    out = None
    for t in range(1, k + 1):
        if s.gauss:
            dv = None if s.error_free else draws.z(2 * t - 1, Bc, s.n_visible)
            pre_v, v_mean, v_sample = sample_v_given_h(s, chain, dv)
            v_in = v_mean                                                   # rbm.py:669
        else:
            Uv = draws.u(2 * t - 1, Bc, s.n_visible)
            pre_v, v_mean, own_v = sample_v_given_h(s, chain, Uv)
            v_sample, n = _follow(own_v, trace_v[t - 1], Uv, v_mean, tie, "v%d" % t)
            flips += n
            v_in = v_sample                                                 # rbm.py:246
        Uh = draws.u(2 * t, Bc, s.n_hidden)
        pre_h, h_mean, h_sample = sample_h_given_v(s, v_in, Uh)
        if t < k or persistent is not None:
            h_sample, n = _follow(h_sample, trace_h[t], Uh, h_mean, tie, "h%d" % t)
            flips += n
        out = [pre_v, v_mean, v_sample, pre_h, h_mean, h_sample]
        chain = h_sample
    return ph_mean, ph_sample, out, flips


def cd_statistics(v0, ph_mean, nv_mean, nh_mean):
    """Un-normalised sufficient statistics of rbm.py:411-417 (what a DP rank sums
    before the all-reduce): S = v0'ph - nv'nh, s_h, s_v."""
    S = v0.T @ ph_mean - nv_mean.T @ nh_mean
    s_h = (ph_mean - nh_mean).sum(axis=0)
    s_v = (v0 - nv_mean).sum(axis=0)
    return S, s_h, s_v


def rbm_grad(s, S, s_h, s_v, batch_size, n_rows, weightcost, strict_reference=True):
    """rbm.py:392-419.  W_grad divides by the batch_size ARGUMENT (:413); the bias
    gradients are true means over the rows present (:416-417).  Weight cost uses the
    frozen snapshot W0 (:415) under strict_reference, the live W otherwise."""
    dt = s.dtype.type
    Wc = s.W0 if (strict_reference and s.W0 is not None) else s.W
    g_W = S / dt(batch_size) - dt(weightcost) * Wc
    return g_W, s_h / dt(n_rows), s_v / dt(n_rows)


def apply_update(s, g_W, g_hb, g_vb, lr, lambda_1, lambda_2, momentum):
    """rbm.py:347-365.  Theano updates are simultaneous: the parameter step uses the
    OLD speed (lags one iteration), the speed is an EMA of the gradient."""
    dt = s.dtype.type
    lr, l1, l2, mu = dt(lr), dt(lambda_1), dt(lambda_2), dt(momentum)
    shrink = dt(1) + dt(2) * lr * l1 / (np.abs(s.W) + dt(EPSILON))      # :349,355
    g_W = g_W / shrink                                                   # :349-350
    m_W = (dt(1) - dt(2) * lr * l2) / shrink                             # :355
    new = {}
    for name, g, m in (("W", g_W, m_W), ("hbias", g_hb, dt(1)), ("vbias", g_vb, dt(1))):
        theta, speed = getattr(s, name), getattr(s, name + "_speed")
        new[name + "_speed"] = g + (speed - g) * mu                      # :361-362
        new[name] = theta * m + speed * lr                               # :364-365 (old speed)
    for kname, val in new.items():
        setattr(s, kname, val.astype(s.dtype))


def cd_step(s, v0, draws, lr=0.1, k=1, lambda_1=0.0, lambda_2=0.0, weightcost=0.0,
            batch_size=None, momentum=0.0, persistent=False, strict_reference=True,
            return_extras=False, symbolic_grad=False, chain_start=None, forced=None, tie=1e-6):
    """One call of the compiled step function of rbm.py:258-376 (get_cost_updates +
    theano.function with updates): mutates ``s`` and returns the monitoring cost.

    ``persistent=True`` = PCD with the chain kept in ``s.persistent`` (rbm.py:367-371).
    ``chain_start`` (tests only): start the CD chain from these hidden samples instead of the
    oracle's own positive-phase sample -- teacher forcing, so that a Bernoulli draw within fp32
    rounding of its probability on the device cannot fork the two chains.
    ``forced`` (tests only): ``(trace_h, trace_v)`` recorded by the device; the whole chain follows
    them (``cd_chain_forced``).
    """
    v0 = np.asarray(v0, dtype=s.dtype)
    if batch_size is None:
        batch_size = v0.shape[0]
    chain0 = s.persistent if persistent else chain_start
    if forced is not None:
        ph_mean, ph_sample, out, _ = cd_chain_forced(s, v0, draws, k, forced[0], forced[1], chain0, tie=tie)
    else:
        ph_mean, ph_sample, out = cd_chain(s, v0, draws, k, chain0)
    pre_nv, nv_mean, nv_sample, pre_nh, nh_mean, nh_sample = out
    if symbolic_grad:
        # rbm.py:341-342,378-390: gradient of mean F(chain_end) - mean F(input) with
        # chain_end = nv_samples[-1] held constant; no weight-cost term, true means.
        g_W, g_hb, g_vb = symbolic_grad_fn(s, v0, nv_sample)
        S, s_h, s_v = cd_statistics(v0, ph_mean, nv_sample, propup(s, nv_sample)[1])
    else:
        S, s_h, s_v = cd_statistics(v0, ph_mean, nv_mean, nh_mean)
        g_W, g_hb, g_vb = rbm_grad(s, S, s_h, s_v, batch_size, v0.shape[0], weightcost,
                                   strict_reference)
    if persistent:
        cost = pseudo_likelihood_cost(s, v0)                             # :371 (pre-update params)
    else:
        cost = reconstruction_cost(s, pre_nv, v0)                        # :374
    apply_update(s, g_W, g_hb, g_vb, lr, lambda_1, lambda_2, momentum)
    if persistent:
        s.persistent = nh_sample                                         # :369
        s.bit_i_idx = (s.bit_i_idx + 1) % s.n_visible                    # :445
    if return_extras:
        return cost, dict(ph_mean=ph_mean, ph_sample=ph_sample, nv_mean=nv_mean,
                          nv_sample=nv_sample, nh_mean=nh_mean, nh_sample=nh_sample,
                          pre_nv=pre_nv, S=S, s_h=s_h, s_v=s_v,
                          g_W=g_W, g_hb=g_hb, g_vb=g_vb)
    return cost


def symbolic_grad_fn(s, v0, chain_end):
    """rbm.py:378-390: d/dtheta [mean F(chain_end) - mean F(input)], chain_end constant.
    Closed form, used only as a cross-check of rbm_grad."""
    def dF(v):
        p = sigmoid(v @ s.W + s.hbias)
        gW = -(v.T @ p) / v.shape[0]
        ghb = -p.mean(axis=0)
        gvb = (-(v - s.vbias) if s.gauss else -v).mean(axis=0)
        return gW, ghb, gvb
    a, b = dF(chain_end), dF(v0)
    return [x - y for x, y in zip(a, b)]


# ---------------------------------------------------------------- host helpers

def get_minibatches_idx(n, batch_size, shuffle=False, rng=None):
    """utils.py:54-75.  The reference shuffles with the unseeded global numpy.random
    (utils.py:62); ``rng`` lets tests pass a seeded RandomState instead."""
    idx_list = np.arange(n, dtype="int32")
    if shuffle:
        (rng if rng is not None else np.random).shuffle(idx_list)
    minibatches = []
    start = 0
    for _ in range(n // batch_size):
        minibatches.append(idx_list[start:start + batch_size])
        start += batch_size
    if start != n:
        minibatches.append(idx_list[start:])
    return range(len(minibatches)), minibatches


def mlp_forward(W_list, b_list, x, layer=-1):
    """mlp.py:103-107 chained as dbn.py:146,214-236: sigmoid(x W_l + b_l) up to ``layer``."""
    n = len(W_list)
    last = layer if layer >= 0 else n + layer
    out = x
    for l in range(last + 1):
        out = sigmoid(out @ W_list[l] + b_list[l])
    return out


symbolic_grad = symbolic_grad_fn      # name used by the cross-check tests
