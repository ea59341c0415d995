"""An independent witness for the oracle's gradients (VERDICT r4 #3b): torch.autograd, float64, CPU.

The reference has two gradient routes: the closed form `compute_rbm_grad` (rbm.py:392-419) that every caller uses, and
`compute_symbolic_grad` (rbm.py:378-390) = `tensor.grad(mean F(chain_end) - mean F(input), params,
consider_constant=[chain_end])`.  Here the cost expression of rbm.py:386-387 is written down literally in torch -- with the
free energies of rbm.py:166-171 (RBM) and :684-688 (GRBM) -- and differentiated by autograd; nothing of the oracle's
closed forms is used on that side.  It must agree with

  * `oracle.rbm_np.symbolic_grad_fn` (the oracle's restatement of rbm.py:378-390), and
  * `oracle.rbm_np.rbm_grad` over `cd_statistics` when both are fed the same negative data (SURVEY a-7: the two routes
    are equal then; weight cost 0, batch_size = the row count),

which pins the sign, the 1/batch_size scaling and the bias means of the closed form the HIP kernels are tested against.
The oracle stays "parity unpinned" (no reference-held vector exists); this is a second derivation, not a pin."""
import numpy as np
import pytest
import torch

from oracle import rbm_np


def _free_energy_torch(v, W, hb, vb, gauss):
    wx_b = v @ W + hb                                               # rbm.py:168 / :685
    hidden_term = torch.logaddexp(wx_b, torch.zeros_like(wx_b)).sum(dim=1)     # nnet.softplus, :170 / :687
    if gauss:
        return -hidden_term + 0.5 * ((v - vb) ** 2).sum(dim=1)      # rbm.py:686-688
    return -hidden_term - v @ vb                                    # rbm.py:169-171


def _autograd_grads(W, hb, vb, v0, chain_end, gauss):
    Wt, hbt, vbt = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (W, hb, vb))
    v0t = torch.tensor(v0, dtype=torch.float64)
    ce = torch.tensor(chain_end, dtype=torch.float64)               # consider_constant=[chain_end]: a leaf without grad
    cost = _free_energy_torch(ce, Wt, hbt, vbt, gauss).mean() - _free_energy_torch(v0t, Wt, hbt, vbt, gauss).mean()   # :386-387
    gW, ghb, gvb = torch.autograd.grad(cost, [Wt, hbt, vbt])
    return gW.numpy(), ghb.numpy(), gvb.numpy()


def _state(V, H, gauss, seed):
    rs = np.random.RandomState(seed)
    s = rbm_np.RBMState(V, H, numpy_rng=rs, dtype=np.float64, gauss=gauss)
    s.hbias = rs.normal(scale=0.3, size=H)
    s.vbias = rs.normal(scale=0.3, size=V)
    return s, rs


@pytest.mark.parametrize("gauss", [False, True], ids=["rbm", "grbm"])
@pytest.mark.parametrize("shape", [(6, 4, 3), (64, 32, 8), (200, 90, 20)], ids=lambda s: "%dx%d_b%d" % s)
def test_symbolic_grad_fn_is_the_autograd_of_the_reference_cost(shape, gauss):
    V, H, B = shape
    s, rs = _state(V, H, gauss, 5)
    v0 = rs.normal(size=(B, V)) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(np.float64)
    chain_end = rs.normal(size=(B, V)) if gauss else (rs.uniform(size=(B, V)) < 0.5).astype(np.float64)
    want = _autograd_grads(s.W, s.hbias, s.vbias, v0, chain_end, gauss)
    got = rbm_np.symbolic_grad_fn(s, v0, chain_end)
    for name, g, w in zip(("W", "hbias", "vbias"), got, want):
        np.testing.assert_allclose(g, w, rtol=0, atol=1e-13 * max(1.0, np.abs(w).max()), err_msg=name)


@pytest.mark.parametrize("gauss", [False, True], ids=["rbm", "grbm"])
@pytest.mark.parametrize("k", [1, 3])
def test_rbm_grad_closed_form_is_the_autograd_of_the_reference_cost_on_the_same_negative_data(gauss, k):
    """compute_rbm_grad (rbm.py:411-417) on the chain's last MEANS == d/dtheta [mean F(nv_mean) - mean F(v0)]
    (nv_mean held constant, nh_mean = sigmoid(nv_mean W + hb) -- which is what gibbs_hvh computes for a GRBM, rbm.py:669;
    for the Bernoulli RBM the chain's nh_mean comes from the SAMPLE, so the closed form is fed sigmoid(nv_mean W + hb) here)."""
    V, H, B = 48, 20, 12
    s, rs = _state(V, H, gauss, 11)
    v0 = rs.normal(size=(B, V)) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(np.float64)
    draws = rbm_np.ArrayDraws({d: rs.uniform(size=(B, H if d % 2 == 0 else V)) for d in range(2 * k + 1)})
    ph_mean, _, out = rbm_np.cd_chain(s, v0, draws, k)
    nv_mean = out[1]
    nh_mean = rbm_np.propup(s, nv_mean)[1]
    if gauss:
        np.testing.assert_array_equal(nh_mean, out[4])              # the GRBM chain IS mean-field (rbm.py:669)
    S, s_h, s_v = rbm_np.cd_statistics(v0, ph_mean, nv_mean, nh_mean)
    got = rbm_np.rbm_grad(s, S, s_h, s_v, batch_size=B, n_rows=B, weightcost=0.0)
    want = _autograd_grads(s.W, s.hbias, s.vbias, v0, nv_mean, gauss)
    for name, g, w in zip(("W", "hbias", "vbias"), got, want):
        np.testing.assert_allclose(g, w, rtol=0, atol=1e-13 * max(1.0, np.abs(w).max()), err_msg=name)


def test_the_update_moves_along_the_autograd_direction():
    """apply_update ADDS lr * speed (rbm.py:364) and the speed is an EMA of the gradient (:361): after two steps on the same
    minibatch with momentum 0 the parameter change is lr * (gradient of step 1), i.e. ascent on F(neg) - F(pos)."""
    V, H, B = 10, 6, 5
    s, rs = _state(V, H, False, 3)
    v0 = (rs.uniform(size=(B, V)) < 0.4).astype(np.float64)
    nv = (rs.uniform(size=(B, V)) < 0.5).astype(np.float64)
    gW, ghb, gvb = _autograd_grads(s.W, s.hbias, s.vbias, v0, nv, False)
    W_before = s.W.copy()
    rbm_np.apply_update(s, gW.copy(), ghb.copy(), gvb.copy(), lr=0.1, lambda_1=0.0, lambda_2=0.0, momentum=0.0)
    np.testing.assert_array_equal(s.W, W_before)                    # first step: old speed = 0 (the one-iteration lag)
    rbm_np.apply_update(s, np.zeros_like(gW), np.zeros_like(ghb), np.zeros_like(gvb), lr=0.1, lambda_1=0.0, lambda_2=0.0,
                        momentum=0.0)
    np.testing.assert_allclose(s.W - W_before, 0.1 * gW, rtol=0, atol=1e-15)
