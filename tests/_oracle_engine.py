"""CPU stand-in for mdbn_amd.engine.HipEngine built on the oracle (TEST-ONLY: lives under
tests/, the product never imports it).  Same method surface, float64 numpy arithmetic
from oracle/rbm_np.py, same Philox addressing -- so host logic (RBM/DBN classes, the
training loops, data-parallel sharding) can be exercised without a GPU, and GPU tests can
run the identical host code on both engines and compare."""
import numpy as np
import torch

from oracle import rbm_np
from oracle.philox_np import PhiloxDraws, uniform, normal


class _Scratch(object):
    pass


class OracleEngine(object):
    name = "oracle"

    def __init__(self, dtype=np.float64):
        self.device = torch.device("cpu")
        self.np_dtype = np.dtype(dtype)
        self.t_dtype = torch.float64 if self.np_dtype == np.float64 else torch.float32
        self._stats = {}
        self.last = None

    # --- arrays
    def alloc_matrix(self, rows, cols, ld=None):
        return torch.zeros((rows, cols), dtype=self.t_dtype)

    def alloc_vector(self, n):
        return torch.zeros(int(n), dtype=self.t_dtype)

    def to_device(self, value):
        if isinstance(value, torch.Tensor):
            return value.detach().to(dtype=self.t_dtype, device="cpu").contiguous().clone()
        return torch.from_numpy(np.array(value, dtype=self.np_dtype))

    @staticmethod
    def is_matrix(t):
        return t.dim() == 2

    def as_matrix(self, x):
        t = getattr(x, "tensor", x)
        if isinstance(t, torch.Tensor) and t.dtype == self.t_dtype and t.device.type == "cpu":
            return t
        return self.to_device(t)

    def to_numpy(self, t):
        return t.detach().cpu().numpy().astype(np.float32)

    def index_tensor(self, indexes, n_rows=None):
        if isinstance(indexes, torch.Tensor):
            return indexes.to(torch.int64)
        return torch.from_numpy(np.asarray(indexes).astype(np.int64))

    def new_stats_buffer(self, V, H, ldv=None, ldh=None):
        return torch.zeros(V * H + H + V + 4, dtype=self.t_dtype)

    def stats_buffer(self, V, H, slot=0, ldv=None, ldh=None):
        key = (V, H, slot)
        if key not in self._stats:
            self._stats[key] = self.new_stats_buffer(V, H)
        return self._stats[key]

    # --- helpers
    def _state(self, W, hbias, vbias, gauss):
        V, H = W.shape
        s = rbm_np.RBMState(V, H, W=W.numpy(), hbias=hbias.numpy(), vbias=vbias.numpy(),
                            dtype=self.np_dtype, gauss=gauss)
        return s

    def _t(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=self.np_dtype))

    def _u(self, rng, rows, cols):
        return uniform(rows, cols, rng.seed, rng.stream_id, rng.step, rng.draw, rng.row_offset)

    # --- propagation
    def propup(self, v, W, hbias, rng=None, want_pre=True, want_mean=True, want_sample=True):
        v = self.as_matrix(v).numpy()
        s = self._state(W, hbias, torch.zeros(W.shape[0], dtype=self.t_dtype), False)
        pre, mean = rbm_np.propup(s, v)
        sample = None
        if want_sample and rng is not None:
            sample = (self._u(rng, v.shape[0], W.shape[1]).astype(self.np_dtype) < mean).astype(self.np_dtype)
        return (self._t(pre) if want_pre else None, self._t(mean) if want_mean else None,
                self._t(sample) if sample is not None else None)

    def propdown(self, h, W, vbias, gauss=False, add_noise=False, rng=None, v0=None):
        h = self.as_matrix(h).numpy()
        s = self._state(W, torch.zeros(W.shape[1], dtype=self.t_dtype), vbias, gauss)
        s.error_free = not add_noise
        B, V = h.shape[0], W.shape[0]
        if gauss:
            draw = normal(B, V, rng.seed, rng.stream_id, rng.step, rng.draw, rng.row_offset) if add_noise else None
        else:
            draw = self._u(rng, B, V)
        pre, mean, sample = rbm_np.sample_v_given_h(s, h, draw)
        out = (self._t(pre), self._t(mean), self._t(sample))
        if v0 is not None:
            v0 = self.as_matrix(v0).numpy()
            if gauss:
                c = ((rbm_np.sigmoid(pre) - v0) ** 2).sum()
            else:
                c = (v0 * rbm_np.softplus(-pre) + (1 - v0) * rbm_np.softplus(pre)).sum()
            return out + (torch.tensor(c, dtype=self.t_dtype),)
        return out

    def free_energy(self, x, W, hbias, vbias, gauss):
        s = self._state(W, hbias, vbias, gauss)
        return self._t(rbm_np.free_energy(s, self.as_matrix(x).numpy()))

    def gather_rows(self, src, indexes):
        return self.as_matrix(src)[self.index_tensor(indexes)].clone()

    # --- CD-k
    def cd_step(self, data, indexes, W, hbias, vbias, gauss, k, rng, persistent=None, add_noise=False,
                stats_slot=0, sample_stats=False, stats=None):
        data = self.as_matrix(data)
        v0 = (data if indexes is None else data[self.index_tensor(indexes)]).numpy()
        s = self._state(W, hbias, vbias, gauss)
        draws = PhiloxDraws(rng.seed, rng.stream_id, rng.step, rng.row_offset)
        chain0 = persistent.numpy().copy() if persistent is not None else None
        ph_mean, ph_sample, out = rbm_np.cd_chain(s, v0, draws, k, chain0)
        pre_nv, nv_mean, nv_sample, pre_nh, nh_mean, nh_sample = out
        if sample_stats:      # compute_symbolic_grad: chain_end = nv_samples[-1] (rbm.py:339-342)
            nv_mean, nh_mean = nv_sample, rbm_np.propup(s, nv_sample)[1]
        S, s_h, s_v = rbm_np.cd_statistics(v0, ph_mean, nv_mean, nh_mean)
        if gauss:
            cost = ((rbm_np.sigmoid(pre_nv) - v0) ** 2).sum()
        else:
            cost = (v0 * rbm_np.softplus(-pre_nv) + (1 - v0) * rbm_np.softplus(pre_nv)).sum()
        if persistent is not None:
            persistent.copy_(self._t(nh_sample))
        if stats is None:
            stats = self.stats_buffer(W.shape[0], W.shape[1], stats_slot)
        stats.copy_(self._t(np.concatenate([S.ravel(), s_h, s_v, [cost, 0, 0, 0]])))
        sc = _Scratch()
        sc.ph_mean, sc.nv_mean, sc.nh_mean, sc.ph_sample = ph_mean, nv_mean, nh_mean, ph_sample
        self.last = sc
        return stats, sc

    def apply_update(self, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats,
                     lr, lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale, phase=0, ldv=None):
        if phase == 3:      # speeds, then parameters from the NEW speeds (mdbn_update_args.phase)
            cost = self.apply_update(W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr, lambda_1,
                                     lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale, 1, ldv)
            self.apply_update(W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr, lambda_1,
                              lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale, 2, ldv)
            return cost
        V, H = W.shape
        s = rbm_np.RBMState(V, H, W=W.numpy(), hbias=hbias.numpy(), vbias=vbias.numpy(),
                            dtype=self.np_dtype)
        s.W_speed, s.hbias_speed, s.vbias_speed = (W_speed.numpy().copy(), hbias_speed.numpy().copy(),
                                                   vbias_speed.numpy().copy())
        s.W0 = W0.numpy() if W0 is not None else None
        st = stats.numpy()
        S = st[:V * H].reshape(V, H)
        s_h = st[V * H:V * H + H]
        s_v = st[V * H + H:V * H + H + V]
        g = rbm_np.rbm_grad(s, S, s_h, s_v, batch_size, n_rows, weightcost, strict_reference=W0 is not None)
        rbm_np.apply_update(s, g[0], g[1], g[2], lr, lambda_1, lambda_2, momentum)
        params = ((W, s.W), (hbias, s.hbias), (vbias, s.vbias))
        speeds = ((W_speed, s.W_speed), (hbias_speed, s.hbias_speed), (vbias_speed, s.vbias_speed))
        for dst, src in (params if phase != 1 else ()) + (speeds if phase != 2 else ()):
            dst.copy_(self._t(src))
        return torch.tensor(float(st[V * H + H + V]) * cost_scale, dtype=self.t_dtype)

    def cd_train_step(self, data, indexes, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, gauss, k,
                      rng, lr, lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale,
                      sample_stats=False, next_indexes=None):
        stats, _ = self.cd_step(data, indexes, W, hbias, vbias, gauss, k, rng, sample_stats=sample_stats)
        return self.apply_update(W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr, lambda_1,
                                 lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale)

    def round_flip(self, x, flip_col=-1):
        x = self.as_matrix(x).numpy()
        xi = np.sign(x) * np.floor(np.abs(x) + 0.5)
        if flip_col >= 0:
            xi[:, flip_col] = 1 - xi[:, flip_col]
        return self._t(xi)

    def pl_cost(self, fe, fe_flip, n_visible):
        return torch.tensor(-np.mean(n_visible * rbm_np.softplus(fe.numpy() - fe_flip.numpy())), dtype=self.t_dtype)

    def recon_cost(self, pre, target, gauss):
        s = rbm_np.RBMState(1, 1, dtype=self.np_dtype, gauss=gauss)
        return torch.tensor(rbm_np.reconstruction_cost(s, self.as_matrix(pre).numpy(), self.as_matrix(target).numpy()),
                            dtype=self.t_dtype)

    def tanh_(self, x):
        return x.tanh_()

    def count_nonfinite(self, *tensors):
        return int(sum((~torch.isfinite(t)).sum().item() for t in tensors))

    def rng_uniform(self, rows, cols, rng, normal_=False):
        f = normal if normal_ else uniform
        return torch.from_numpy(f(rows, cols, rng.seed, rng.stream_id, rng.step, rng.draw, rng.row_offset))

    def narrow_bf16(self, src, out=None):
        return src.to(torch.bfloat16)

    def widen_bf16(self, wire, dst):
        dst.copy_(wire)
        return dst

    def cost_values(self, costs):
        return [float(c) for c in costs]

    def synchronize(self):
        pass
