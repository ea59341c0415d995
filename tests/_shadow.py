"""ShadowEngine: the HIP engine with a float64 oracle riding along (TEST-ONLY).

Every CD step the host code issues (through RBM.training, DBN.training, MDBN.train_*, step functions)
runs on the device with the chain taps on (mdbn_cd_args.trace_*), and is then replayed by the oracle
FOLLOWING the device's recorded samples (oracle.rbm_np.cd_chain_forced) -- so device and oracle are
compared on identical chain states at every half-step, for any k, with no dependence on lucky seeds.
The shadow keeps one float64 RBMState per weight matrix; ``report()`` gives the worst deviations."""
import numpy as np
import torch

import mdbn_amd
from oracle import rbm_np
from oracle.philox_np import PhiloxDraws


class ShadowEngine(mdbn_amd.HipEngine):
    cd_forward = None           # every CD step goes through cd_step / cd_train_step below, where the oracle rides along

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.trace_chain = True
        self.tie = 1e-6             # |u - p| below which a recorded draw may differ from the oracle's (grows with allowed drift)
        self.shadow = {}            # W.data_ptr() -> RBMState (float64)
        self.cost_err = 0.0         # worst relative deviation of a step's monitoring cost
        self.stat_err = 0.0         # worst deviation of S / s_h / s_v relative to their max (cd_step path)
        self.flips = 0              # near-tie draws that fell the other way on the device
        self.steps = 0
        self.pl_costs = []          # oracle pseudo-likelihood costs of the PCD steps, in order
        self._pending_stats = {}

    # -- oracle state bound to a weight matrix
    def _state(self, W, hbias, vbias, gauss, speeds=None):
        key = W.data_ptr()
        st = self.shadow.get(key)
        if st is None:
            V, H = W.shape
            st = rbm_np.RBMState(V, H, W=W.cpu().numpy(), hbias=hbias.cpu().numpy(), vbias=vbias.cpu().numpy(),
                                 dtype=np.float64, gauss=gauss)
            if speeds is not None:
                st.W_speed, st.hbias_speed, st.vbias_speed = [t.cpu().numpy().astype(np.float64) for t in speeds]
            self.shadow[key] = st
        return st

    def _rows(self, data, indexes):
        data = self.as_matrix(data)
        if indexes is None:
            return data.cpu().numpy().astype(np.float64)
        idx = self.index_tensor(indexes, data.shape[0]).to(torch.int64)
        return data[idx].cpu().numpy().astype(np.float64)

    def _traces(self, gauss):
        sc = self.last_scratch
        th = sc.trace_h.cpu().numpy()[:, :, :sc.H]
        tv = None if gauss else sc.trace_v.cpu().numpy()[:, :, :sc.V]
        return th, tv

    # -- single-device step function
    def cd_train_step(self, data, indexes, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, gauss, k,
                      rng, lr, lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale,
                      sample_stats=False, next_indexes=None, cache_out=None):     # (the chain taps keep the gather-ahead and the struct cache off)
        st = self._state(W, hbias, vbias, gauss, (W_speed, hbias_speed, vbias_speed))
        st.W0 = None if W0 is None else W0.cpu().numpy().astype(np.float64)
        v0 = self._rows(data, indexes)
        cost = super().cd_train_step(data, indexes, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, gauss, k,
                                     rng, lr, lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows,
                                     cost_scale, sample_stats)
        th, tv = self._traces(gauss)
        want = rbm_np.cd_step(st, v0, PhiloxDraws(rng.seed, rng.stream_id, rng.step, rng.row_offset), lr=lr, k=k,
                              lambda_1=lambda_1, lambda_2=lambda_2, weightcost=weightcost, batch_size=batch_size,
                              momentum=momentum, strict_reference=W0 is not None, symbolic_grad=sample_stats,
                              forced=(th, tv), tie=self.tie)
        got = float(cost)
        self.cost_err = max(self.cost_err, abs(got - want) / max(abs(want), 1e-30))
        self.steps += 1
        return cost

    # -- PCD / data-parallel path: statistics, then the update
    def cd_step(self, data, indexes, W, hbias, vbias, gauss, k, rng, persistent=None, add_noise=False,
                stats_slot=0, sample_stats=False, stats=None, comm_cus=0):     # comm_cus: a launch-geometry hint
        st = self._state(W, hbias, vbias, gauss)
        v0 = self._rows(data, indexes)
        chain0 = None if persistent is None else persistent.cpu().numpy().astype(np.float64)
        stats, sc = super().cd_step(data, indexes, W, hbias, vbias, gauss, k, rng, persistent=persistent,
                                    add_noise=add_noise, stats_slot=stats_slot, sample_stats=sample_stats, stats=stats)
        th, tv = self._traces(gauss)
        draws = PhiloxDraws(rng.seed, rng.stream_id, rng.step, rng.row_offset)
        ph_mean, _, out, flips = rbm_np.cd_chain_forced(st, v0, draws, k, th, tv, chain0, tie=self.tie)
        self.flips += flips
        S, s_h, s_v = rbm_np.cd_statistics(v0, ph_mean, out[1], out[4])
        V, H = W.shape
        ldh, ldv = sc.P2.stride(0), sc.V2.stride(0)
        d = stats.cpu().numpy()
        for got, want in ((d[:V * ldh].reshape(V, ldh)[:, :H], S), (d[V * ldh:V * ldh + H], s_h),
                          (d[V * ldh + ldh:V * ldh + ldh + V], s_v)):
            self.stat_err = max(self.stat_err, np.abs(got - want).max() / max(1.0, np.abs(want).max()))
        if persistent is not None:
            self.pl_costs.append(rbm_np.pseudo_likelihood_cost(st, v0))
            st.bit_i_idx = (st.bit_i_idx + 1) % st.n_visible
            assert np.array_equal(persistent.cpu().numpy()[:, :H], th[k]), "persistent chain != recorded nh_sample"
        self._pending_stats[W.data_ptr()] = (S, s_h, s_v, v0.shape[0])
        self.steps += 1
        return stats, sc

    def apply_update(self, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr, lambda_1, lambda_2,
                     weightcost, momentum, batch_size, n_rows, cost_scale, phase=0, ldv=None):
        out = super().apply_update(W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr, lambda_1,
                                   lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale, phase, ldv)
        pend = self._pending_stats.pop(W.data_ptr(), None)
        if pend is not None and phase == 0:
            st = self.shadow[W.data_ptr()]
            st.W0 = None if W0 is None else W0.cpu().numpy().astype(np.float64)
            S, s_h, s_v, _ = pend
            S, s_h, s_v = self._reduce_oracle_stats(S, s_h, s_v)
            g = rbm_np.rbm_grad(st, S, s_h, s_v, batch_size, n_rows, weightcost, strict_reference=W0 is not None)
            rbm_np.apply_update(st, g[0], g[1], g[2], lr, lambda_1, lambda_2, momentum)
        return out

    def _reduce_oracle_stats(self, S, s_h, s_v):
        """Data-parallel runs: the oracle's per-shard statistics are summed over the ranks in float64 (the device sums
        its float32 shards through the job's collective), so every rank's shadow applies the update of the GLOBAL
        minibatch -- `N ranks == the oracle on the global batch`, teacher-forced shard by shard."""
        import torch.distributed as td
        if not (td.is_available() and td.is_initialized() and td.get_world_size() > 1):
            return S, s_h, s_v
        out = []
        for a in (S, s_h, s_v):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
            if td.get_backend() == "nccl":
                t = t.to(self.device)
            td.all_reduce(t)
            out.append(t.cpu().numpy())
        return out

    # -- verdicts
    def param_err(self, rbm):
        """Worst deviation of a layer's device parameters / speeds from its shadow, relative to max|.|."""
        st = self.shadow[rbm.W.tensor.data_ptr()]
        worst = 0.0
        for name in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed"):
            got, want = getattr(rbm, name).get_value(), getattr(st, name)
            worst = max(worst, float(np.abs(got - want).max() / max(1.0, np.abs(want).max())))
        return worst
