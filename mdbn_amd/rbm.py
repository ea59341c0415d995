"""RBM / GRBM with the class surface of the reference's src/rbm.py, executed eagerly on
MI355X through the HIP engine (no Theano graphs).

Where the reference returns symbolic expressions these methods return device arrays
(``SharedArray``; ``numpy.asarray`` / ``.get_value()`` bring them to the host).  The
compiled step function of the reference -- ``theano.function(..., updates=updates,
givens={input: train_set_x[indexes], momentum: m})`` (rbm.py:533-544, dbn.py:302-312) --
is ``mdbn_amd.function(updates, train_set_x)``, called as ``fn(indexes=, momentum=, lr=)``.

Randomness: Philox streams addressed by (seed, stream, step, draw, global row); every
sampling call and every step-function call consumes one ``step`` (see csrc/philox.h).
"""
from __future__ import print_function

import timeit

import numpy
import torch

from . import dist as _dist
from .engine import RngAddr, get_engine
from .rng import RandomStreams
from .shared import HostTable, SharedArray, as_tensor, shared, weight_ld
from .utils import get_minibatches_idx


class Scalar(object):
    """A symbolic scalar input of a step function (``tensor.scalar('lr')``, dbn.py:272)."""

    def __init__(self, name):
        self.name = name


class CostHandle(object):
    """What ``get_cost_updates`` returns as the monitoring cost: the value itself is
    produced by each call of the step function (rbm.py:367-374)."""

    def __init__(self, kind):
        self.kind = kind      # 'reconstruction' | 'pseudo_likelihood'


class UpdatePlan(object):
    """The ``updates`` dictionary of rbm.py:353-371 in closed form: everything one
    CD-k / PCD-k step needs, bound when ``get_cost_updates`` is called."""

    def __init__(self, rbm, lr, k, lambda_1, lambda_2, weightcost, batch_size, persistent, W0,
                 symbolic_grad=False):
        self.rbm, self.lr, self.k = rbm, lr, int(k)
        self.lambda_1, self.lambda_2, self.weightcost = lambda_1, lambda_2, weightcost
        self.batch_size, self.persistent, self.W0 = batch_size, persistent, W0
        # compute_symbolic_grad (rbm.py:378-390): negative data = nv_samples[-1], no weight-cost
        # term, true means (the batch_size argument is not used)
        self.symbolic_grad = bool(symbolic_grad)


class LazyCost(object):
    """Monitoring cost of a data-parallel step whose all-reduce is still in flight: resolving
    it (``float(c)``) completes the deferred part of that step first."""

    def __init__(self, stepfn, token):
        self._stepfn, self._token, self._value = stepfn, token, None

    def _resolve(self):
        if self._value is None:
            self._stepfn._resolve_cost(self)
        return self._value

    def reshape(self, *shape):
        return self._resolve().reshape(*shape)

    def __float__(self):
        return float(self._resolve())

    def item(self):
        return float(self)


class StepFunction(object):
    """One compiled training step: ``fn(indexes=, momentum=, lr=)`` -> monitoring cost
    (0-d device tensor; ``float(cost)`` synchronises).

    Data-parallel: with an initialised ``torch.distributed`` group of N ranks
    each rank computes the statistics of its contiguous slice of the minibatch, the packed
    [S | s_h | s_v | cost] buffer is sum-all-reduced (RCCL over xGMI), and every rank applies
    the same update; ``batch_size`` is the global minibatch size (rbm.py:413).

    Overlap: the parameter step uses the OLD speed (rbm.py:364-365), so theta(t+1) does not
    depend on step t's gradient.  With ``overlap=True`` (default when eligible: lambda_1 == 0
    and weightcost == 0 or a frozen W0) the all-reduce of step t is started asynchronously,
    theta(t+1) is applied at once, and the speed update that needs the reduced statistics runs
    at the end of step t+1 -- the collective hides behind a whole step of compute while every
    value stays bit-identical to the synchronous order.  ``flush()`` completes the deferred
    part (called automatically when speeds or the cost are read)."""

    accepts_next_indexes = True         # fn(indexes=, momentum=, lr=, next_indexes=): the trainers pass the hint when they can

    def __init__(self, updates, train_set_x, input_fn=None, name=None, data_parallel="auto", overlap=True):
        self.plan = updates
        self.rbm = updates.rbm
        self.engine = self.rbm.engine
        self.train_set_x = train_set_x
        self.input_fn = input_fn            # maps the DBN's data matrix to this layer's input
        self.name = name
        self.group = _dist.default_group() if data_parallel == "auto" else data_parallel
        p = updates
        self.overlap = bool(overlap and self.group is not None and self.group.world_size > 1
                            and p.persistent is None and p.lambda_1 == 0.0
                            and (p.weightcost == 0.0 or p.W0 is not None))
        self._pending = None                # (work, stats, hyper-parameters, LazyCost) of the last step
        self._staging = None                # feed of a host-resident table (RowFeeder, announced minibatches)
        self._n_calls = 0
        # statistics buffers of the data-parallel path are OWNED by this step function: a deferred
        # update reads them a whole call later, so another step function of the same shape (two
        # equal-sized modalities, alternating layers) must not share them
        self._stats_slots = [None, None]
        self._fast = []                     # argument structs of the previous single-device call (engine.cd_train_step_cached)
        # nan_guard: check cost and parameters for NaN / Inf after every call (synchronises; what the
        # reference's commented-out NanGuardMode would do, rbm.py:542-543, dbn.py:311)
        self.nan_guard = bool(getattr(self.engine, "nan_guard", False))
        self.comm_cus = 0
        # Overlapped data-parallel order with the deferred update of step t-1 INSIDE the statistics GEMM of step t (one
        # launch and one queue packet fewer: 163 -> 146 us per step without a collective on one GPU), at the price of waiting
        # for the all-reduce of step t-1 before that GEMM instead of after it (~100 us of cover instead of ~150).  The
        # balanced launches (comm_cus > 0) can carry it too (a flat share of the arrays per workgroup), but gain only 3 us
        # from it and lose the longer cover (185.0 vs 187.8 us alone, 216 vs 203 beside a 120-us stand-in): only with
        # MDBN_DP_FUSED_UPDATE=2 / fn.fuse_deferred = 2.  0 restores the update launch after the step everywhere;
        # bench.py --gpus N measures all of them.
        import os as _os
        self.fuse_deferred = int(_os.environ.get("MDBN_DP_FUSED_UPDATE", "1"))      # 0 | 1 | 2 (2: balanced launches too)
        if self.group is not None and self.group.world_size > 1 and hasattr(self.engine, "set_option"):
            # The collective's kernels run beside the next step's GEMMs and take whole CUs (RCCL's gfx950 all-reduce
            # kernel: 248-256 VGPRs per wave, 37.6 KB LDS -- nothing of ours fits next to it), and a one-workgroup-per-
            # CU GEMM grid on fewer CUs needs a second round: measured 163 -> 219 us per step with EIGHT CUs taken.
            # So an overlapped data-parallel step CAN leave `comm_cus` CUs to the collective and launch its GEMMs
            # balanced on the rest (mdbn_planes.hip, "BALANCED launches"): MDBN_COMM_CUS /
            # fn.comm_cus, default 0 (dist.DEFAULT_COMM_CUS says why); bench.py --gpus N measures the choices.
            self.comm_cus = _dist.comm_cus() if self.overlap else 0      # handed to every cd_step call
            self.engine.set_option("gemm_cw", 1)        # exact-f32 fallback kernels: one MFMA wave per SIMD
        if self.overlap:
            for arr in self.rbm.params_speed:
                arr._sync_hook = self.flush

    def _data(self):
        if self.input_fn is not None:
            return self.input_fn()
        return self.engine.as_matrix(self.train_set_x)

    # -- host-resident training table (shared.HostTable): the minibatch rows are gathered over PCIe into one of two
    #    device staging buffers, on a side stream, ONE MINIBATCH AHEAD of the step that consumes them
    def _host_table(self):
        return self.train_set_x if (self.input_fn is None and isinstance(self.train_set_x, HostTable)) else None

    @staticmethod
    def _same_indexes(a, b):
        if a is b:
            return True
        if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
            # the prefetched view is kept alive, so its storage cannot have been reused: equal address, length and
            # type mean the same (never mutated) index list
            return a.data_ptr() == b.data_ptr() and a.numel() == b.numel() and a.dtype == b.dtype and a.device == b.device
        if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
            return False
        a, b = numpy.asarray(a), numpy.asarray(b)
        return a.shape == b.shape and bool((a == b).all())

    # A host-resident table feeds the step through a RowFeeder (engine.row_feeder, mdbn_feeder_*): CPU threads gather an
    # ANNOUNCED minibatch's rows into pinned staging, one SDMA copy moves them, three slots deep -- no kernel runs beside
    # the step (a PCIe gather kernel there costs it 60 us).  Unannounced minibatches are gathered on the spot by the PCIe
    # gather kernel, in stream order.
    def _feed_state(self):
        if self._staging is None:
            from collections import deque
            self._staging = {"feeder": None, "queue": deque(), "held": None, "now": None}
        return self._staging

    def _shard(self, n_global):
        return (0, n_global) if self.group is None else self.group.shard(n_global)

    def announce(self, batches, host_indexes=None):
        """Tell the step function which minibatches come next, in order (the trainers know the epoch's order up front):
        with a host-resident table their rows start moving now, up to two minibatches ahead of the step that reads them.
        ``batches``: the index lists / tensors exactly as the following calls will pass them; ``host_indexes``: the same
        values as host arrays, when the caller has them (saves one device-to-host copy of the list).  A pure hint: a call
        whose ``indexes`` are not the announced ones drops the remaining announcements.  No-op for device-resident data."""
        table = self._host_table()
        if table is None or not torch.cuda.is_available() or not hasattr(self.engine, "row_feeder"):
            return
        batches = list(batches)
        if not batches:
            return
        if host_indexes is None:
            if any(isinstance(b, torch.Tensor) and b.device.type != "cpu" for b in batches):
                flat = torch.cat([torch.as_tensor(b).reshape(-1).to(torch.int64) for b in batches]).cpu()   # one copy for all
                host_indexes = list(torch.split(flat, [len(b) for b in batches]))
            else:
                host_indexes = batches
        st = self._feed_state()
        self._drop_feed()                          # "these are the next calls": whatever was announced before is void
        shards = []
        for key, h in zip(batches, host_indexes):
            h = h if isinstance(h, torch.Tensor) else torch.from_numpy(numpy.ascontiguousarray(numpy.asarray(h), dtype=numpy.int64))
            h = h.to(torch.int64).reshape(-1)
            if len(h):      # as engine.index_tensor: numpy fancy indexing (negative values count from the end)
                lo_i, hi_i = int(h.min()), int(h.max())
                if lo_i < -len(table) or hi_i >= len(table):
                    raise IndexError("minibatch index out of range for %d rows: [%d, %d]" % (len(table), lo_i, hi_i))
                if lo_i < 0:
                    h = torch.where(h < 0, h + len(table), h)
            lo, hi = self._shard(len(h))
            shards.append((key, h[lo:hi].contiguous(), len(h), lo, hi))
        need = max(hi - lo for _, _, _, lo, hi in shards)
        fd = st["feeder"]
        if need > 0 and (fd is None or fd.max_rows < need or fd.host is not table.host):     # (set_value replaces .host)
            if fd is not None:
                fd.close()
            fd = st["feeder"] = self.engine.row_feeder(table.host, table.cols, need)
        for key, h, n_global, lo, hi in shards:
            ticket = fd.submit(h) if hi > lo else None
            # (the version counter of an announced index TENSOR: a buffer refilled in place afterwards is not the announced list)
            st["queue"].append((key, ticket, n_global, lo, hi, getattr(key, "_version", None)))

    def prefetch(self, indexes):
        """``announce([indexes])``: the next minibatch only."""
        self.announce([indexes])

    def _drop_feed(self):
        st = self._staging
        if st is None:
            return
        if st["held"] is not None:
            st["feeder"].release(st["held"])
            st["held"] = None
        if st["queue"]:
            st["queue"].clear()
            if st["feeder"] is not None:
                st["feeder"].cancel()

    def _staged(self, table, indexes):
        """Device rows of this step's minibatch shard: the fed buffer if ``indexes`` is what was announced next, else
        gathered now by the PCIe kernel."""
        st = self._feed_state()
        eng = self.engine
        if st["held"] is not None:                 # a step that raised before handing its rows back
            self._fed()
        if st["queue"]:
            key, ticket, n_global, lo, hi, key_version = st["queue"][0]
            if self._same_indexes(key, indexes) and key_version == getattr(indexes, "_version", None):
                st["queue"].popleft()
                if ticket is None:
                    return eng.alloc_matrix(0, table.cols, table.host.stride(0)), n_global, lo, hi
                buf = st["feeder"].acquire(ticket, hi - lo)
                st["held"] = ticket
                return buf, n_global, lo, hi
            self._drop_feed()                      # not what was announced: the remaining announcements are void
        n_global = len(table) if indexes is None else len(indexes)
        lo, hi = self._shard(n_global)
        buf = st["now"]
        if buf is None or buf.shape[0] != hi - lo:
            buf = st["now"] = eng.alloc_matrix(hi - lo, table.cols, table.host.stride(0))
        if hi > lo:
            idx = torch.arange(lo, hi, dtype=torch.int64, device=eng.device) if indexes is None else \
                eng.index_tensor(indexes, len(table))[lo:hi]
            table.rows(idx, out=buf)               # same stream as the steps: the previous reader of `buf` is ahead of it
        return buf, n_global, lo, hi

    def _fed(self):
        """The step that reads the fed rows has been enqueued: their slot may be refilled after it."""
        st = self._staging
        if st is not None and st["held"] is not None:
            st["feeder"].release(st["held"])
            st["held"] = None

    # -- deferred half of an overlapped step: speeds (and cost) from the reduced statistics
    def _complete_pending(self):
        if self._pending is None:
            return
        work, stats, hp, lazy = self._pending
        self._pending = None
        work.wait()
        rbm, p = self.rbm, self.plan
        cost = self.engine.apply_update(
            rbm.W.tensor, rbm.W_speed.tensor, p.W0.tensor if p.W0 is not None else None,
            rbm.hbias.tensor, rbm.hbias_speed.tensor, rbm.vbias.tensor, rbm.vbias_speed.tensor, stats,
            hp["lr"], p.lambda_1, p.lambda_2, p.weightcost, hp["momentum"], hp["batch_size"],
            hp["n_rows"], hp["cost_scale"], phase=1, ldv=hp["ldv"])
        lazy._value = cost

    def _resolve_cost(self, lazy):
        if self._pending is not None and self._pending[3] is lazy:
            self._complete_pending()
        if lazy._value is None:
            raise RuntimeError("cost of a step that was never completed")

    def flush(self):
        """Complete the deferred speed update of the last overlapped step (no-op otherwise)."""
        self._complete_pending()

    def __call__(self, indexes=None, momentum=0.0, lr=None, next_indexes=None):
        """``next_indexes``: the minibatch of the NEXT call, when the caller knows it (the trainers do: the epoch's order
        is drawn up front, dbn.py:446-458) -- a pure hint: the single-device plane path then gathers those rows inside
        this step's statistics kernel, a host-resident table starts moving them over PCIe; results never change."""
        p, rbm, eng = self.plan, self.rbm, self.engine
        if lr is None:
            lr = p.lr
        if isinstance(lr, Scalar):
            raise TypeError("step function needs lr= (learning rate is a symbolic input)")
        table = self._host_table()
        distributed = self.group is not None and self.group.world_size > 1
        staged_slot = None
        if table is not None and torch.cuda.is_available():
            # host-resident table: the shard's rows arrive in a device buffer (fed ahead of the step when the trainer
            # announced them); the step then runs on that buffer with the identity index
            data, n_global, lo, hi = self._staged(table, indexes)
            staged_slot = True
            idx = None
        else:
            data = self._data()
            n_global = data.shape[0] if indexes is None else len(indexes)
            # data-parallel shard of the minibatch: contiguous rows [lo, hi) of `indexes`
            lo, hi = 0, n_global
            if self.group is not None:
                lo, hi = self.group.shard(n_global)
            if indexes is None:
                idx = None if (lo == 0 and hi == data.shape[0]) else \
                    torch.arange(lo, hi, dtype=torch.int64, device=data.device)
            else:
                idx = eng.index_tensor(indexes, data.shape[0])[lo:hi]
        batch_size = p.batch_size if (p.batch_size is not None and not p.symbolic_grad) else n_global
        step = rbm._take_step()
        persistent = None
        if p.persistent is not None:
            persistent = p.persistent.tensor
            if persistent.shape[0] != n_global:
                raise ValueError("persistent chain has %d rows but the minibatch has %d "
                                 "(the reference fails the same way, rbm.py:416)"
                                 % (persistent.shape[0], n_global))
            if distributed:
                # PCD under data parallelism: chain row g belongs to the rank that owns minibatch row g
                # (rbm.py:308-311,369); a rank only ever reads and writes its own rows of the chain
                persistent = persistent[lo:hi]
        if not distributed and p.persistent is None and hi > lo:
            # single device: the whole step function in one library call (mdbn_cd_train_step)
            if len(self._fast) == 8 and idx is not None and staged_slot is None and not self.nan_guard and self._fast[7] == \
                    (batch_size, n_global, rbm.W.tensor.data_ptr(), rbm.hbias.tensor.data_ptr(), rbm.vbias.tensor.data_ptr()):
                # the argument structs of the previous call (small layers are host-bound: engine.cd_train_step_cached)
                out = eng.cd_train_step_cached(self._fast, data, idx, step, lr, momentum, next_indexes)
                if out is not None:
                    rbm._n_updates += 1
                    return out
            cost_scale = 1.0 / (n_global * rbm.n_visible) if rbm.gauss else 1.0 / n_global
            out = eng.cd_train_step(data, idx, rbm.W.tensor, rbm.W_speed.tensor,
                                    p.W0.tensor if p.W0 is not None else None,
                                    rbm.hbias.tensor, rbm.hbias_speed.tensor, rbm.vbias.tensor,
                                    rbm.vbias_speed.tensor, rbm.gauss, p.k,
                                    RngAddr(rbm.theano_rng.seed, rbm.stream_id, step, 0, lo),
                                    lr, p.lambda_1, p.lambda_2, p.weightcost, momentum, batch_size,
                                    n_global, cost_scale, sample_stats=p.symbolic_grad,
                                    **dict({"next_indexes": next_indexes} if (next_indexes is not None and idx is not None
                                                                              and staged_slot is None) else {},
                                           **({"cache_out": self._fast} if hasattr(eng, "cd_train_step_cached") else {})))
            if self._fast:        # (what else must stay as it was for the structs to be reused)
                self._fast.append((batch_size, n_global, rbm.W.tensor.data_ptr(), rbm.hbias.tensor.data_ptr(), rbm.vbias.tensor.data_ptr()))
            rbm._n_updates += 1
            if staged_slot is not None:
                self._fed()
                if next_indexes is not None and not self._staging["queue"]:
                    self.announce([next_indexes])
            if self.nan_guard:
                self._check_finite(out)
            return out
        slot = self._n_calls & 1 if self.overlap else 0     # the other buffer may still be reducing
        self._n_calls += 1
        if self._stats_slots[slot] is None:
            self._stats_slots[slot] = eng.new_stats_buffer(rbm.n_visible, rbm.n_hidden, data.stride(0),
                                                           rbm.W.tensor.stride(0))
        stats = self._stats_slots[slot]
        args = (rbm.W.tensor, rbm.W_speed.tensor, p.W0.tensor if p.W0 is not None else None,
                rbm.hbias.tensor, rbm.hbias_speed.tensor, rbm.vbias.tensor, rbm.vbias_speed.tensor)
        deferred_done = False
        if hi > lo:
            extra = {"comm_cus": self.comm_cus} if self.comm_cus else {}
            if self.overlap and self._pending is not None and int(self.fuse_deferred) >= (2 if self.comm_cus else 1) and \
                    getattr(eng, "cd_forward", None) is not None:
                # Overlapped order with the update INSIDE the statistics GEMM: the forward half of step t (everything that
                # reads theta(t)), then wait for the all-reduce of step t-1, then the statistics half, whose kernel applies
                # the deferred update of step t-1 (speeds from its reduced statistics, theta(t+1) from those speeds) beside
                # its own main loop -- bitwise what apply_update(phase=3) after cd_step does, one launch fewer.
                token = eng.cd_forward(data, idx, rbm.W.tensor, rbm.hbias.tensor, rbm.vbias.tensor, rbm.gauss, p.k,
                                       RngAddr(rbm.theano_rng.seed, rbm.stream_id, step, 0, lo),
                                       sample_stats=p.symbolic_grad, stats=stats,
                                       next_indexes=(eng.index_tensor(next_indexes, data.shape[0])[lo:hi]
                                                     if (next_indexes is not None and idx is not None and staged_slot is None
                                                         and len(next_indexes) == n_global) else None), **extra)
                pwork, pstats, hp, plazy = self._pending
                self._pending = None
                pwork.wait()
                _, _, pcost = eng.cd_statistics(token, deferred=args + (pstats, lr, p.lambda_1, p.lambda_2, p.weightcost,
                                                                        hp["momentum"], hp["batch_size"], hp["n_rows"],
                                                                        hp["cost_scale"], 3, hp["ldv"]))
                plazy._value = pcost
                deferred_done = True
            else:
                eng.cd_step(data, idx, rbm.W.tensor, rbm.hbias.tensor, rbm.vbias.tensor, rbm.gauss, p.k,
                            RngAddr(rbm.theano_rng.seed, rbm.stream_id, step, 0, lo),
                            persistent=persistent, sample_stats=p.symbolic_grad, stats=stats, **extra)
        else:                                  # this rank holds no row of a short minibatch
            stats.zero_()

        if p.persistent is not None:
            # PCD monitors the pseudo-likelihood (rbm.py:371) with pre-update parameters
            if hi > lo:
                cost = rbm._pseudo_likelihood_value(eng.gather_rows(data, idx) if idx is not None else data)
            else:
                cost = None
                rbm.bit_i_idx = (rbm.bit_i_idx + 1) % rbm.n_visible
            cost_scale = 0.0
            if distributed:
                # a mean over rows: every rank contributes (its mean * its rows) through the cost slot of the
                # statistics buffer, the update kernel divides the all-reduced sum by the global row count
                slot_idx = rbm.n_visible * rbm.W.tensor.stride(0) + rbm.W.tensor.stride(0) + data.stride(0)
                stats[slot_idx] = cost * float(hi - lo) if cost is not None else 0.0
                cost, cost_scale = None, 1.0 / n_global
        else:
            cost = None
            # rbm.py:480 (sum over units, mean over rows) / rbm.py:697 (mean over everything)
            cost_scale = 1.0 / (n_global * rbm.n_visible) if rbm.gauss else 1.0 / n_global
        if staged_slot is not None:
            self._fed()
            if next_indexes is not None and not self._staging["queue"]:
                self.announce([next_indexes])        # host-resident table: the next shard starts moving now
        if self.overlap:
            work = self.group.all_reduce_sum_async(stats, self.engine)
            if deferred_done:
                pass                                   # (step t-1's update went into this step's statistics GEMM)
            elif self._pending is not None:
                # speeds from the reduced statistics of step t-1, then theta(t+1) from those speeds:
                # phases 1 and 2 back to back touch the same arrays, so they run as ONE pass (phase 3;
                # the speed half keeps step t-1's momentum and divisors, the parameter half this lr)
                pwork, pstats, hp, plazy = self._pending
                self._pending = None
                pwork.wait()
                plazy._value = eng.apply_update(*args, pstats, lr, p.lambda_1, p.lambda_2, p.weightcost,
                                                hp["momentum"], hp["batch_size"], hp["n_rows"],
                                                hp["cost_scale"], phase=3, ldv=hp["ldv"])
            else:
                eng.apply_update(*args, stats, lr, p.lambda_1, p.lambda_2, p.weightcost, momentum,
                                 batch_size, n_global, cost_scale, phase=2, ldv=data.stride(0))   # theta(t+1)
            lazy = LazyCost(self, self._n_calls)
            self._pending = (work, stats, dict(lr=lr, momentum=momentum, batch_size=batch_size,
                                               n_rows=n_global, cost_scale=cost_scale, ldv=data.stride(0)), lazy)
            rbm._n_updates += 1
            return lazy

        if distributed:
            self.group.all_reduce_sum(stats, self.engine)
        out = eng.apply_update(*args, stats, lr, p.lambda_1, p.lambda_2, p.weightcost, momentum,
                               batch_size, n_global, cost_scale, ldv=data.stride(0))
        rbm._n_updates += 1
        if self.nan_guard:
            self._check_finite(cost if cost is not None else out)
        return cost if cost is not None else out

    def _check_finite(self, cost):
        rbm = self.rbm
        c = float(cost)
        bad = self.engine.count_nonfinite(rbm.W.tensor, rbm.W_speed.tensor, rbm.hbias.tensor, rbm.vbias.tensor) \
            if hasattr(self.engine, "count_nonfinite") else 0
        if bad or not numpy.isfinite(c):
            raise FloatingPointError("step %d of %s: cost = %r, %d non-finite parameter / speed values (learning rate "
                                     "too large for this data?)" % (rbm._n_updates, self.name or "step function", c, bad))


def function(updates, train_set_x=None, input_fn=None, name=None, data_parallel="auto", overlap=True):
    """Stand-in for ``theano.function([indexes, momentum, lr], cost, updates=updates,
    givens={x: train_set_x[indexes], rbm.momentum: momentum})`` (dbn.py:302-312)."""
    return StepFunction(updates, train_set_x, input_fn=input_fn, name=name,
                        data_parallel=data_parallel, overlap=overlap)


class RBM(object):
    """Restricted Boltzmann Machine (RBM) -- Bernoulli visible and hidden units."""

    gauss = False

    def __init__(self, input=None, n_visible=784, n_hidden=500, W=None, hbias=None,
                 vbias=None, numpy_rng=None, theano_rng=None, engine=None):
        """Same arguments as reference rbm.py:49-59.  ``W`` / ``hbias`` / ``vbias`` may be
        ``SharedArray`` objects shared with another network (a DBN's HiddenLayer),
        ndarrays, or None (initialised as rbm.py:100-131).  ``theano_rng`` is a
        ``mdbn_amd.RandomStreams``."""
        self.engine = engine if engine is not None else get_engine()
        self.n_visible = n_visible
        self.n_hidden = n_hidden

        if numpy_rng is None:
            numpy_rng = numpy.random.RandomState(1234)             # rbm.py:87-89
        if theano_rng is None:
            theano_rng = RandomStreams(numpy_rng.randint(2 ** 30))  # rbm.py:91-92

        if W is None:
            bound = 4 * numpy.sqrt(6. / (n_hidden + n_visible))     # rbm.py:100-107
            initial_W = numpy.asarray(numpy_rng.uniform(low=-bound, high=bound,
                                                        size=(n_visible, n_hidden)),
                                      dtype=numpy.float32)
            W = shared(initial_W, name='W', engine=self.engine, ld=weight_ld(self.engine, n_visible, n_hidden))
        if hbias is None:
            hbias = shared(numpy.zeros(n_hidden, dtype=numpy.float32), name='hbias', engine=self.engine)
        if vbias is None:
            vbias = shared(numpy.zeros(n_visible, dtype=numpy.float32), name='vbias', engine=self.engine)

        self.input = input                     # None: the data rows themselves
        self.W = shared(W, name='W', engine=self.engine)
        self.hbias = shared(hbias, name='hbias', engine=self.engine)
        self.vbias = shared(vbias, name='vbias', engine=self.engine)
        if self.W.shape != (n_visible, n_hidden):
            raise ValueError("W has shape %r, expected %r" % (self.W.shape, (n_visible, n_hidden)))
        self.theano_rng = theano_rng
        self.stream_id = theano_rng.new_stream()
        self.params = [self.W, self.hbias, self.vbias]

        self.momentum = 0.0                    # rbm.py:151; supplied per step-function call

        z = numpy.zeros
        self.W_speed = shared(z((n_visible, n_hidden), numpy.float32), name='W_speed', engine=self.engine,
                              ld=self.W.tensor.stride(0))        # (same layout as W: the update walks both)
        self.hbias_speed = shared(z(n_hidden, numpy.float32), name='hbias_speed', engine=self.engine)
        self.vbias_speed = shared(z(n_visible, numpy.float32), name='vbias_speed', engine=self.engine)
        self.params_speed = [self.W_speed, self.hbias_speed, self.vbias_speed]

        # rbm.py:415 captures W.get_value() when the update graph is built: a frozen
        # snapshot.  strict_reference=False uses the live W instead.
        self.strict_reference = True
        self._W0_snapshot = None               # the frozen weight-cost constant of the last get_cost_updates
        self._resume_W0 = None                 # ... restored from a checkpoint, consumed by the next one
        self.bit_i_idx = 0                     # rbm.py:425
        self._rng_step = 0
        self._n_updates = 0

    @property
    def Wt(self):
        """rbm.py:139: a transposed view; the kernels read W transposed in place."""
        return self.W.tensor.t()

    # ------------------------------------------------------------------ helpers
    def _take_step(self):
        s = self._rng_step
        self._rng_step += 1
        return s

    def _rng(self, draw=0):
        return RngAddr(self.theano_rng.seed, self.stream_id, self._take_step(), draw, 0)

    def _wrap(self, t):
        return None if t is None else SharedArray(None, engine=self.engine, _tensor=t)

    # ------------------------------------------------------------------ energies
    def free_energy(self, v_sample):
        ''' Function to compute the free energy (rbm.py:166-171) '''
        return self._wrap(self.engine.free_energy(as_tensor(v_sample, self.engine), self.W.tensor,
                                                  self.hbias.tensor, self.vbias.tensor, self.gauss))

    def free_energy_gap(self, train, test):
        """mean F(test) - mean F(train) (rbm.py:173-180); returns a Python float."""
        ft, fs = self.free_energy(train).tensor, self.free_energy(test).tensor
        return float(fs.mean() - ft.mean())

    def free_energies(self, train, test):
        """rbm.py:182-185, evaluated: two host vectors (as dbn.py:498-501 consumes them)."""
        return self.free_energy(train).get_value(), self.free_energy(test).get_value()

    # ------------------------------------------------------------------ propagation
    def propup(self, vis):
        '''[pre_sigmoid_activation, sigmoid(pre)] (rbm.py:187-199)'''
        pre, mean, _ = self.engine.propup(as_tensor(vis, self.engine), self.W.tensor, self.hbias.tensor,
                                          want_sample=False)
        return [self._wrap(pre), self._wrap(mean)]

    def sample_h_given_v(self, v0_sample):
        ''' [pre_sigmoid_h1, h1_mean, h1_sample] (rbm.py:201-213) '''
        pre, mean, sample = self.engine.propup(as_tensor(v0_sample, self.engine), self.W.tensor,
                                               self.hbias.tensor, rng=self._rng())
        return [self._wrap(pre), self._wrap(mean), self._wrap(sample)]

    def propdown(self, hid):
        '''[pre_sigmoid_activation, sigmoid(pre)] (rbm.py:215-227)'''
        pre, mean, _ = self.engine.propdown(as_tensor(hid, self.engine), self.W.tensor, self.vbias.tensor,
                                            gauss=False, rng=self._rng())
        return [self._wrap(pre), self._wrap(mean)]

    def sample_v_given_h(self, h0_sample):
        ''' [pre_sigmoid_v1, v1_mean, v1_sample] (rbm.py:229-240) '''
        pre, mean, sample = self.engine.propdown(as_tensor(h0_sample, self.engine), self.W.tensor,
                                                 self.vbias.tensor, gauss=False, rng=self._rng())
        return [self._wrap(pre), self._wrap(mean), self._wrap(sample)]

    def gibbs_hvh(self, h0_sample):
        ''' One Gibbs step starting from the hidden state (rbm.py:242-248) '''
        pre_sigmoid_v1, v1_mean, v1_sample = self.sample_v_given_h(h0_sample)
        pre_sigmoid_h1, h1_mean, h1_sample = self.sample_h_given_v(v1_sample)
        return [pre_sigmoid_v1, v1_mean, v1_sample,
                pre_sigmoid_h1, h1_mean, h1_sample]

    def gibbs_vhv(self, v0_sample):
        ''' One Gibbs step starting from the visible state (rbm.py:250-256) '''
        pre_sigmoid_h1, h1_mean, h1_sample = self.sample_h_given_v(v0_sample)
        pre_sigmoid_v1, v1_mean, v1_sample = self.sample_v_given_h(h1_sample)
        return [pre_sigmoid_h1, h1_mean, h1_sample,
                pre_sigmoid_v1, v1_mean, v1_sample]

    def gibbs_vhv_chain(self, v0_sample, n_steps):
        """``n_steps`` of ``gibbs_vhv`` as ONE device call (the ``theano.scan`` over gibbs_vhv of rbm.py:822-838):
        the six outputs of the LAST step, identical bit for bit to calling ``gibbs_vhv`` ``n_steps`` times,
        without the per-step allocations and launches' host overhead.  Consumes 2 * n_steps RNG steps."""
        step = self._rng_step
        self._rng_step += 2 * int(n_steps)
        if not hasattr(self.engine, "gibbs_chain"):          # checker engine: compose the eager steps
            self._rng_step = step
            out, v = None, v0_sample
            for _ in range(int(n_steps)):
                out = self.gibbs_vhv(v)
                v = out[5]
            return out
        out = self.engine.gibbs_chain(as_tensor(v0_sample, self.engine), self.W.tensor, self.hbias.tensor,
                                      self.vbias.tensor, self.gauss, int(n_steps),
                                      RngAddr(self.theano_rng.seed, self.stream_id, step, 0, 0),
                                      add_noise=self.gauss and not getattr(self, "error_free", True))
        return [self._wrap(t) for t in out]

    def make_sample_fn(self, persistent_vis_chain, n_steps=500):
        """The ``sample_fn`` of rbm.py:844-853: each call runs ``n_steps`` Gibbs steps from the persistent visible
        chain, stores ``vis_samples[-1]`` back into it and returns ``(vis_mfs[-1], vis_samples[-1])`` as host arrays."""
        chain = shared(persistent_vis_chain, engine=self.engine)

        def sample_fn():
            out = self.gibbs_vhv_chain(chain, n_steps)
            chain.set_value(out[5])
            return out[4].get_value(), out[5].get_value()
        return sample_fn

    # ------------------------------------------------------------------ CD-k / PCD-k
    def get_cost_updates(self, lr=0.1, k=1, lambda_1=0.0, lambda_2=0.0, weightcost=0.0,
                         batch_size=None, persistent=None, symbolic_grad=False):
        """One step of CD-k or PCD-k (rbm.py:258-376).  Returns ``(cost, updates)``; pass
        ``updates`` to ``mdbn_amd.function`` to obtain the step function.

        ``lr`` may be a float or a ``Scalar`` (then the step function takes ``lr=``).
        ``persistent``: None for CD, a SharedArray [batch_size, n_hidden] for PCD."""
        if symbolic_grad and self.gauss and not getattr(self, "error_free", True):
            raise NotImplementedError("symbolic_grad with a noisy GRBM (error_free=False) is not supported")
        if symbolic_grad:
            weightcost = 0.0           # tensor.grad of the free-energy difference has no such term
        W0 = None
        if weightcost != 0.0 and self.strict_reference:
            # rbm.py:415; same padded layout as W (a plain clone() would drop the leading dimension)
            snap = self.engine.alloc_matrix(self.n_visible, self.n_hidden, self.W.tensor.stride(0))
            if self._resume_W0 is not None:
                # resumed run: the constant is the W of the ORIGINAL graph construction (checkpoint.py)
                snap.copy_(as_tensor(self._resume_W0, self.engine))
                self._resume_W0 = None
            else:
                snap.copy_(self.W.tensor)
            W0 = SharedArray(None, engine=self.engine, _tensor=snap)
            self._W0_snapshot = W0
        if persistent is not None:
            persistent = shared(persistent, engine=self.engine)
            if persistent.tensor.stride(0) != self.W.tensor.stride(0):
                t = self.engine.alloc_matrix(persistent.shape[0], persistent.shape[1], self.W.tensor.stride(0))
                t.copy_(persistent.tensor)
                persistent.tensor = t
        updates = UpdatePlan(self, lr, k, lambda_1, lambda_2, weightcost, batch_size, persistent, W0,
                             symbolic_grad=symbolic_grad)
        cost = CostHandle('pseudo_likelihood' if persistent is not None else 'reconstruction')
        return cost, updates

    def _pseudo_likelihood_value(self, x):
        """rbm.py:421-447 on the current minibatch; advances bit_i_idx (rbm.py:445)."""
        eng = self.engine
        xi = eng.round_flip(x)                                        # tensor.round
        fe_xi = eng.free_energy(xi, self.W.tensor, self.hbias.tensor, self.vbias.tensor, self.gauss)
        xi_flip = eng.round_flip(x, self.bit_i_idx)                   # bit i flipped (rbm.py:436)
        fe_flip = eng.free_energy(xi_flip, self.W.tensor, self.hbias.tensor, self.vbias.tensor, self.gauss)
        cost = eng.pl_cost(fe_xi, fe_flip, self.n_visible)            # rbm.py:442
        self.bit_i_idx = (self.bit_i_idx + 1) % self.n_visible
        return cost

    def get_pseudo_likelihood_cost(self, v):
        """Stochastic approximation to the pseudo-likelihood (rbm.py:421-447) of rows ``v``."""
        return float(self._pseudo_likelihood_value(as_tensor(v, self.engine)))

    def get_reconstruction_cost(self, pre_sigmoid_nv, v0):
        """rbm.py:449-482 evaluated on given arrays (monitoring helper; the step function
        computes the same quantity fused into the last propdown)."""
        return float(self.engine.recon_cost(as_tensor(pre_sigmoid_nv, self.engine), as_tensor(v0, self.engine),
                                            gauss=False))

    # ------------------------------------------------------------------ stand-alone trainer
    def training(self, train_set_x, validation_set_x, training_epochs, batch_size=10,
                 learning_rate=0.1, k=1, initial_momentum=0.0, final_momentum=0.0,
                 weightcost=0.0, lambda_2=0.0, persistent=True, display_fn=None, graph_output=False):
        """rbm.py:484-520 (note: like the reference, ``lambda_2`` is accepted but not used)."""
        if persistent:
            persistent_chain = shared(numpy.zeros((batch_size, self.n_hidden), dtype=numpy.float32),
                                      engine=self.engine)               # rbm.py:496-498
        else:
            persistent_chain = None
        cost, updates = self.get_cost_updates(lr=learning_rate, k=k, weightcost=weightcost,
                                              batch_size=batch_size, persistent=persistent_chain)
        return self.learn_model(train_set_x=train_set_x, validation_set_x=validation_set_x,
                                training_epochs=training_epochs, batch_size=batch_size,
                                initial_momentum=initial_momentum, final_momentum=final_momentum,
                                cost=cost, updates=updates, display_fn=display_fn,
                                graph_output=graph_output)

    def learn_model(self, train_set_x, validation_set_x, training_epochs, batch_size,
                    initial_momentum, final_momentum, cost, updates, display_fn, graph_output,
                    verbose=True, shuffle_rng=None):
        """Epoch loop of rbm.py:522-629.  ``display_fn`` / ``graph_output`` are accepted and
        ignored (plotting is outside the engine).  Returns the per-epoch (cost, gap) list."""
        train_set_x = shared(train_set_x, engine=self.engine)
        validation_set_x = shared(validation_set_x, engine=self.engine) if validation_set_x is not None else None
        train_rbm = function(updates, train_set_x, name='train_rbm')
        n_train_data = train_set_x.shape[0]
        n_val = validation_set_x.shape[0] if validation_set_x is not None else 0
        history = []
        start_time = timeit.default_timer()
        momentum = initial_momentum
        for epoch in range(training_epochs):
            if epoch == 6:                                               # rbm.py:584-585
                momentum = final_momentum
            _, minibatches = get_minibatches_idx(n_train_data, batch_size, shuffle=True, rng=shuffle_rng)
            dev_idx = self.engine.index_tensor(numpy.concatenate(minibatches))
            # costs are 0-d device scalars (views into the engine's cost blocks): kept unread for the whole epoch -- no
            # synchronisation inside it -- and read back with one copy per block at its end; the mean is formed on the host
            # in float64, as the reference's numpy.mean over its per-minibatch floats (rbm.py:587-592)
            costs = []
            views = list(torch.split(dev_idx, [len(b) for b in minibatches]))
            train_rbm.announce(views, host_indexes=minibatches)      # the epoch's order (a host-resident table starts feeding)
            for b_i in range(len(minibatches)):
                # the next minibatch of the epoch as a hint (gathered inside this step's statistics kernel)
                nxt = views[b_i + 1] if b_i + 1 < len(minibatches) else None
                costs.append(train_rbm(views[b_i], momentum, next_indexes=nxt))
            train_rbm.flush()
            mean_cost = float(numpy.sum(numpy.asarray(self.engine.cost_values(costs), dtype=numpy.float64))) / len(minibatches)
            feg = None
            if n_val:
                # rbm.py:597: gap between the first n_val training rows and the validation set
                feg = self.free_energy_gap(train_set_x[numpy.arange(n_val)], validation_set_x)
            if verbose:
                print('Training epoch %d, cost is ' % epoch, mean_cost)
                print('Free energy gap is ', feg)
            history.append((mean_cost, feg))
        end_time = timeit.default_timer()
        if verbose:
            print('Training took %f minutes' % ((end_time - start_time) / 60.))
        return history


class GRBM(RBM):
    """Gaussian-Bernoulli RBM, unit variance, mean-field visibles (rbm.py:631-728)."""

    gauss = True

    def __init__(self, input=None, n_visible=784, n_hidden=500, W=None, hbias=None, vbias=None,
                 numpy_rng=None, theano_rng=None, error_free=True, engine=None):
        super(GRBM, self).__init__(input, n_visible, n_hidden, W, hbias, vbias, numpy_rng,
                                   theano_rng, engine=engine)
        self.error_free = error_free

    def sample_v_given_h(self, h0_sample):
        ''' [v1_mean, v1_mean, v1_sample]; linear mean, optional N(0,1) noise (rbm.py:647-660) '''
        pre, mean, sample = self.engine.propdown(as_tensor(h0_sample, self.engine), self.W.tensor,
                                                 self.vbias.tensor, gauss=True,
                                                 add_noise=not self.error_free, rng=self._rng())
        return [self._wrap(mean), self._wrap(mean), self._wrap(sample)]

    def gibbs_hvh(self, h0_sample):
        ''' Gibbs step from the hidden state; h1 from the visible MEAN (rbm.py:662-671) '''
        pre_sigmoid_v1, v1_mean, v1_sample = self.sample_v_given_h(h0_sample)
        pre_sigmoid_h1, h1_mean, h1_sample = self.sample_h_given_v(v1_mean)
        return [pre_sigmoid_v1, v1_mean, v1_sample,
                pre_sigmoid_h1, h1_mean, h1_sample]

    def gibbs_vhv(self, v0_sample):
        ''' Gibbs step from the visible state; v1 from the hidden MEAN (rbm.py:673-682) '''
        pre_sigmoid_h1, h1_mean, h1_sample = self.sample_h_given_v(v0_sample)
        pre_sigmoid_v1, v1_mean, v1_sample = self.sample_v_given_h(h1_mean)
        return [pre_sigmoid_h1, h1_mean, h1_sample,
                pre_sigmoid_v1, v1_mean, v1_sample]

    def get_reconstruction_cost(self, pre_sigmoid_nv, v0):
        """mean((sigmoid(v1_mean) - v0)^2) over samples and features (rbm.py:690-699)."""
        return float(self.engine.recon_cost(as_tensor(pre_sigmoid_nv, self.engine), as_tensor(v0, self.engine),
                                            gauss=True))

    def training(self, train_set_x, validation_set_x, training_epochs, batch_size=10,
                 learning_rate=0.01, k=1, initial_momentum=0.0, final_momentum=0.0,
                 weightcost=0.0, lambda_1=0.0, lambda_2=0.1, persistent=False,
                 display_fn=None, graph_output=False):
        """rbm.py:701-728: always CD (``persistent`` is ignored, as in the reference)."""
        cost, updates = self.get_cost_updates(lr=learning_rate, k=k, lambda_1=lambda_1,
                                              lambda_2=lambda_2, weightcost=weightcost,
                                              batch_size=batch_size)
        return self.learn_model(train_set_x=train_set_x, validation_set_x=validation_set_x,
                                training_epochs=training_epochs, batch_size=batch_size,
                                initial_momentum=initial_momentum, final_momentum=final_momentum,
                                cost=cost, updates=updates, display_fn=display_fn,
                                graph_output=graph_output)
