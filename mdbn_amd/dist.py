"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm, riding xGMI inside a node).  The reference is single-process; DP is added by
this engine: rows of each minibatch are sharded contiguously over the ranks and
the packed CD statistics are sum-all-reduced once per step."""
import os

import torch
import torch.distributed as td


class _StreamWork(object):
    """Handle of a collective enqueued on a side HIP stream: ``wait()`` makes the CURRENT stream wait for it."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class _NoWork(object):
    """Handle of a collective that was not issued (``Group.stub_collective``)."""
    event = None

    def wait(self):
        pass

    def is_completed(self):
        return True


class _WireWork(object):
    """Handle of an all-reduce that ran on a bfloat16 copy (``Group.wire_bf16``): ``wait()`` waits for the collective and
    widens the sums back into the float32 statistics buffer on the current stream (the engine's mdbn_bf16_to_f32)."""

    def __init__(self, work, wire, tensor, engine):
        self.work, self.wire, self.tensor, self.engine = work, wire, tensor, engine

    def wait(self):
        self.work.wait()
        self.engine.widen_bf16(self.wire, self.tensor)

    def is_completed(self):
        return self.work.is_completed()


class Group(object):
    """The ranks of one data-parallel job.

    The collective (``MDBN_DP_COLLECTIVE`` = ``auto`` | ``capi`` | ``torch``, or ``native=None | True | False``):

    * ``auto`` (default): the library's own RCCL collective behind the C-ABI -- ``mdbn_allreduce_stats`` on a communicator made
      by ``mdbn_comm_unique_id`` / ``mdbn_comm_init_rank`` (include/mdbn_hip.h; torch.distributed only hands the 128-byte id
      to the ranks) -- ONCE IT HAS PROVEN ALIVE: the communicator is built and one small all-reduce is run and checked on a
      helper thread against a deadline (``MDBN_DP_CAPI_DEADLINE_S``, 60 s), every rank reports, and the ranks agree (MIN over
      torch.distributed).  Anything else -- no HIP engine, a backend other than RCCL (gloo rehearsals), librccl missing, an init
      that hangs or a wrong sum on ANY rank -- and every rank falls back to ``torch.distributed``'s all-reduce (backend nccl
      = the same RCCL); ``native_error`` then says why.
    * ``capi``: the C-ABI collective or an exception.  ``torch``: torch.distributed's, no probe."""

    def __init__(self, pg=None, native=None):
        self.pg = pg
        self.rank = td.get_rank(pg)
        self.world_size = td.get_world_size(pg)
        if native is None:
            native = {"capi": True, "torch": False}.get(os.environ.get("MDBN_DP_COLLECTIVE", "auto"))
        self.native = native                  # True: C-ABI or raise; False: torch.distributed; None: C-ABI once proven alive
        self.native_error = None              # why `auto` fell back (None: it did not, or has not been decided)
        self._auto_off = False                # auto mode decided against the C-ABI collective (for good: never probed twice)
        self._comm_engine = None
        self._side = None
        self._wire = {}                       # persistent bfloat16 wire buffers, one per statistics tensor
        # MEASUREMENT ONLY (bench.py `distributed.exposed_comm_us`): True makes both all-reduce calls no-ops, so that the same
        # step can be timed without its collective.  The statistics are then those of the local shard: results are wrong.
        self.stub_collective = False
        # SURVEY section 5's bf16 wire format, OPT-IN and NOT a parity path (``MDBN_WIRE_BF16=1``): the packed statistics are
        # narrowed to bfloat16 for the all-reduce (half the bytes on xGMI: 8.4 instead of 16.8 MB per rank at c2) and the sums
        # widened back -- 8 significant bits per addend, so parameters drift from the float32-wire run by ~1e-3 relative per
        # step.  A REPORTING mode like "bf16_inputs": never used for a parity claim; torch.distributed collective only.
        self.wire_bf16 = os.environ.get("MDBN_WIRE_BF16", "0") == "1"
        # every rank must run the same wire format: ranks that disagree would issue all-reduces of different types and sizes
        # (a hang or garbage on RCCL).  Checked once, over the group itself.
        flag = torch.tensor([int(self.wire_bf16), -int(self.wire_bf16)], dtype=torch.int32, device=self._flag_device())
        td.all_reduce(flag, op=td.ReduceOp.MAX, group=self.pg)
        if int(flag[0]) != -int(flag[1]):
            raise RuntimeError("MDBN_WIRE_BF16 differs between the ranks of this group (rank %d has %d): set it on every rank or on none"
                               % (self.rank, int(self.wire_bf16)))
        if self.wire_bf16 and self.native is True:
            raise RuntimeError("MDBN_WIRE_BF16=1 needs torch.distributed's collective: it cannot be combined with "
                               "MDBN_DP_COLLECTIVE=capi (mdbn_allreduce_stats sums float32)")

    def _flag_device(self):
        return torch.device("cuda", torch.cuda.current_device()) if td.get_backend(self.pg) == "nccl" else torch.device("cpu")

    def shard(self, n):
        """Contiguous rows [lo, hi) of an n-row minibatch owned by this rank."""
        return shard_bounds(n, self.rank, self.world_size)

    # -- the library's own RCCL communicator
    def _native_possible(self, engine):
        """Can this engine / backend run the C-ABI collective at all?  (a HIP engine, one GPU per rank = backend nccl)"""
        return engine is not None and hasattr(engine, "ctx") and torch.cuda.is_available() and td.get_backend(self.pg) == "nccl"

    def _build_native(self, engine, ident):
        """Communicator + one checked all-reduce of 1024 ones, on the calling (helper) thread.  Raises on any failure."""
        import ctypes as C
        from . import _lib
        _lib.check(engine.lib.mdbn_comm_init_rank(engine.ctx, ident, self.world_size, self.rank), "mdbn_comm_init_rank")
        side = torch.cuda.Stream(device=engine.device)
        buf = torch.ones(1024, dtype=torch.float32, device=engine.device)
        torch.cuda.synchronize(engine.device)
        _lib.check(engine.lib.mdbn_allreduce_stats(engine.ctx, C.c_void_p(side.cuda_stream), C.c_void_p(buf.data_ptr()), buf.numel()),
                   "mdbn_allreduce_stats")
        side.synchronize()
        got = buf.cpu()
        if float(got.min()) != float(self.world_size) or float(got.max()) != float(self.world_size):
            raise RuntimeError("mdbn_allreduce_stats summed %d ones to [%g, %g]" % (self.world_size, float(got.min()), float(got.max())))
        return side

    def probe_native(self, engine, deadline_s=None):
        """Build and prove the C-ABI collective (see the class comment); all ranks return the same verdict.  On success the
        communicator stays installed (``collective`` names it); on failure it is torn down and ``native_error`` is set."""
        import ctypes as C
        import threading
        from . import _lib
        if self._comm_engine is engine:
            return True
        if deadline_s is None:
            deadline_s = float(os.environ.get("MDBN_DP_CAPI_DEADLINE_S", "60"))
        ok, err, box = 1, None, {}
        if not self._native_possible(engine):
            ok, err = 0, "needs a HipEngine and the nccl (RCCL) backend, one GPU per rank"
        ident = [None]
        if ok:
            try:
                if self.rank == 0:
                    buf = C.create_string_buffer(128)
                    _lib.check(engine.lib.mdbn_comm_unique_id(buf), "mdbn_comm_unique_id")
                    ident[0] = buf.raw
            except Exception as exc:                 # (librccl.so missing on rank 0: the others must still learn of it)
                err = repr(exc)[:200]
        if self._native_possible(engine):
            td.broadcast_object_list(ident, src=0, group=self.pg)
            if ident[0] is None:
                ok, err = 0, err or "rank 0 could not make an RCCL id"
        if ok:
            def work():
                try:
                    box["side"] = self._build_native(engine, ident[0])
                except Exception as exc:
                    box["err"] = repr(exc)[:200]
            t = threading.Thread(target=work, daemon=True)
            t.start()
            t.join(deadline_s)
            if t.is_alive():
                ok, err = 0, "communicator init / first all-reduce did not finish within %.0f s (helper thread abandoned)" % deadline_s
            elif "err" in box:
                ok, err = 0, box["err"]
        flag = torch.tensor([ok], dtype=torch.int32, device=self._flag_device())
        td.all_reduce(flag, op=td.ReduceOp.MIN, group=self.pg)      # through torch's own communicator
        agreed = bool(int(flag.item()))
        if agreed:
            self._comm_engine, self._side, self.native_error = engine, box["side"], None
            return True
        if ok and "side" in box:                     # this rank was fine, another was not: drop the communicator again
            try:
                engine.lib.mdbn_comm_destroy(engine.ctx)
            except Exception:
                pass
        self.native_error = err or "another rank could not bring the C-ABI collective up"
        return False

    def _native(self, engine):
        if self.native is False or self.wire_bf16:
            return False
        if self._comm_engine is engine and engine is not None:
            return True
        if self.native is None:
            if self._auto_off or not self._native_possible(engine):
                return False                       # (gloo rehearsals, CPU engines: nothing to probe, identical on every rank)
            if self.probe_native(engine):
                return True
            self._auto_off = True
            return False
        if not self._native_possible(engine):
            return False
        if not self.probe_native(engine):
            from . import _lib
            raise _lib.MdbnError("MDBN_DP_COLLECTIVE=capi: " + str(self.native_error))
        return True

    def _native_launch(self, tensor, engine, stream):
        import ctypes as C
        from . import _lib
        _lib.check(engine.lib.mdbn_allreduce_stats(engine.ctx, C.c_void_p(stream.cuda_stream),
                                                   C.c_void_p(tensor.data_ptr()), tensor.numel()),
                   "mdbn_allreduce_stats")

    @property
    def collective(self):
        return "mdbn_allreduce_stats (RCCL through the C-ABI)" if self._comm_engine is not None and self.native is not False else \
            "torch.distributed all_reduce (%s)" % td.get_backend(self.pg)

    def _wire_of(self, tensor, engine):
        if engine is None or not hasattr(engine, "narrow_bf16"):
            raise RuntimeError("MDBN_WIRE_BF16=1 needs the engine's conversion kernels (mdbn_f32_to_bf16)")
        key = (tensor.data_ptr(), tensor.numel())
        wire = self._wire[key] = engine.narrow_bf16(tensor, self._wire.get(key))
        return wire

    def all_reduce_sum(self, tensor, engine=None):
        if self.stub_collective:
            return tensor
        if self._native(engine):
            self._native_launch(tensor, engine, torch.cuda.current_stream(engine.device))
            return tensor
        if self.wire_bf16:
            wire = self._wire_of(tensor, engine)
            td.all_reduce(wire, op=td.ReduceOp.SUM, group=self.pg)
            engine.widen_bf16(wire, tensor)
            return tensor
        td.all_reduce(tensor, op=td.ReduceOp.SUM, group=self.pg)
        return tensor

    def all_reduce_sum_async(self, tensor, engine=None):
        """Start the sum all-reduce and return the work handle; ``handle.wait()`` makes the
        current stream (not the host, on RCCL) wait for the result."""
        if self.stub_collective:
            return _NoWork()
        if self._native(engine):
            cur = torch.cuda.current_stream(engine.device)
            self._side.wait_stream(cur)                    # the statistics are complete on the compute stream
            self._native_launch(tensor, engine, self._side)
            ev = torch.cuda.Event()
            ev.record(self._side)
            return _StreamWork(ev)
        if self.wire_bf16:
            wire = self._wire_of(tensor, engine)
            return _WireWork(td.all_reduce(wire, op=td.ReduceOp.SUM, group=self.pg, async_op=True), wire, tensor, engine)
        return td.all_reduce(tensor, op=td.ReduceOp.SUM, group=self.pg, async_op=True)


# 0 until a multi-GPU run says otherwise.  One GPU beside a stand-in with RCCL's footprint: if the
# collective really holds 32 CUs for most of the step, 32 beats 0 by 4 % (203 vs 212 us); if it is short or narrow, 0 beats
# 32 by 29 % (146 vs 188 us).  bench.py --gpus N measures 0 / 8 / 16 / 32 / 64 and RCCL channel caps and quotes the best.
DEFAULT_COMM_CUS = 0


def comm_cus():
    """CUs an overlapped data-parallel step leaves to the collective (``MDBN_COMM_CUS``; e.g. 32 of 256: the step's GEMMs
    then run balanced on 224 workgroups, a multiple of the tile counts at the c2 shape)."""
    return max(0, min(192, int(os.environ.get("MDBN_COMM_CUS", DEFAULT_COMM_CUS))))


def shard_bounds(n, rank, world_size):
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def default_group():
    """The world group if torch.distributed is initialised with more than one rank."""
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        return Group()
    return None


def init_from_env(backend=None):
    """Initialise from torchrun's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*).
    Returns (rank, local_rank, world_size); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("MDBN_RCCL_MAX_CHANNELS"):
            # opt-in: one RCCL channel = one workgroup = one CU, so capping the channels at `comm_cus` keeps the
            # collective within the CUs an overlapped step leaves it.  NOT applied by default: the 32 / 224 split was
            # sized against a stand-in kernel on one GPU and a cap may throttle the 16.8 MB
            # all-reduce; bench.py --gpus N measures the step at several comm_cus and reports RCCL's own choice.
            os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ["MDBN_RCCL_MAX_CHANNELS"])
        if backend is None:
            # MDBN_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals (RCCL needs
            # one GPU per rank); the default on GPUs is nccl = RCCL over xGMI
            backend = os.environ.get("MDBN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        td.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world
