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
    widens the sums back into the float32 statistics buffer on the current stream."""

    def __init__(self, work, wire, tensor):
        self.work, self.wire, self.tensor = work, wire, tensor

    def wait(self):
        self.work.wait()
        self.tensor.copy_(self.wire)

    def is_completed(self):
        return self.work.is_completed()


class Group(object):
    """The ranks of one data-parallel job.  The collective is ``torch.distributed``'s all-reduce (backend nccl =
    RCCL) by default; with ``MDBN_DP_COLLECTIVE=capi`` (or ``native=True``) it is the library's own
    ``mdbn_allreduce_stats`` on a communicator created through ``mdbn_comm_init_rank`` (include/mdbn_hip.h),
    with torch.distributed used only to hand the 128-byte RCCL id to the ranks."""

    def __init__(self, pg=None, native=None):
        self.pg = pg
        self.rank = td.get_rank(pg)
        self.world_size = td.get_world_size(pg)
        self.native = (os.environ.get("MDBN_DP_COLLECTIVE") == "capi") if native is None else bool(native)
        self._comm_engine = None
        self._side = None
        # MEASUREMENT ONLY (bench.py `distributed.exposed_comm_us`): True makes both all-reduce calls no-ops, so that the same
        # step can be timed without its collective.  The statistics are then those of the local shard: results are wrong.
        self.stub_collective = False
        # SURVEY section 5's bf16 wire format, OPT-IN and NOT a parity path (``MDBN_WIRE_BF16=1``): the packed statistics are
        # narrowed to bfloat16 for the all-reduce (half the bytes on xGMI: 8.4 instead of 16.8 MB per rank at c2) and the sums
        # widened back -- 8 significant bits per addend, so parameters drift from the float32-wire run by ~1e-3 relative per
        # step.  A REPORTING mode like "bf16_inputs": never used for a parity claim; torch.distributed collective only.
        self.wire_bf16 = os.environ.get("MDBN_WIRE_BF16", "0") == "1"

    def shard(self, n):
        """Contiguous rows [lo, hi) of an n-row minibatch owned by this rank."""
        return shard_bounds(n, self.rank, self.world_size)

    # -- the library's own RCCL communicator
    def _native(self, engine):
        if not self.native or engine is None or not hasattr(engine, "ctx") or not torch.cuda.is_available():
            return False
        if self._comm_engine is not engine:
            import ctypes as C
            from . import _lib
            ident = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                _lib.check(engine.lib.mdbn_comm_unique_id(buf), "mdbn_comm_unique_id")
                ident[0] = buf.raw
            td.broadcast_object_list(ident, src=0, group=self.pg)
            _lib.check(engine.lib.mdbn_comm_init_rank(engine.ctx, ident[0], self.world_size, self.rank),
                       "mdbn_comm_init_rank")
            self._comm_engine = engine
            self._side = torch.cuda.Stream(device=engine.device)
        return True

    def _native_launch(self, tensor, engine, stream):
        import ctypes as C
        from . import _lib
        _lib.check(engine.lib.mdbn_allreduce_stats(engine.ctx, C.c_void_p(stream.cuda_stream),
                                                   C.c_void_p(tensor.data_ptr()), tensor.numel()),
                   "mdbn_allreduce_stats")

    @property
    def collective(self):
        return "mdbn_allreduce_stats (RCCL through the C-ABI)" if self._comm_engine is not None else \
            "torch.distributed all_reduce (%s)" % td.get_backend(self.pg)

    def all_reduce_sum(self, tensor, engine=None):
        if self.stub_collective:
            return tensor
        if self._native(engine):
            self._native_launch(tensor, engine, torch.cuda.current_stream(engine.device))
            return tensor
        if self.wire_bf16:
            wire = tensor.to(torch.bfloat16)
            td.all_reduce(wire, op=td.ReduceOp.SUM, group=self.pg)
            tensor.copy_(wire)
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
            wire = tensor.to(torch.bfloat16)
            return _WireWork(td.all_reduce(wire, op=td.ReduceOp.SUM, group=self.pg, async_op=True), wire, tensor)
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
