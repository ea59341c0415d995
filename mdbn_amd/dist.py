"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" is
RCCL on ROCm, riding xGMI inside a node).  The reference is single-process; DP is added by
this engine: rows of each minibatch are sharded contiguously over the ranks and
the packed CD statistics are sum-all-reduced once per step."""
import os

import torch
import torch.distributed as td


class Group(object):
    def __init__(self, pg=None):
        self.pg = pg
        self.rank = td.get_rank(pg)
        self.world_size = td.get_world_size(pg)

    def shard(self, n):
        """Contiguous rows [lo, hi) of an n-row minibatch owned by this rank."""
        return shard_bounds(n, self.rank, self.world_size)

    def all_reduce_sum(self, tensor):
        td.all_reduce(tensor, op=td.ReduceOp.SUM, group=self.pg)
        return tensor

    def all_reduce_sum_async(self, tensor):
        """Start the sum all-reduce and return the work handle; ``handle.wait()`` makes the
        current stream (not the host, on RCCL) wait for the result."""
        return td.all_reduce(tensor, op=td.ReduceOp.SUM, group=self.pg, async_op=True)


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
        if backend is None:
            # MDBN_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals (RCCL needs
            # one GPU per rank); the default on GPUs is nccl = RCCL over xGMI
            backend = os.environ.get("MDBN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        td.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world
