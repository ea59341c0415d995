"""SharedArray: the stand-in for ``theano.shared`` variables (reference call sites:
utils.py:113-115, dbn.py:172-173,384-406, rbm.py:109-131,153-162, AMLsm2.py:81-103).

Wraps a device tensor (a padded matrix or a vector owned by an engine) and offers the
observable protocol the reference's callers use: ``get_value(borrow=)``,
``set_value``, ``.name``, plus numpy conversion."""
import numpy
import torch

from .engine import get_engine


class SharedArray(object):
    def __init__(self, value, name=None, engine=None, _tensor=None, ld=None):
        self.engine = engine if engine is not None else get_engine()
        self.name = name
        self._sync_hook = None      # called before host reads (completes deferred device work)
        self.version = 0            # bumped by every set_value (caches of derived data key on it)
        if _tensor is not None:
            self.tensor = _tensor
        elif ld is not None:        # a weight matrix on a wider leading dimension (Engine.weight_ld); pad columns zero
            src = self.engine.to_device(getattr(value, "tensor", value))
            self.tensor = self.engine.alloc_matrix(src.shape[0], src.shape[1], ld=ld)
            self.tensor.copy_(src)
        else:
            self.tensor = self.engine.to_device(getattr(value, "tensor", value))

    # --- theano.shared protocol
    def get_value(self, borrow=False, return_internal_type=False):
        """Host copy (float32 ndarray).  ``borrow`` is accepted for signature parity; device
        memory cannot be borrowed by numpy, so a copy is always returned."""
        if self._sync_hook is not None:
            self._sync_hook()
        if return_internal_type:
            return self.tensor
        return self.engine.to_numpy(self.tensor)

    def set_value(self, value, borrow=False):
        self.version += 1
        new = getattr(value, "tensor", value)
        if not isinstance(new, torch.Tensor):
            new = torch.from_numpy(numpy.ascontiguousarray(new, dtype=numpy.float32))
        if tuple(new.shape) != tuple(self.tensor.shape):
            self.tensor = self.engine.to_device(new)        # (a new shape: the default leading dimension again)
        else:
            self.tensor.copy_(new.to(self.tensor.device))

    # --- conveniences
    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def ndim(self):
        return self.tensor.dim()

    dtype = numpy.dtype("float32")

    def __len__(self):
        return self.tensor.shape[0]

    def __array__(self, dtype=None, copy=None):
        a = self.get_value()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, item):
        """Row selection ``train_set_x[indexes]`` (dbn.py:307): a device gather."""
        if isinstance(item, slice):
            return SharedArray(None, engine=self.engine, _tensor=self.engine.as_matrix(self.tensor[item]))
        return SharedArray(None, engine=self.engine, _tensor=self.engine.gather_rows(self.tensor, item))

    def __repr__(self):
        return "SharedArray(name=%r, shape=%r)" % (self.name, self.shape)


class HostTable(SharedArray):
    """A training table that stays in PINNED HOST memory (``shared(x, resident="host")``): for tables that should
    not, or cannot, be uploaded whole -- the reference keeps ``train_set_x`` in a ``theano.shared`` on the device
    (utils.py:113-115) and so does the default here.

    The pinned pages are device-accessible, so the minibatch gather of dbn.py:307 is still the library's gather
    kernel (``mdbn_gather_rows``), now reading its source rows over PCIe: ``rows(indexes, out=)`` enqueues it on
    the current stream.  A step function double-buffers that gather one minibatch ahead on a side stream
    (``StepFunction.prefetch``); nothing on the host touches the rows.  ``.tensor`` is a device MIRROR made on first
    use for the cold paths that want the whole matrix (validation rows, a plain ``get_output``); the training step
    and the DBN's lower-layer cache never ask for it."""

    def __init__(self, value, name=None, engine=None):
        self.engine = engine if engine is not None else get_engine()
        self.name = name
        self._sync_hook = None
        self.version = 0
        self._mirror = None
        self._set_host(value)

    def _set_host(self, value):
        from .engine import padded_ld
        a = numpy.ascontiguousarray(getattr(value, "tensor", value) if not isinstance(value, torch.Tensor)
                                    else value.detach().cpu().numpy(), dtype=numpy.float32)
        if a.ndim != 2:
            raise ValueError("a host-resident table is a 2-D matrix")
        rows, cols = a.shape
        ld = padded_ld(cols)
        host = torch.zeros((rows, ld), dtype=torch.float32)
        host[:, :cols] = torch.from_numpy(a)
        self.host = host.pin_memory() if torch.cuda.is_available() else host
        self.cols = cols
        self._mirror = None

    # --- the matrix protocol the trainers use
    @property
    def shape(self):
        return (self.host.shape[0], self.cols)

    @property
    def ndim(self):
        return 2

    def __len__(self):
        return self.host.shape[0]

    @property
    def tensor(self):
        if self._mirror is None:
            self._mirror = self.engine.to_device(self.host[:, :self.cols])
        return self._mirror

    def drop_mirror(self):
        self._mirror = None

    def get_value(self, borrow=False, return_internal_type=False):
        if return_internal_type:
            return self.tensor
        return self.host[:, :self.cols].numpy().copy()

    def set_value(self, value, borrow=False):
        self.version += 1
        self._set_host(value)

    def rows(self, indexes, out=None):
        """Device matrix of ``table[indexes]`` (index list / tensor, or a slice): the gather kernel reads the pinned
        rows over PCIe on the current stream."""
        eng = self.engine
        if not hasattr(eng, "gather_host_rows") or not self.host.is_pinned():
            # an engine without a device (the CPU checker of the tests): plain host indexing
            if not isinstance(indexes, slice):
                indexes = torch.as_tensor(numpy.asarray(getattr(indexes, "cpu", lambda: indexes)()), dtype=torch.int64)
            got = eng.to_device(self.host[indexes][:, :self.cols])
            if out is not None:
                out.copy_(got)
                return out
            return got
        if isinstance(indexes, slice):
            lo, hi, step = indexes.indices(len(self))
            indexes = torch.arange(lo, hi, step, dtype=torch.int64, device=eng.device)
        return eng.gather_host_rows(self.host, self.cols, indexes, out=out)

    def __getitem__(self, item):
        return SharedArray(None, engine=self.engine, _tensor=self.rows(item))

    def __repr__(self):
        return "HostTable(name=%r, shape=%r, pinned=%r)" % (self.name, self.shape, self.host.is_pinned())


def host_table_threshold():
    """Bytes above which ``shared(x, resident="auto")`` keeps a table on the host (MDBN_HOST_TABLE_BYTES; default
    128 GiB of the 288 GB of HBM)."""
    import os
    return int(os.environ.get("MDBN_HOST_TABLE_BYTES", 128 << 30))


def shared(value, name=None, borrow=False, engine=None, resident="device", ld=None):
    """``theano.shared(value, name=, borrow=)``.  ``resident``: "device" (default, as the reference), "host" (a
    ``HostTable``: pinned memory, rows gathered over PCIe one minibatch ahead), or "auto" (host above
    ``host_table_threshold()`` bytes)."""
    if isinstance(value, SharedArray):
        return value
    if resident == "auto":
        nbytes = getattr(value, "nbytes", 0)
        resident = "host" if nbytes > host_table_threshold() else "device"
    if resident == "host":
        return HostTable(value, name=name, engine=engine)
    if resident != "device":
        raise ValueError("resident must be 'device', 'host' or 'auto'")
    return SharedArray(value, name=name, engine=engine, ld=ld)


def weight_ld(engine, n_in, n_out):
    """Leading dimension for an [n_in, n_out] weight matrix on ``engine`` (None: the default policy)."""
    fn = getattr(engine, "weight_ld", None)
    return fn(n_in, n_out) if fn is not None else None


def as_tensor(x, engine):
    """numpy / SharedArray / tensor -> device tensor (matrix-padded when 2-D)."""
    t = getattr(x, "tensor", x)
    if isinstance(t, torch.Tensor) and t.dim() == 2:
        return engine.as_matrix(t)
    return engine.to_device(t)
