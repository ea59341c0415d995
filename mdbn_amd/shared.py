"""SharedArray: the stand-in for ``theano.shared`` variables (reference call sites:
utils.py:113-115, dbn.py:172-173,384-406, rbm.py:109-131,153-162, AMLsm2.py:81-103).

Wraps a device tensor (a padded matrix or a vector owned by an engine) and offers the
observable protocol the reference's callers use: ``get_value(borrow=)``,
``set_value``, ``.name``, plus numpy conversion."""
import numpy
import torch

from .engine import get_engine


class SharedArray(object):
    def __init__(self, value, name=None, engine=None, _tensor=None):
        self.engine = engine if engine is not None else get_engine()
        self.name = name
        self._sync_hook = None      # called before host reads (completes deferred device work)
        self.version = 0            # bumped by every set_value (caches of derived data key on it)
        if _tensor is not None:
            self.tensor = _tensor
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
            self.tensor = self.engine.to_device(new)
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


def shared(value, name=None, borrow=False, engine=None):
    """``theano.shared(value, name=, borrow=)``."""
    if isinstance(value, SharedArray):
        return value
    return SharedArray(value, name=name, engine=engine)


def as_tensor(x, engine):
    """numpy / SharedArray / tensor -> device tensor (matrix-padded when 2-D)."""
    t = getattr(x, "tensor", x)
    if isinstance(t, torch.Tensor) and t.dim() == 2:
        return engine.as_matrix(t)
    return engine.to_device(t)
