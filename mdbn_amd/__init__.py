"""mdbn_amd -- MI355X-native CD-k engine behind the RBM / GRBM / DBN class surface of
glgerard/MDBN (src/rbm.py, src/dbn.py, src/mlp.py, src/MDBN.py).

Host code is Python; all arithmetic runs in hand-written HIP kernels (gfx950) reached
through the C-ABI of ``libmdbn_hip.so`` (include/mdbn_hip.h).  There is no CPU fallback:
using the classes without the built library or without a GPU raises ``MdbnError``.
"""
from ._lib import MdbnError
from .engine import HipEngine, RngAddr, get_engine, set_engine
from .rng import RandomStreams
from .shared import HostTable, SharedArray, shared
from .utils import get_minibatches_idx
from .mlp import HiddenLayer
from .rbm import RBM, GRBM, Scalar, function
from .dbn import DBN
from . import MDBN, checkpoint, dist, utils

__all__ = ["MdbnError", "HipEngine", "RngAddr", "get_engine", "set_engine", "RandomStreams",
           "SharedArray", "HostTable", "shared", "get_minibatches_idx", "HiddenLayer", "RBM", "GRBM",
           "Scalar", "function", "DBN", "MDBN", "dist", "checkpoint", "utils"]
