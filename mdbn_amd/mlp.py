"""HiddenLayer with the surface of the reference's src/mlp.py:36-110.

``output`` is a ``LayerOutput``: the deferred ``activation(input . W + b)`` that upper
layers / RBMs take as their input (dbn.py:146,190,198); ``output.eval(x)`` evaluates the
chain for a data matrix ``x`` on the device (the forward pass K11 of SURVEY 2.2)."""
import numpy
import torch

from .engine import get_engine
from .shared import as_tensor, shared, weight_ld


class LayerOutput(object):
    def __init__(self, layer):
        self.layer = layer

    def eval(self, x):
        inp = self.layer.input
        if isinstance(inp, LayerOutput):
            x = inp.eval(x)
        return self.layer.forward(x)


def sigmoid(x):
    """Marker for ``theano.tensor.nnet.sigmoid`` as the activation argument."""
    return torch.sigmoid(x)


def tanh(x):
    return torch.tanh(x)


class HiddenLayer(object):
    def __init__(self, rng, input, n_in, n_out, W=None, b=None, activation=tanh, engine=None):
        """Fully-connected layer ``activation(dot(input, W) + b)``; W is (n_in, n_out), b is
        (n_out,).  Initialisation as mlp.py:80-96 (uniform +-sqrt(6/(n_in+n_out)), x4 for
        sigmoid)."""
        self.engine = engine if engine is not None else get_engine()
        self.input = input
        if W is None:
            bound = numpy.sqrt(6. / (n_in + n_out))
            W_values = numpy.asarray(rng.uniform(low=-bound, high=bound, size=(n_in, n_out)),
                                     dtype=numpy.float32)
            if activation is sigmoid:
                W_values *= 4
            W = shared(W_values, name='W', engine=self.engine, ld=weight_ld(self.engine, n_in, n_out))
        if b is None:
            b = shared(numpy.zeros((n_out,), dtype=numpy.float32), name='b', engine=self.engine)
        self.W = shared(W, name='W', engine=self.engine)
        self.b = shared(b, name='b', engine=self.engine)
        self.activation = activation
        self.output = LayerOutput(self)
        self.params = [self.W, self.b]

    def forward(self, x):
        """Device forward pass for a data matrix (mlp.py:103-107)."""
        x = as_tensor(x, self.engine)
        if self.activation is sigmoid:
            _, mean, _ = self.engine.propup(x, self.W.tensor, self.b.tensor,
                                            want_pre=False, want_sample=False)
            return mean
        pre, _, _ = self.engine.propup(x, self.W.tensor, self.b.tensor,
                                       want_mean=False, want_sample=False)
        if self.activation is None:
            return pre
        if self.activation is tanh:
            return self.engine.tanh_(pre)
        return self.engine.as_matrix(self.activation(pre))      # a caller-supplied torch callable
