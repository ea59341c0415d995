"""Deep Belief Network with the class surface of the reference's src/dbn.py:51-559,
trained layer-wise with CD-k on the HIP engine.

Kept from the reference: constructor arguments and numpy draw order (dbn.py:110-114,
155-159), shared W/b between HiddenLayer and RBM (dbn.py:168-202), ``params`` order
[W0, b0, W1, b1, ...] with names 'W'/'b', the greedy layer-wise loop with its momentum
schedule and patience rule (dbn.py:426-508), ``get_output`` / ``number_of_nodes``.
Not kept: matplotlib output (``graph_output`` is accepted and ignored) and Theano MonitorMode
(``monitor`` is accepted and ignored).
"""
from __future__ import print_function

import sys
import timeit

import numpy

from . import mlp
from .engine import get_engine
from .mlp import HiddenLayer
from .rbm import GRBM, RBM, Scalar, function
from .rng import RandomStreams
from .shared import SharedArray, shared
from .utils import get_minibatches_idx


class DBN(object):
    """Deep Belief Network: stacked RBMs sharing weights with an MLP (dbn.py:51-62)."""

    verbose = True          # the reference prints progress; set False to silence
    cache_lower = True      # cache frozen lower-layer activations while layer i trains

    def __init__(self, numpy_rng=None, theano_rng=None, n_ins=784, gauss=True,
                 hidden_layers_sizes=[400], n_outs=40, W_list=None, b_list=None, engine=None):
        self.engine = engine if engine is not None else get_engine()
        self.n_ins = n_ins
        self.sigmoid_layers = []
        self.rbm_layers = []
        self.params = []
        self.stacked_layers_sizes = list(hidden_layers_sizes) + [n_outs]
        self.n_layers = len(self.stacked_layers_sizes)
        self.shuffle_rng = None      # None = numpy's global state, as the reference (utils.py:62)

        assert self.n_layers > 0

        if numpy_rng is None:
            numpy_rng = numpy.random.RandomState(123)
        if theano_rng is None:
            theano_rng = RandomStreams(numpy_rng.randint(2 ** 30))

        self.x = 'x'                 # the data matrix placeholder (dbn.py:119)

        for i in range(self.n_layers):
            input_size = n_ins if i == 0 else self.stacked_layers_sizes[i - 1]
            layer_input = None if i == 0 else self.sigmoid_layers[-1].output
            n_in, n_out = input_size, self.stacked_layers_sizes[i]
            self._print('Adding a layer with %i input and %i outputs' % (n_in, n_out))

            if W_list is None:
                W = numpy.asarray(numpy_rng.uniform(low=-4. * numpy.sqrt(6. / (n_in + n_out)),
                                                    high=4. * numpy.sqrt(6. / (n_in + n_out)),
                                                    size=(n_in, n_out)), dtype=numpy.float32)
            else:
                W = W_list[i]
            b = numpy.zeros((n_out,), dtype=numpy.float32) if b_list is None else b_list[i]

            sigmoid_layer = HiddenLayer(rng=numpy_rng, input=layer_input, n_in=n_in, n_out=n_out,
                                        W=shared(W, name='W', engine=self.engine),
                                        b=shared(b, name='b', engine=self.engine),
                                        activation=mlp.sigmoid, engine=self.engine)
            self.sigmoid_layers.append(sigmoid_layer)
            # only the MLP's W, b are parameters of the DBN; the RBMs' visible biases are
            # not (dbn.py:179-184)
            self.params.extend(sigmoid_layer.params)

            cls = GRBM if (i == 0 and gauss) else RBM
            rbm_layer = cls(numpy_rng=numpy_rng, theano_rng=theano_rng, input=layer_input,
                            n_visible=input_size, n_hidden=n_out,
                            W=sigmoid_layer.W, hbias=sigmoid_layer.b, engine=self.engine)
            self.rbm_layers.append(rbm_layer)
        self._lower_cache = {}

    def _print(self, *a, **kw):
        if self.verbose:
            print(*a, **kw)

    def number_of_nodes(self):
        '''[n_ins] + stacked layer sizes (dbn.py:206-212)'''
        return [self.n_ins] + self.stacked_layers_sizes

    def _forward(self, input, layer):
        """Device activations of ``sigmoid_layers[layer]`` for a data matrix."""
        return self.sigmoid_layers[layer].output.eval(getattr(input, "tensor", input))

    def get_output(self, input, layer=-1):
        '''Output of MLP layer ``layer`` for the samples ``input`` (dbn.py:214-236): host array,
        or None if the input is None.'''
        if input is None:
            return None
        return self.engine.to_numpy(self._forward(input, layer))

    def _layer_input_fn(self, i, train_set_x):
        """Input matrix of RBM i: the data for i == 0, else the activations of layer i-1
        (dbn.py:146).  Lower layers are frozen while layer i trains (dbn.py:426-458), so
        the activations are cached until a lower layer is updated again."""
        if i == 0:
            return None

        def provider():
            version = tuple(r._n_updates for r in self.rbm_layers[:i])
            hit = self._lower_cache.get(i)
            if self.cache_lower and hit is not None and hit[0] == version and hit[1] is train_set_x:
                return hit[2]
            out = self._forward(train_set_x, i - 1)
            self._lower_cache[i] = (version, train_set_x, out)
            return out
        return provider

    def training_functions(self, train_set_x, batch_size, k, lambda_1=0.0, lambda_2=0.1,
                           monitor=False):
        '''Per-layer step functions ``fn(indexes=, momentum=, lr=)`` and free-energy functions
        ``fn(train_sample, test_sample)`` (dbn.py:238-332).'''
        learning_rate = Scalar('lr')
        assert batch_size > 1                                          # dbn.py:276
        train_set_x = shared(train_set_x, engine=self.engine)
        train_fns, free_energy_gap_fns = [], []
        for i, rbm in enumerate(self.rbm_layers):
            if isinstance(rbm, GRBM):
                cost, updates = rbm.get_cost_updates(learning_rate, lambda_1=lambda_1,
                                                     lambda_2=lambda_2, batch_size=batch_size,
                                                     persistent=None, k=k)
            else:
                cost, updates = rbm.get_cost_updates(learning_rate, weightcost=0.0002,
                                                     batch_size=batch_size, persistent=None, k=k)
            train_fns.append(function(updates, train_set_x,
                                      input_fn=self._layer_input_fn(i, train_set_x)))
            free_energy_gap_fns.append(rbm.free_energies)
        return train_fns, free_energy_gap_fns

    def training(self, train_set_x, batch_size, k, pretraining_epochs, pretrain_lr,
                 lambda_1=0.0, lambda_2=0.1, validation_set_x=None, monitor=False,
                 graph_output=False):
        '''Greedy layer-wise pre-training (dbn.py:334-517).  Returns, per layer, the list of
        (iteration, cost, free_energy_gap) records taken at the validation points.'''
        train_set_x = shared(train_set_x, engine=self.engine)
        if validation_set_x is not None:
            validation_set_x = shared(validation_set_x, engine=self.engine)
        self._print('... getting the pretraining functions')
        self._print('Training set sample size %i' % train_set_x.shape[0])
        if validation_set_x is not None:
            self._print('Validation set sample size %i' % validation_set_x.shape[0])

        training_fns, free_energy_gap_fns = self.training_functions(
            train_set_x=train_set_x, batch_size=batch_size, k=k,
            lambda_1=lambda_1, lambda_2=lambda_2, monitor=monitor)

        self._print('... pre-training the model')
        start_time = timeit.default_timer()
        n_data = train_set_x.shape[0]

        patience_increase = 2            # dbn.py:410
        improvement_threshold = 0.995    # dbn.py:412

        idx_minibatches, minibatches = get_minibatches_idx(n_data, batch_size, shuffle=True,
                                                           rng=self.shuffle_rng)
        n_train_batches = idx_minibatches[-1] + 1
        history = []

        for i in range(self.n_layers):
            momentum = 0.0 if isinstance(self.rbm_layers[i], GRBM) else 0.6    # dbn.py:430-433
            best_cost = numpy.inf
            epoch = 0
            done_looping = False
            records = []

            patience = pretraining_epochs[i]      # compared against the ITERATION count (dbn.py:440,506)
            validation_frequency = max(1, min(20 * n_train_batches, patience // 2))
            self._print('Validation frequency: %d' % validation_frequency)

            while (epoch < pretraining_epochs[i]) and (not done_looping):
                epoch = epoch + 1
                idx_minibatches, minibatches = get_minibatches_idx(n_data, batch_size, shuffle=True,
                                                                   rng=self.shuffle_rng)
                dev_idx = self.engine.index_tensor(numpy.concatenate(minibatches))
                if not isinstance(self.rbm_layers[i], GRBM) and epoch == 6:      # dbn.py:452-453
                    momentum = 0.9

                start = 0
                for mb, minibatch in enumerate(minibatches):
                    n_mb = len(minibatch)
                    current_cost = training_fns[i](indexes=dev_idx[start:start + n_mb],
                                                   momentum=momentum, lr=pretrain_lr[i])
                    start += n_mb
                    iter = (epoch - 1) * n_train_batches + mb

                    if (iter + 1) % validation_frequency == 0:
                        current_cost = float(current_cost)
                        self._print('Pre-training cost (layer %i, epoch %d): ' % (i, epoch), end=' ')
                        self._print(current_cost)
                        free_energy_gap = None
                        if current_cost < best_cost:
                            if current_cost < best_cost * improvement_threshold:
                                patience = max(patience, iter * patience_increase)
                            best_cost = current_cost
                            if validation_set_x is not None:
                                n_val = validation_set_x.shape[0]
                                if i == 0:
                                    input_t_set, input_v_set = train_set_x, validation_set_x
                                else:                                   # dbn.py:494-496
                                    input_t_set = self._forward(train_set_x.tensor[:n_val], i - 1)
                                    input_v_set = self._forward(validation_set_x, i - 1)
                                free_energy_train, free_energy_test = free_energy_gap_fns[i](
                                    input_t_set, input_v_set)
                                free_energy_gap = float(free_energy_test.mean() - free_energy_train.mean())
                                self._print('Free energy gap (layer %i, epoch %i): ' % (i, epoch), end=' ')
                                self._print(free_energy_gap)
                        records.append((iter, current_cost, free_energy_gap))

                    if patience <= iter:                                # dbn.py:506-508
                        done_looping = True
                        break
            training_fns[i].flush()
            history.append(records)

        end_time = timeit.default_timer()
        if self.verbose:
            print('The pretraining ran for %.2fm' % ((end_time - start_time) / 60.), file=sys.stderr)
        return history

    def MLP_output_from_datafile(self, datafile, holdout=0.0, repeats=1, clip=None,
                                 transform_fn=None, exponent=1.0, datadir='data'):
        """Reload a table (unshuffled) and return the network's outputs for its train and
        validation parts (dbn.py:519-536)."""
        from .utils import load_n_preprocess_data
        train_set, validation_set = load_n_preprocess_data(datafile, holdout=holdout, clip=clip,
                                                           transform_fn=transform_fn, exponent=exponent,
                                                           repeats=repeats, shuffle=False, datadir=datadir)
        return (self.get_output(train_set), self.get_output(validation_set))
