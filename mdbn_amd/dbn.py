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
from .shared import HostTable, SharedArray, shared, weight_ld
from .utils import get_minibatches_idx


class _Patience(object):
    """Early stopping of dbn.py:408-441,476-508.  ``limit`` starts at the layer's EPOCH budget but is
    compared with the ITERATION count (the reference does exactly that); every ``every`` iterations the
    cost is looked at: a new best that beats the old one by more than 0.5 % stretches the limit to
    twice the current iteration."""

    growth, threshold = 2, 0.995                     # dbn.py:410,412

    def __init__(self, budget, n_batches):
        self.limit = budget
        self.every = max(1, min(20 * n_batches, budget // 2))       # dbn.py:441
        self.best = numpy.inf

    def due(self, it):
        return (it + 1) % self.every == 0

    def observe(self, it, cost):
        """Record a validation-point cost; True if it is a new best."""
        if not cost < self.best:
            return False
        if cost < self.best * self.threshold:
            self.limit = max(self.limit, it * self.growth)
        self.best = cost
        return True

    def exhausted(self, it):
        return self.limit <= it                      # dbn.py:506


class DBN(object):
    """Deep Belief Network: stacked RBMs sharing weights with an MLP (dbn.py:51-62)."""

    verbose = True          # the reference prints progress; set False to silence
    cache_lower = True      # cache frozen lower-layer activations while layer i trains
    data_parallel = "auto"  # step functions shard minibatches over the ranks of an initialised process group ("auto"),
                            # or None: this network trains in this process alone (modality-parallel placement, MDBN.py)

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
                                        W=shared(W, name='W', engine=self.engine, ld=weight_ld(self.engine, n_in, n_out)),
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
        self.trainer_state = None    # where DBN.training stands (set while it runs with on_step; see training())

    def _print(self, *a, **kw):
        if self.verbose:
            print(*a, **kw)

    def number_of_nodes(self):
        '''[n_ins] + stacked layer sizes (dbn.py:206-212)'''
        return [self.n_ins] + self.stacked_layers_sizes

    host_chunk_rows = 16384     # rows of a host-resident table forwarded per chunk (dbn.py:146's activations)

    def _forward(self, input, layer):
        """Device activations of ``sigmoid_layers[layer]`` for a data matrix.  A host-resident table (HostTable) is
        streamed through in row chunks -- the pinned rows are gathered over PCIe, forwarded, and only the (narrower)
        activations stay on the device -- so the table itself is never uploaded whole."""
        if isinstance(input, HostTable) and input._mirror is None and len(input) > self.host_chunk_rows:
            import torch
            out = None
            for lo in range(0, len(input), self.host_chunk_rows):
                hi = min(len(input), lo + self.host_chunk_rows)
                act = self.sigmoid_layers[layer].output.eval(input.rows(slice(lo, hi)))
                if out is None:
                    out = self.engine.alloc_matrix(len(input), act.shape[1], act.stride(0))
                out[lo:hi].copy_(act)
            return out
        if isinstance(input, HostTable) and input._mirror is None:
            return self.sigmoid_layers[layer].output.eval(input.rows(slice(0, len(input))))
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
            # what the cached activations depend on: the lower layers' training steps, direct writes to
            # their parameters (SharedArray.set_value, e.g. weights loaded into an existing DBN) and
            # in-place writes to the data tensor
            version = tuple((r._n_updates, r.W.version, r.hbias.version) for r in self.rbm_layers[:i]) + \
                (train_set_x.version, 0 if isinstance(train_set_x, HostTable) else getattr(train_set_x.tensor, "_version", 0))
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
            train_fns.append(function(updates, train_set_x, input_fn=self._layer_input_fn(i, train_set_x),
                                      data_parallel=self.data_parallel))
            free_energy_gap_fns.append(rbm.free_energies)
        return train_fns, free_energy_gap_fns

    def training(self, train_set_x, batch_size, k, pretraining_epochs, pretrain_lr,
                 lambda_1=0.0, lambda_2=0.1, validation_set_x=None, monitor=False,
                 graph_output=False, resume=None, on_step=None):
        '''Greedy layer-wise pre-training (dbn.py:334-517).  Returns, per layer, the list of
        (iteration, cost, free_energy_gap) records taken at the validation points.

        Added (the reference cannot resume, AMLsm2.py:112-205 saves W and b only): after every step ``self.trainer_state``
        describes where the loop stands -- layer, epoch, next minibatch, early-stopping state, the shuffle generator's state
        at the start of the epoch, the records so far -- and ``on_step(self)`` is called (e.g. to write a checkpoint:
        ``checkpoint.save_network(..., resume=True)`` stores the trainer state with the layers' own).  ``resume=state``
        continues such a run: same minibatch order, same validation points, same parameters bit for bit.'''
        data = shared(train_set_x, engine=self.engine)
        held_out = None if validation_set_x is None else shared(validation_set_x, engine=self.engine)
        self._print('... getting the pretraining functions')
        self._print('Training set sample size %i' % data.shape[0])
        if held_out is not None:
            self._print('Validation set sample size %i' % held_out.shape[0])
        step_fns, energy_fns = self.training_functions(train_set_x=data, batch_size=batch_size, k=k,
                                                       lambda_1=lambda_1, lambda_2=lambda_2, monitor=monitor)
        self._print('... pre-training the model')
        t_start = timeit.default_timer()
        if resume is None:
            # the batch count is fixed once, from a first split that is otherwise unused (dbn.py:404-406)
            n_batches = len(get_minibatches_idx(data.shape[0], batch_size, shuffle=True, rng=self.shuffle_rng)[1])
            history, first = [], 0
        else:
            n_batches, first = int(resume['n_batches']), int(resume['layer'])
            history = [[tuple(r) for r in h] for h in resume['history']]
            if resume.get('epoch') is None:                  # interrupted between two layers
                self._set_shuffle_state(resume['shuffle_state'])
        for i in range(first, self.n_layers):
            at = resume if (resume is not None and i == first and resume.get('epoch') is not None) else None
            history.append(self._pretrain_layer(i, step_fns[i], energy_fns[i], data, held_out, batch_size, n_batches,
                                                pretraining_epochs[i], pretrain_lr[i], history, at, on_step))
        self.trainer_state = None
        if self.verbose:
            print('The pretraining ran for %.2fm' % ((timeit.default_timer() - t_start) / 60.), file=sys.stderr)
        return history

    def _shuffle_state(self):
        return self.shuffle_rng.get_state() if self.shuffle_rng is not None else numpy.random.get_state()

    def _set_shuffle_state(self, state):
        if self.shuffle_rng is not None:
            self.shuffle_rng.set_state(state)
        else:
            numpy.random.set_state(state)

    def _pretrain_layer(self, i, step_fn, energy_fn, data, held_out, batch_size, n_batches, epoch_budget, lr,
                        history=(), at=None, on_step=None):
        """One layer of dbn.py:426-508: epochs of reshuffled minibatches under the layer's momentum
        schedule until the epoch budget or the early-stopping state (``_Patience``) ends it.  ``at``: the trainer
        state of an interrupted run of THIS layer (epoch, next minibatch, patience, records, shuffle state)."""
        bernoulli = not isinstance(self.rbm_layers[i], GRBM)
        stop = _Patience(epoch_budget, n_batches)
        self._print('Validation frequency: %d' % stop.every)
        hinted = getattr(step_fn, "accepts_next_indexes", False)     # step functions are duck-typed: fn(indexes=, momentum=, lr=)
        records = []
        first_epoch, first_mb = 1, 0
        if at is not None:
            first_epoch, first_mb = int(at['epoch']), int(at['next_mb'])
            stop.limit, stop.best = at['patience']
            records = [tuple(r) for r in at['records']]
            self._set_shuffle_state(at['shuffle_state'])              # ... as it was when this epoch's order was drawn

        def state(epoch, next_mb, shuffle_state, layer=i, recs=None):
            self.trainer_state = {'layer': layer, 'epoch': epoch, 'next_mb': next_mb, 'n_batches': n_batches,
                                  'patience': (stop.limit, stop.best), 'shuffle_state': shuffle_state,
                                  'records': list(records if recs is None else recs), 'history': [list(h) for h in history]}

        for epoch in range(first_epoch, epoch_budget + 1):
            # Gaussian layer: no momentum at all; Bernoulli layers 0.6, 0.9 from the sixth epoch (dbn.py:430-433,452-453)
            momentum = 0.0 if not bernoulli else (0.6 if epoch < 6 else 0.9)
            drawn_from = self._shuffle_state()
            batches = get_minibatches_idx(data.shape[0], batch_size, shuffle=True, rng=self.shuffle_rng)[1]
            order = self.engine.index_tensor(numpy.concatenate(batches))
            bounds = numpy.cumsum([0] + [len(b) for b in batches])
            views = [order[bounds[mb]:bounds[mb + 1]] for mb in range(len(batches))]
            start = first_mb if epoch == first_epoch else 0
            if hasattr(step_fn, "announce"):            # the epoch's order (a host-resident table starts feeding its rows)
                step_fn.announce(views[start:], host_indexes=batches[start:])
            for mb in range(start, len(batches)):
                hint = {}
                if hinted and mb + 1 < len(batches):    # the next minibatch of the epoch (a pure hint: StepFunction.__call__)
                    hint["next_indexes"] = views[mb + 1]
                cost = step_fn(indexes=views[mb], momentum=momentum, lr=lr, **hint)
                it = (epoch - 1) * n_batches + mb
                if stop.due(it):
                    cost = float(cost)
                    self._print('Pre-training cost (layer %i, epoch %d): ' % (i, epoch), end=' ')
                    self._print(cost)
                    gap = None
                    if stop.observe(it, cost) and held_out is not None:
                        gap = self._free_energy_gap(i, energy_fn, data, held_out)
                        self._print('Free energy gap (layer %i, epoch %i): ' % (i, epoch), end=' ')
                        self._print(gap)
                    records.append((it, cost, gap))
                done = stop.exhausted(it)
                if on_step is not None or done:
                    if done or (epoch == epoch_budget and mb + 1 == len(batches)):
                        # the layer is finished: the state points at the start of the next one
                        self.trainer_state = {'layer': i + 1, 'epoch': None, 'next_mb': 0, 'n_batches': n_batches,
                                              'patience': None, 'shuffle_state': self._shuffle_state(), 'records': [],
                                              'history': [list(h) for h in history] + [list(records)]}
                    else:
                        state(epoch, mb + 1, drawn_from)
                    if on_step is not None:
                        step_fn.flush()                 # (an overlapped data-parallel step: its deferred half belongs to this state)
                        on_step(self)
                if done:
                    break
            else:
                continue
            break
        step_fn.flush()
        return records

    def _free_energy_gap(self, i, energy_fn, data, held_out):
        """mean F(validation) - mean F(training rows) (dbn.py:476-501).  The reference's two cases differ in the
        training rows they use: at layer 0 the WHOLE training set (``input_t_set = t_set``, dbn.py:478), at the
        layers above only its first ``n_val`` rows, seen through the layers below
        (``get_output(t_set[range(v_set.shape[0])], i-1)``, dbn.py:481-483)."""
        if i == 0:
            if isinstance(data, HostTable) and data._mirror is None and len(data) > self.host_chunk_rows:
                # a host-resident table is streamed through in row chunks, as in _forward; the mean is taken over
                # the one vector of all rows' energies, so chunking does not change it
                f_val = energy_fn(held_out, held_out)[1]
                rbm = self.rbm_layers[0]
                f_train = numpy.concatenate([
                    rbm.free_energy(data.rows(slice(lo, min(len(data), lo + self.host_chunk_rows)))).get_value()
                    for lo in range(0, len(data), self.host_chunk_rows)])
                return float(f_val.mean() - f_train.mean())
            below_train = data.rows(slice(0, len(data))) if isinstance(data, HostTable) and data._mirror is None \
                else data.tensor
            below_val = held_out
        else:
            n_val = held_out.shape[0]
            first = data.rows(slice(0, n_val)) if isinstance(data, HostTable) else data.tensor[:n_val]
            below_train = self._forward(first, i - 1)
            below_val = self._forward(held_out, i - 1)
        f_train, f_val = energy_fn(below_train, below_val)
        return float(f_val.mean() - f_train.mean())

    def MLP_output_from_datafile(self, datafile, holdout=0.0, repeats=1, clip=None,
                                 transform_fn=None, exponent=1.0, datadir='data'):
        """Reload a table (unshuffled) and return the network's outputs for its train and
        validation parts (dbn.py:519-536)."""
        from .utils import load_n_preprocess_data
        train_set, validation_set = load_n_preprocess_data(datafile, holdout=holdout, clip=clip,
                                                           transform_fn=transform_fn, exponent=exponent,
                                                           repeats=repeats, shuffle=False, datadir=datadir)
        return (self.get_output(train_set), self.get_output(validation_set))
