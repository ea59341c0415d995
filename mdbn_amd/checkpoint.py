"""npz checkpoints with the schema of the reference's ``save_network`` / ``load_network``
(src/AMLsm2.py:112-205): per network ``<name>_config`` (a dict with 'number_of_nodes', ...),
``<name>_params`` (list of ``{p.name: p.get_value()}`` in DBN.params order: W0, b0, W1, b1, ...),
plus ``classes``, ``holdout``, ``repeats`` -- so files written here load in the reference's
notebooks and vice versa.

The reference saves only W and b at the end of a run (vbias, momentum speeds, RNG position and
epoch are lost, so it cannot resume).  ``save_network(..., resume=True)`` adds
``<name>_resume``: visible biases, the three speeds, RNG seed / stream / step, the update count and
-- where a step function with a weight cost was built -- the frozen weight-cost constant W0
(rbm.py:415) of every RBM layer; ``load_network`` restores them when present, and the next
``get_cost_updates`` of a restored layer reuses that W0 instead of snapshotting the resumed W.

Trainer-loop resume (round 4): while ``DBN.training(..., on_step=)`` runs, ``dbn.trainer_state`` holds where its loop
stands -- layer index, epoch, next minibatch, the early-stopping state (patience limit, best cost), the state of the
shuffle generator when the epoch's order was drawn, the records so far and the finished layers' histories.
``save_network(..., resume=True)`` stores it as ``<name>_trainer``; ``load_network`` hands it back as
``dbn.trainer_state`` and ``dbn.training(<same arguments>, resume=dbn.trainer_state)`` continues the interrupted run with
the same minibatch order and validation points: parameters and histories equal the uninterrupted run's bit for bit
(tests/test_plumbing.py, tests/test_gpu_parity.py)."""
import numpy

from .dbn import DBN
from .rbm import GRBM


def _resume_state(dbn):
    out = []
    for r in dbn.rbm_layers:
        w0 = r._W0_snapshot if r._W0_snapshot is not None else None
        out.append({'W0': None if w0 is None else w0.get_value(),
                    'vbias': r.vbias.get_value(), 'W_speed': r.W_speed.get_value(),
                    'hbias_speed': r.hbias_speed.get_value(), 'vbias_speed': r.vbias_speed.get_value(),
                    'rng_seed': r.theano_rng.seed, 'rng_stream': r.stream_id, 'rng_step': r._rng_step,
                    'n_updates': r._n_updates, 'bit_i_idx': r.bit_i_idx})
    return out


def save_network(output_file, networks, classes=None, holdout=0.0, repeats=1, configs=None, resume=False):
    """``networks``: ``{'me': dbn, 'ge': dbn, ..., 'top': dbn}`` (entries may be None).
    ``configs``: optional ``{name: dict}`` merged into ``<name>_config`` next to
    'number_of_nodes' (the reference stores epochs / learning_rate / batch_size / k there)."""
    blob = {'holdout': holdout, 'repeats': repeats}
    if classes is not None:
        blob['classes'] = classes
    for name, dbn in networks.items():
        if dbn is None:
            continue
        cfg = {'number_of_nodes': dbn.number_of_nodes(),
               'gauss': isinstance(dbn.rbm_layers[0], GRBM)}
        cfg.update((configs or {}).get(name, {}))
        blob[name + '_config'] = cfg
        blob[name + '_params'] = numpy.array([{p.name: p.get_value()} for p in dbn.params], dtype=object)
        if resume:
            blob[name + '_resume'] = numpy.array(_resume_state(dbn), dtype=object)
            if getattr(dbn, 'trainer_state', None) is not None:
                holder = numpy.empty(1, dtype=object)          # (one pickled dict: lists of tuples, a RandomState tuple)
                holder[0] = dbn.trainer_state
                blob[name + '_trainer'] = holder
    numpy.savez(output_file, **blob)


def load_network(input_file, names=None, engine=None):
    """Rebuild the DBNs of a checkpoint (reference AMLsm2.py:165-205).  Returns
    ``{name: DBN}`` plus the scalar entries under their own keys.  'gauss' comes from the
    config when present; otherwise the reference's rule applies (only 'top' is Bernoulli)."""
    npz = numpy.load(input_file, allow_pickle=True)
    if names is None:
        names = [k[:-len('_params')] for k in npz.files if k.endswith('_params')]
    out = {}
    for name in names:
        config = npz[name + '_config'].tolist()
        params = npz[name + '_params']
        layer_sizes = config['number_of_nodes']
        n_layers = len(layer_sizes) - 1
        W_list = [params[2 * i]['W'] for i in range(n_layers)]
        b_list = [params[2 * i + 1]['b'] for i in range(n_layers)]
        gauss = config.get('gauss', name != 'top')
        dbn = DBN(n_ins=layer_sizes[0], hidden_layers_sizes=list(layer_sizes[1:-1]), n_outs=layer_sizes[-1],
                  gauss=gauss, W_list=W_list, b_list=b_list, engine=engine)
        if name + '_resume' in npz.files:
            for r, st in zip(dbn.rbm_layers, npz[name + '_resume']):
                r.vbias.set_value(st['vbias'])
                r.W_speed.set_value(st['W_speed'])
                r.hbias_speed.set_value(st['hbias_speed'])
                r.vbias_speed.set_value(st['vbias_speed'])
                r.theano_rng.seed, r.stream_id = int(st['rng_seed']), int(st['rng_stream'])
                r._rng_step, r._n_updates, r.bit_i_idx = int(st['rng_step']), int(st['n_updates']), int(st['bit_i_idx'])
                if st.get('W0') is not None:
                    r._resume_W0 = st['W0']
        if name + '_trainer' in npz.files:
            dbn.trainer_state = npz[name + '_trainer'][0]
        out[name] = dbn
    for key in ('classes', 'holdout', 'repeats'):
        if key in npz.files:
            out[key] = npz[key]
    return out
