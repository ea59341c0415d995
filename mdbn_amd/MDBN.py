"""Multimodal glue: one DBN per modality, one joint DBN on top (the call surface of the reference's
src/MDBN.py:29-76, which its experiment drivers use: AMLsm2.py:38-92)."""
from __future__ import print_function

from .dbn import DBN
from .shared import shared

# the joint network of MDBN.py:31-42: Bernoulli 24 -> 3 on the concatenated modality outputs, CD-1
TOP_SHAPE = dict(gauss=False, hidden_layers_sizes=[24], n_outs=3)
TOP_SCHEDULE = dict(k=1, pretraining_epochs=[800, 800], pretrain_lr=[0.1, 0.1])


def _fit(net, data, held_out, batch_size, graph_output, **schedule):
    net.training(data, batch_size, validation_set_x=held_out, graph_output=graph_output, **schedule)
    return net


def train_top(batch_size, graph_output, joint_train_set, joint_val_set, rng):
    """Joint DBN over the concatenated top-layer activations of the modalities (MDBN.py:31-42)."""
    joint = shared(joint_train_set)
    net = DBN(numpy_rng=rng, n_ins=joint.shape[1], **TOP_SHAPE)
    return _fit(net, joint, joint_val_set, batch_size, graph_output, **TOP_SCHEDULE)


def build_bottom_layer(n_visible, layers_sizes, rng):
    """The (untrained) DBN of one modality: Gaussian first layer, ``layers_sizes[-1]`` output nodes (MDBN.py:58-62).
    Constructing it is the only thing that draws from ``rng`` (dbn.py:110-114,155-159)."""
    return DBN(numpy_rng=rng, n_ins=n_visible, hidden_layers_sizes=layers_sizes[:-1], n_outs=layers_sizes[-1])


def train_modalities(modalities, rng, group="auto", shuffle_seed=None, graph_output=False):
    """One DBN per modality, as successive ``train_bottom_layer`` calls with one ``rng`` threaded through them
    (AMLsm2.py:38-62) -- placed MODALITY-PARALLEL when a process group of N > 1 ranks is up (the c5
    alternative to row-sharding tiny layers): the modalities are independent until the joint layer, so modality i
    is trained on rank i % N alone, with single-process step functions and no collective per step, and its
    parameters are broadcast once when it is done.

    ``modalities``: ordered list of dicts with ``train_set``, optional ``validation_set``, and the keyword
    arguments of ``train_bottom_layer`` (batch_size, k, layers_sizes, pretraining_epochs, pretrain_lr, lambda_1,
    lambda_2).  Returns ``[(net, out_train, out_val), ...]`` in the same order, identical on every rank.

    Every rank CONSTRUCTS all networks in order -- construction is all that consumes ``rng``, so the initial weights
    and the state ``rng`` is left in (for ``train_top``) are those of the sequential reference on every rank.  The
    minibatch shuffles cannot come from numpy's global state as in the reference (utils.py:62: its sequence would
    depend on how long the previous modality trained): modality i shuffles with ``RandomState(shuffle_seed + i)``
    (``shuffle_seed`` defaults to 0 under a process group; None in a single process keeps the reference's global
    state).  With the same ``shuffle_seed`` the placement does not change any result."""
    import numpy
    from . import dist
    grp = dist.default_group() if group == "auto" else group
    world, rank = (grp.world_size, grp.rank) if grp is not None else (1, 0)
    if world > 1 and shuffle_seed is None:
        shuffle_seed = 0
    nets = []
    for i, m in enumerate(modalities):
        rows = shared(m["train_set"])
        net = build_bottom_layer(rows.shape[1], m.get("layers_sizes", [40]), rng)
        if shuffle_seed is not None:
            net.shuffle_rng = numpy.random.RandomState(shuffle_seed + i)
        if world > 1:
            net.data_parallel = None                 # trained by its owner alone
        nets.append((net, rows, None if m.get("validation_set") is None else shared(m["validation_set"])))
    for i, (m, (net, rows, held_out)) in enumerate(zip(modalities, nets)):
        if i % world == rank:
            _fit(net, rows, held_out, m.get("batch_size", 20), graph_output, k=m.get("k", 1),
                 pretraining_epochs=m.get("pretraining_epochs", [800]), pretrain_lr=m.get("pretrain_lr", [0.005]),
                 lambda_1=m.get("lambda_1", 0.0), lambda_2=m.get("lambda_2", 0.1))
    if world > 1:
        import torch.distributed as td
        for i, (net, _, _) in enumerate(nets):       # one broadcast per array, once per modality
            for r in net.rbm_layers:
                for arr in (r.W, r.hbias, r.vbias, r.W_speed, r.hbias_speed, r.vbias_speed):
                    t = arr.tensor
                    base = t._base if getattr(t, "_base", None) is not None else t     # the padded storage of a matrix
                    td.broadcast(base, src=i % world, group=grp.pg)
                    arr.version += 1                 # caches keyed on the array (lower-layer activations) are stale
            # the host-side counters of the owner's layers (Philox step, update count, pseudo-likelihood bit): replicas stay
            # interchangeable for whatever follows (sampling, further training, checkpoints)
            state = [[(r._rng_step, r._n_updates, r.bit_i_idx) for r in net.rbm_layers]]
            td.broadcast_object_list(state, src=i % world, group=grp.pg)
            for r, (step, n_upd, bit) in zip(net.rbm_layers, state[0]):
                r._rng_step, r._n_updates, r.bit_i_idx = step, max(n_upd, r._n_updates + 1), bit
    return [(net, net.get_output(rows), net.get_output(held_out)) for net, rows, held_out in nets]


def train_bottom_layer(train_set, validation_set, batch_size=20, k=1, layers_sizes=[40],
                       pretraining_epochs=[800], pretrain_lr=[0.005], lambda_1=0.0, lambda_2=0.1,
                       rng=None, graph_output=False):
    """DBN of one modality, Gaussian first layer (MDBN.py:45-76).  Returns the network and its
    outputs for the training and the validation rows (None without validation rows)."""
    rows = shared(train_set)
    held_out = None if validation_set is None else shared(validation_set)
    n_visible, n_top = rows.shape[1], layers_sizes[-1]
    if DBN.verbose:
        print('Visible nodes: %i' % n_visible)
        print('Output nodes: %i' % n_top)
    net = _fit(build_bottom_layer(n_visible, layers_sizes, rng),
               rows, held_out, batch_size, graph_output, k=k, pretraining_epochs=pretraining_epochs,
               pretrain_lr=pretrain_lr, lambda_1=lambda_1, lambda_2=lambda_2)
    return net, net.get_output(rows), net.get_output(held_out)
