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
    net = _fit(DBN(numpy_rng=rng, n_ins=n_visible, hidden_layers_sizes=layers_sizes[:-1], n_outs=n_top),
               rows, held_out, batch_size, graph_output, k=k, pretraining_epochs=pretraining_epochs,
               pretrain_lr=pretrain_lr, lambda_1=lambda_1, lambda_2=lambda_2)
    return net, net.get_output(rows), net.get_output(held_out)
