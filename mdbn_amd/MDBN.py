"""Multimodal DBN glue with the surface of the reference's src/MDBN.py:29-76."""
from __future__ import print_function

from .dbn import DBN
from .shared import shared


def train_top(batch_size, graph_output, joint_train_set, joint_val_set, rng):
    """Joint Bernoulli DBN over the concatenated modality outputs (MDBN.py:31-42)."""
    joint_train_set = shared(joint_train_set)
    top_DBN = DBN(numpy_rng=rng, n_ins=joint_train_set.get_value().shape[1],
                  gauss=False,
                  hidden_layers_sizes=[24],
                  n_outs=3)
    top_DBN.training(joint_train_set,
                     batch_size, k=1,
                     pretraining_epochs=[800, 800],
                     pretrain_lr=[0.1, 0.1],
                     validation_set_x=joint_val_set,
                     graph_output=graph_output)
    return top_DBN


def train_bottom_layer(train_set, validation_set,
                       batch_size=20,
                       k=1, layers_sizes=[40],
                       pretraining_epochs=[800],
                       pretrain_lr=[0.005],
                       lambda_1=0.0,
                       lambda_2=0.1,
                       rng=None,
                       graph_output=False):
    """Per-modality DBN with a Gaussian first layer (MDBN.py:45-76); returns
    ``(dbn, output_train_set, output_val_set)``."""
    train_set = shared(train_set)
    if validation_set is not None:
        validation_set = shared(validation_set)
    if DBN.verbose:
        print('Visible nodes: %i' % train_set.get_value().shape[1])
        print('Output nodes: %i' % layers_sizes[-1])
    dbn = DBN(numpy_rng=rng, n_ins=train_set.get_value().shape[1],
              hidden_layers_sizes=layers_sizes[:-1],
              n_outs=layers_sizes[-1])

    dbn.training(train_set,
                 batch_size, k=k,
                 pretraining_epochs=pretraining_epochs,
                 pretrain_lr=pretrain_lr,
                 lambda_1=lambda_1,
                 lambda_2=lambda_2,
                 validation_set_x=validation_set,
                 graph_output=graph_output)

    output_train_set = dbn.get_output(train_set)
    if validation_set is not None:
        output_val_set = dbn.get_output(validation_set)
    else:
        output_val_set = None

    return dbn, output_train_set, output_val_set
