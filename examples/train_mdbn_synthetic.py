#!/usr/bin/env python
"""Three-modality MDBN on synthetic TCGA-shaped tables, following the reference driver
src/AMLsm2.py:16-110 call for call (train_ME / train_GE / train_SM -> train_bottom_layer,
concatenate the top activations, train_top, find classes, save_network / load_network) with
mdbn_amd in place of Theano.  BASELINE.json configs[4] uses the same shapes.

    python examples/train_mdbn_synthetic.py [--rows 1024 --epochs 40 --batch 64]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import mdbn_amd                                                    # noqa: E402
from mdbn_amd import DBN, checkpoint, shared                        # noqa: E402
from mdbn_amd.MDBN import train_bottom_layer                        # noqa: E402
from mdbn_amd.utils import find_unique_classes, preprocess_table    # noqa: E402


def synthetic_tables(n_persons, seed=0):
    """Feature-by-person tables like the reference's TSVs: GE / miRNA expression ~ lognormal,
    somatic mutations sparse 0/1 (preprocess_AML_sm.ipynb)."""
    rs = numpy.random.RandomState(seed)
    groups = rs.randint(0, 3, n_persons)                    # three planted patient groups
    def expr(n_feat, scale):
        centers = rs.normal(0, 1, (3, n_feat))
        return numpy.exp(scale * (centers[groups] + rs.normal(0, 1, (n_persons, n_feat)))).T
    sm = (rs.uniform(size=(n_persons, 256)) < 0.02 + 0.05 * (groups[:, None] == rs.randint(0, 3, 256)[None, :])).T
    return {"GE": expr(2048, 0.5), "ME": expr(512, 0.5), "SM": sm.astype(numpy.float64)}, groups


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    DBN.verbose = False
    rng = numpy.random.RandomState(123)                     # AMLsm2.py:30-31: one rng threaded through
    numpy.random.seed(0)                                    # the reference shuffles with the global state
    tables, groups = synthetic_tables(args.rows)
    t0 = time.time()
    nets, outs = {}, {}
    presets = {                                             # AMLsm2.py:242-340 (epochs shortened; lr for GRBM 0.002)
        "ME": dict(k=5, layers_sizes=[40], lr=[0.002]),
        "GE": dict(k=1, layers_sizes=[400, 40], lr=[0.002, 0.1]),
        "SM": dict(k=1, layers_sizes=[200, 20], lr=[0.002, 0.1]),
    }
    for name in ("ME", "GE", "SM"):                         # same order as AMLsm2.py:38-62
        p = presets[name]
        train, val = preprocess_table(tables[name], holdout=0.1, repeats=1, shuffle=False)
        dbn, out_t, out_v = train_bottom_layer(shared(train), shared(val), batch_size=args.batch, k=p["k"],
                                               layers_sizes=p["layers_sizes"],
                                               pretraining_epochs=[args.epochs * (len(train) // args.batch)] * len(p["layers_sizes"]),
                                               pretrain_lr=p["lr"], lambda_1=0.0, lambda_2=0.1, rng=rng)
        nets[name.lower()], outs[name] = dbn, (out_t, out_v)
        print("%s: DBN %s trained, output %s" % (name, dbn.number_of_nodes(), out_t.shape))
    joint_t = numpy.concatenate([outs[n][0] for n in ("ME", "GE", "SM")], axis=1)      # AMLsm2.py:81-83
    joint_v = numpy.concatenate([outs[n][1] for n in ("ME", "GE", "SM")], axis=1)
    top = DBN(numpy_rng=rng, n_ins=joint_t.shape[1], gauss=False, hidden_layers_sizes=[128], n_outs=3)
    top.training(shared(joint_t), args.batch, k=1, pretraining_epochs=[args.epochs * (len(joint_t) // args.batch)] * 2,
                 pretrain_lr=[0.1, 0.1], validation_set_x=shared(joint_v))
    nets["top"] = top
    classes, D = find_unique_classes((top.get_output(joint_t) > 0.5) * numpy.ones(1))  # AMLsm2.py:103-105
    print("joint layer %s -> %d classes; trained in %.1f s" % (top.number_of_nodes(), len(numpy.unique(classes)), time.time() - t0))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "parameters_and_classes.npz")
        checkpoint.save_network(path, nets, classes=classes, holdout=0.1, repeats=1)
        back = checkpoint.load_network(path)
        same = all(numpy.array_equal(a.get_value(), b.get_value())
                   for n in ("me", "ge", "sm", "top") for a, b in zip(nets[n].params, back[n].params))
        print("checkpoint round trip:", "ok" if same else "MISMATCH")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
