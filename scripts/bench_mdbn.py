#!/usr/bin/env python
"""BASELINE configs[4] on ONE GPU, end to end through the reference's own driver surface (MDBN.train_bottom_layer per
modality, joint DBN on the concatenated outputs): GE 2048 -> 400 -> 40, miRNA 512 -> 40, SM 256 -> 200 -> 20 with CD-5 on
the first layers, batch 512, joint Bernoulli layer 100 -> 128 -> 3.  Synthetic TCGA-shaped rows.  Prints
one JSON line: CD steps / s and samples / s of the whole pipeline (pretraining of all 7 layers, lower-layer forward
passes and epoch bookkeeping included).
    python scripts/bench_mdbn.py [--rows 16384] [--epochs 3]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mdbn_amd
from mdbn_amd import MDBN

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--epochs", type=int, default=400, help="pretraining_epochs of every layer (= its patience in iterations)")
args = ap.parse_args()
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
mdbn_amd.DBN.verbose = False
N, B, E = args.rows, 512, args.epochs
rs = np.random.RandomState(0)
ge = rs.normal(size=(N, 2048)).astype(np.float32)
me = rs.normal(size=(N, 512)).astype(np.float32)
sm = (rs.uniform(size=(N, 256)) < 0.02).astype(np.float32)
sm = ((sm - sm.mean(0)) / (sm.std(0) + 1e-3)).astype(np.float32)


def pipeline():
    rng = np.random.RandomState(123)
    np.random.seed(3)
    outs, nets = [], []
    for data, sizes, lr in ((me, [40], [0.002]), (ge, [400, 40], [0.001, 0.1]), (sm, [200, 20], [0.002, 0.1])):
        net, out_t, _ = MDBN.train_bottom_layer(data, None, batch_size=B, k=5, layers_sizes=sizes,
                                                pretraining_epochs=[E] * len(sizes), pretrain_lr=lr, rng=rng)
        outs.append(out_t)
        nets.append(net)
    joint = np.concatenate(outs, axis=1)
    top = mdbn_amd.DBN(numpy_rng=rng, n_ins=joint.shape[1], gauss=False, hidden_layers_sizes=[128], n_outs=3)
    top.training(mdbn_amd.shared(joint), batch_size=B, k=1, pretraining_epochs=[E, E], pretrain_lr=[0.1, 0.1])
    eng.synchronize()
    return top, sum(r._n_updates for net in nets + [top] for r in net.rbm_layers)


pipeline()                                  # warm-up: allocations, first-use costs
t0 = time.perf_counter()
top, steps = pipeline()
dt = time.perf_counter() - t0                # steps: CD updates actually made (the reference's patience counts ITERATIONS,
                                             # dbn.py:440,506: a layer stops after about pretraining_epochs[i] minibatches)
print(json.dumps({"workload": "configs[4]: three-modality MDBN, CD-5 first layers, batch 512, one GPU, synthetic %d rows, %d epochs "
                  "(patience) per layer" % (N, E), "seconds": dt, "cd_steps": steps, "cd_steps_per_s": steps / dt,
                  "samples_per_s": steps * B / dt,
                  "final_joint_W_abs_mean": float(np.abs(top.rbm_layers[0].W.get_value()).mean())}))
