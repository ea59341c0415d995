"""Generates tests/golden/cd_cases.npz from the oracle (oracle/rbm_np.py, float64) with the
Philox twin as the source of uniforms.  The reference holds no fixtures of its own and cannot
run offline (Theano), so these vectors are restatement outputs, not reference outputs
(PARITY UNPINNED, see oracle/__init__.py).  Inputs are regenerated from numpy's frozen legacy
RandomState, so only the expected outputs (sub-sampled for the large case) are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import rbm_np                       # noqa: E402
from oracle.philox_np import PhiloxDraws        # noqa: E402

# (name, V, H, B, k, gauss, hyper-parameters)
CASES = [
    ("rbm_6x4_b3_k1", 6, 4, 3, 1, False, dict(lr=0.1, weightcost=2e-4, momentum=0.6)),
    ("grbm_6x4_b3_k1", 6, 4, 3, 1, True, dict(lr=0.005, lambda_2=0.1, momentum=0.0)),
    ("rbm_64x32_b8_k1", 64, 32, 8, 1, False, dict(lr=0.1, weightcost=2e-4, momentum=0.9)),
    ("grbm_64x32_b8_k1", 64, 32, 8, 1, True, dict(lr=0.005, lambda_1=0.01, lambda_2=0.1, momentum=0.0)),
    ("rbm_64x32_b8_k5", 64, 32, 8, 5, False, dict(lr=0.1, weightcost=2e-4, momentum=0.6)),
    ("grbm_64x32_b8_k5", 64, 32, 8, 5, True, dict(lr=0.005, lambda_1=0.01, lambda_2=0.01, momentum=0.0)),
    ("rbm_784x500_b20_k1", 784, 500, 20, 1, False, dict(lr=0.1, weightcost=2e-4, momentum=0.6)),
    ("grbm_784x500_b20_k1", 784, 500, 20, 1, True, dict(lr=0.005, lambda_2=0.1, momentum=0.0)),
]
SEED, STREAM, N_STEPS = 20161230, 0, 3


def make_inputs(V, H, B, gauss):
    """Data + initial state, all from RandomState(123) in the reference's draw order
    (randint for the stream seed, then the weight matrix: rbm.py:87-107)."""
    rs = np.random.RandomState(123)
    rs.randint(2 ** 30)
    W = rbm_np.init_W(rs, V, H, np.float32)                      # float32 like floatX=float32
    n = 4 * B
    if gauss:
        data = rs.normal(size=(n, V)).astype(np.float32)
    else:
        data = (np.round(255 * rs.beta(0.1, 0.7, size=(n, V))) / 255).astype(np.float32)
    idx = np.stack([rs.permutation(n)[:B] for _ in range(N_STEPS)]).astype(np.int32)
    return W, data, idx


def sub(a):
    """Deterministic sub-sample of big arrays (kept small in the repo)."""
    a = np.asarray(a)
    if a.size <= 4096:
        return a
    if a.ndim == 2:
        return a[::max(1, a.shape[0] // 24), ::max(1, a.shape[1] // 24)]
    return a[::max(1, a.size // 512)]


def run_case(V, H, B, k, gauss, hp):
    W, data, idx = make_inputs(V, H, B, gauss)
    s = rbm_np.RBMState(V, H, W=W, dtype=np.float64, gauss=gauss)
    if hp.get("weightcost", 0.0):
        s.freeze_W0()
    out = {}
    for t in range(N_STEPS):
        v0 = data[idx[t]].astype(np.float64)
        cost, ex = rbm_np.cd_step(s, v0, PhiloxDraws(SEED, STREAM, t), k=k, batch_size=B,
                                  return_extras=True, **hp)
        out["cost_%d" % t] = np.float64(cost)
        if t == 0:
            for key in ("ph_mean", "nv_mean", "nh_mean", "S", "s_h", "s_v"):
                out[key] = sub(ex[key])
            out["ph_sample"] = sub(ex["ph_sample"])
    for key in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed"):
        out[key] = sub(getattr(s, key))
    out["free_energy"] = rbm_np.free_energy(s, data[:B].astype(np.float64))
    return out


def main():
    blob = {}
    for name, V, H, B, k, gauss, hp in CASES:
        for key, val in run_case(V, H, B, k, gauss, hp).items():
            blob["%s/%s" % (name, key)] = val
    path = os.path.join(HERE, "cd_cases.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
