"""Does the bench workload (BASELINE configs[1]: GRBM 4096 -> 1024, CD-1, batch 512, lambda_2 0.1) diverge at the
reference's lr = 0.005 (MDBN.py:49) because of the REFERENCE'S ARITHMETIC or because of the HIP engine?  (VERDICT r4 #3a)

TEST INFRASTRUCTURE (not collected by pytest; lives under tests/ because it drives the oracle).  The SAME host code
(mdbn_amd.GRBM + mdbn_amd.function, the update rule of rbm.py:347-365) runs free on either engine:

    python tests/probe_lr_stability.py --engine oracle   # float64 numpy restatement (oracle/rbm_np.py), CPU
    python tests/probe_lr_stability.py --engine hip      # libmdbn_hip.so on cuda:0

on the same synthetic rows (torch.randn, CPU generator seed 0), the same minibatch order, the same W init
(RandomState(123)) and the same Philox draws, and prints one line per checkpoint: cost, max|W|, max|W_speed|, the
Frobenius norm of W.  `--compare a.json b.json` prints the two trajectories side by side with their ratio.
Logs of both: profiles/r05a_lr_stability_{oracle,hip}.json, the comparison profiles/r05a_lr_stability_compare.log."""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

V, H, B, N = 4096, 1024, 512, 32768


def run(engine_name, lrs, steps, every, out):
    import numpy as np
    import torch
    import mdbn_amd
    if engine_name == "oracle":
        from _oracle_engine import OracleEngine
        eng = mdbn_amd.set_engine(OracleEngine())
    else:
        eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
    g = torch.Generator(device="cpu").manual_seed(0)
    rows = torch.randn((N, V), generator=g)
    if engine_name == "oracle":
        rows = rows.to(torch.float64)          # the checker engine takes its dtype as is (no per-step conversion)
        data = mdbn_amd.shared(rows, engine=eng)
    else:
        data = mdbn_amd.shared(rows.to(eng.device))
    perm = np.random.RandomState(1).permutation(N)
    result = {"engine": engine_name, "shape": [V, H, B], "steps": steps, "runs": []}
    for lr in lrs:
        rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), engine=eng)
        _, up = rbm.get_cost_updates(lr=lr, k=1, lambda_1=0.0, lambda_2=0.1, batch_size=B)
        fn = mdbn_amd.function(up, data)
        traj = []
        for it in range(steps):
            mb = it % (N // B)
            idx = perm[mb * B:(mb + 1) * B]
            c = float(fn(indexes=torch.from_numpy(idx).to(eng.device) if engine_name != "oracle" else idx, momentum=0.0))
            if it % every == 0 or it == steps - 1:
                W = rbm.W.get_value().astype(np.float64)
                Ws = rbm.W_speed.get_value().astype(np.float64)
                rec = {"step": it, "cost": c, "W_absmax": float(np.abs(W).max()), "Ws_absmax": float(np.abs(Ws).max()),
                       "W_fro": float(np.sqrt((W * W).sum()))}
                traj.append(rec)
                print("%s lr %g step %4d cost %.6g |W|max %.6g |Ws|max %.6g |W|F %.6g"
                      % (engine_name, lr, it, c, rec["W_absmax"], rec["Ws_absmax"], rec["W_fro"]), flush=True)
                if not np.isfinite(rec["W_absmax"]):
                    break
        result["runs"].append({"lr": lr, "trajectory": traj})
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "w") as fh:
            json.dump(result, fh, indent=1)


def compare(a_path, b_path):
    a, b = json.load(open(a_path)), json.load(open(b_path))
    print("# %s vs %s, shape %s" % (a["engine"], b["engine"], a["shape"]))
    worst = 0.0
    for ra, rb in zip(a["runs"], b["runs"]):
        assert ra["lr"] == rb["lr"]
        print("lr %g" % ra["lr"])
        print("  step   cost(%s)   cost(%s)   rel     |W|max(%s)  |W|max(%s)  rel" % (a["engine"], b["engine"], a["engine"], b["engine"]))
        for x, y in zip(ra["trajectory"], rb["trajectory"]):
            assert x["step"] == y["step"]
            rc = abs(x["cost"] - y["cost"]) / max(abs(x["cost"]), 1e-300)
            rw = abs(x["W_absmax"] - y["W_absmax"]) / max(abs(x["W_absmax"]), 1e-300)
            print("  %4d  %-11.6g %-11.6g %-7.1e %-11.6g %-11.6g %-7.1e" % (x["step"], x["cost"], y["cost"], rc, x["W_absmax"], y["W_absmax"], rw))
            if x["W_absmax"] < 1e6:           # while the run is still in float32's comfortable range
                worst = max(worst, rc, rw)
    print("worst relative difference while |W|max < 1e6: %.2e" % worst)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", choices=("oracle", "hip"), default="oracle")
    ap.add_argument("--lrs", default="0.005,0.001")
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--out", default="")
    ap.add_argument("--compare", nargs=2)
    a = ap.parse_args()
    if a.compare:
        compare(*a.compare)
    else:
        run(a.engine, [float(x) for x in a.lrs.split(",")], a.steps, a.every, a.out)
