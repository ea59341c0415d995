"""The PRODUCT configuration -- a fresh HipEngine, no option touched, no float32 inspection copies, the trainers' next_indexes
hints -- on every dispatch path of the CD step, against the float64 oracle.  (The shared `hip_engine` fixture of the other GPU
tests keeps the inspection copies and serves every whole-tile shape on planes; VERDICT r4 called that coverage brittle.)

The product path records no chain, so the oracle cannot be teacher-forced along it.  Instead each shape runs twice on the
same seeds: once in the product configuration, once with chain taps (which turn the hints and the copy-skipping off); the
tapped run is checked against the forced oracle (rbm.py:258-376 step by step), and the product run must reproduce the
tapped run's parameters -- bit for bit where both take the same kernels, to fp32 summation order where the product run is on
planes (the tapped run too) or prepares minibatches ahead (same arithmetic, same order: still bit for bit)."""
import numpy as np
import pytest

from oracle import rbm_np
from oracle.philox_np import PhiloxDraws
from _margins import check

pytestmark = pytest.mark.gpu

CASES = [  # id, V, H, B, k, gauss, hyper-parameters, expected path of the PRODUCT run
    ("small_one_launch", 100, 24, 512, 1, False, dict(lr=0.1, weightcost=2e-4), "small"),
    ("thin_batch20", 784, 500, 20, 1, False, dict(lr=0.1, weightcost=2e-4), "thin"),
    ("thin_ge_layer", 19937, 400, 20, 1, True, dict(lr=0.0005, lambda_2=0.1), "thin"),
    ("stream_c4", 1024, 256, 512, 1, False, dict(lr=0.1, weightcost=2e-4), "stream"),
    ("stream_ge_cd5", 2048, 400, 512, 5, True, dict(lr=0.002, lambda_2=0.1), "stream"),
    ("stream_small_layer", 256, 200, 512, 2, True, dict(lr=0.002, lambda_2=0.1), "stream"),
    ("planes_headline", 4096, 1024, 512, 1, True, dict(lr=0.001, lambda_2=0.1), "planes"),
]
STEPS = 4


def _run(eng, V, H, B, k, gauss, hp, tapped):
    import mdbn_amd
    N = 3 * B + 7
    rs = np.random.RandomState(31)
    data = rs.normal(size=(N, V)).astype(np.float32) if gauss else (rs.uniform(size=(N, V)) < 0.3).astype(np.float32)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(7), engine=eng)
    st = rbm_np.RBMState(V, H, W=rbm.W.get_value(), gauss=gauss)
    if hp.get("weightcost"):
        st.freeze_W0()
    _, updates = rbm.get_cost_updates(k=k, batch_size=B, **hp)
    fn = mdbn_amd.function(updates, mdbn_amd.shared(data, engine=eng), data_parallel=None)
    order = [eng.index_tensor(rs.permutation(N)[:B], N) for _ in range(STEPS + 1)]
    eng.trace_chain = tapped
    eng.kernel_timing(True)
    costs, prepared = [], 0
    try:
        for t in range(STEPS):
            mom = 0.5 if t < 2 else 0.9
            c = fn(indexes=order[t], momentum=mom, next_indexes=order[t + 1])       # (as dbn.py's trainers call it)
            prepared += int(eng.last_scratch.ahead is not None)
            if tapped:
                sc = eng.last_scratch
                forced = (sc.trace_h.cpu().numpy()[:, :, :H], None if gauss else sc.trace_v.cpu().numpy()[:, :, :V])
                want = rbm_np.cd_step(st, data[order[t].cpu().numpy()], PhiloxDraws(7, rbm.stream_id, t), k=k, batch_size=B,
                                      momentum=mom, forced=forced, **hp)
                check("product-defaults bridge (tapped run): cost rel", abs(float(c) - want) / abs(want), 1e-5)
            costs.append(float(c))
        eng.synchronize()
        kinds = [kd for _, _, _, kd in eng.kernel_timing_detail()]
    finally:
        eng.kernel_timing(False)
        eng.trace_chain = False
    params = {n: getattr(rbm, n).get_value() for n in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed")}
    return params, np.array(costs), kinds, prepared, st


@pytest.mark.parametrize("name,V,H,B,k,gauss,hp,path", CASES, ids=[c[0] for c in CASES])
def test_product_configuration_reproduces_the_oracle_checked_run(built_lib, name, V, H, B, k, gauss, hp, path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdbn_amd
    eng = mdbn_amd.HipEngine()                     # product defaults: nothing set
    assert not eng.keep_f32
    tapped, tcosts, _, _, st = _run(eng, V, H, B, k, gauss, hp, tapped=True)
    for pname in ("W", "hbias", "vbias", "W_speed", "hbias_speed", "vbias_speed"):
        ref = getattr(st, pname)
        check("product-defaults bridge (tapped run): %s after %d steps / max" % (pname, STEPS),
              np.abs(tapped[pname] - ref).max() / max(1.0, np.abs(ref).max()), 2e-6, "update")
    prod, pcosts, kinds, prepared, _ = _run(eng, V, H, B, k, gauss, hp, tapped=False)
    # the path the product run took (GEMM launch kinds: 1000 = register streaming, 100 / 200 = bf16 pipe, 2000 = planes)
    if path == "small" or path == "thin":
        assert kinds == [], kinds
    elif path == "stream":
        assert kinds and all(1100 <= kd < 2000 for kd in kinds), kinds
    else:
        assert kinds and all(kd >= 2000 for kd in kinds), kinds
    if path in ("thin", "planes"):
        assert prepared == STEPS, "next_indexes was honoured on %d of %d steps" % (prepared, STEPS)
    for pname in prod:
        np.testing.assert_array_equal(prod[pname], tapped[pname], err_msg="%s: %s" % (name, pname))
    np.testing.assert_array_equal(pcosts, tcosts)
