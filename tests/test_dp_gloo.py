"""Data-parallel path on CPU: world_size 2 over gloo with the checker engine.

  * N ranks on row shards of each minibatch == one process on the whole minibatch with
    batch_size = global size, for GRBM and Bernoulli RBM, even and ragged shards;
  * the overlapped mode (all-reduce of step t hidden behind step t+1, speed update deferred)
    is BIT-identical to the synchronous order, including after flush();
  * a rank that owns no row of a short minibatch still takes part in the collective.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_problem(gauss):
    V, H, N = 12, 7, 40
    rs = np.random.RandomState(0)
    data = rs.normal(size=(N, V)) if gauss else (rs.uniform(size=(N, V)) < 0.4).astype(np.float64)
    batches = [rs.permutation(N)[:n].astype(np.int64) for n in (10, 10, 7, 10, 1, 10)]
    return V, H, data, batches


def run_steps(gauss, overlap, group_mode):
    """Runs in every rank (or alone): returns final parameters, speeds and costs."""
    import mdbn_amd
    from _oracle_engine import OracleEngine
    eng = mdbn_amd.set_engine(OracleEngine())
    V, H, data, batches = make_problem(gauss)
    cls = mdbn_amd.GRBM if gauss else mdbn_amd.RBM
    rbm = cls(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5),
              theano_rng=mdbn_amd.RandomStreams(11), engine=eng)
    hp = dict(lambda_2=0.1) if gauss else dict(weightcost=2e-4)
    _, up = rbm.get_cost_updates(lr=0.05, k=2, batch_size=10, **hp)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng),
                           data_parallel="auto" if group_mode else None, overlap=overlap)
    assert fn.overlap == (overlap and group_mode)
    costs = [fn(indexes=b, momentum=0.5) for b in batches]
    mid_speed = rbm.W_speed.get_value().copy()          # reading a speed flushes the pipeline
    costs = [float(c) for c in costs]
    fn.flush()
    return dict(W=rbm.W.tensor.numpy().copy(), Ws=rbm.W_speed.tensor.numpy().copy(),
                hb=rbm.hbias.tensor.numpy().copy(), vbs=rbm.vbias_speed.tensor.numpy().copy(),
                mid=mid_speed, costs=np.array(costs))


def run_pcd(group_mode):
    """PCD-2 with the persistent chain sharded by global row (rbm.py:308-311,369) and the pseudo-likelihood
    monitor summed over ranks."""
    import mdbn_amd
    from _oracle_engine import OracleEngine
    eng = mdbn_amd.set_engine(OracleEngine())
    V, H, data, _ = make_problem(0)
    rs = np.random.RandomState(3)
    batches = [rs.permutation(len(data))[:9].astype(np.int64) for _ in range(5)]      # 9 rows: ragged over 2 ranks
    rbm = mdbn_amd.RBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(5),
                       theano_rng=mdbn_amd.RandomStreams(11), engine=eng)
    chain = mdbn_amd.shared(np.zeros((9, H)), engine=eng)
    _, up = rbm.get_cost_updates(lr=0.05, k=2, batch_size=9, weightcost=2e-4, persistent=chain)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel="auto" if group_mode else None)
    costs = [float(fn(indexes=b, momentum=0.5)) for b in batches]
    lo, hi = fn.group.shard(9) if fn.group is not None else (0, 9)
    return dict(W=rbm.W.tensor.numpy().copy(), Ws=rbm.W_speed.tensor.numpy().copy(), costs=np.array(costs),
                chain=chain.tensor.numpy().copy(), span=np.array([lo, hi]), bit=np.array([rbm.bit_i_idx]))


def run_mdbn(group_mode):
    """The c5 stacking (MDBN.train_bottom_layer x 2 modalities + a joint layer) with every layer's step function data
    parallel: nothing in DBN.training / MDBN knows about ranks, the step functions shard each minibatch themselves."""
    import mdbn_amd
    from mdbn_amd import MDBN
    from _oracle_engine import OracleEngine
    mdbn_amd.set_engine(OracleEngine())
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(0)
    rng = np.random.RandomState(123)
    np.random.seed(7)                    # the shuffles (utils.py:62) must agree on every rank
    outs, Ws = [], []
    for width, sizes in ((20, [8, 4]), (12, [5])):
        x = rs.normal(size=(24, width))
        net, out_t, _ = MDBN.train_bottom_layer(x, None, batch_size=8, k=2, layers_sizes=sizes,
                                                pretraining_epochs=[3] * len(sizes),
                                                pretrain_lr=[0.005] + [0.1] * (len(sizes) - 1), rng=rng)
        outs.append(out_t)
        Ws.extend(p.get_value() for p in net.params)
    joint = np.concatenate(outs, axis=1)
    top = mdbn_amd.DBN(numpy_rng=rng, n_ins=joint.shape[1], gauss=False, hidden_layers_sizes=[6], n_outs=3)
    top.training(mdbn_amd.shared(joint), batch_size=8, k=1, pretraining_epochs=[3, 3], pretrain_lr=[0.1, 0.1])
    Ws.extend(p.get_value() for p in top.params)
    return {"p%d" % i: w for i, w in enumerate(Ws)} | {"out": top.get_output(joint)}


def run_modalities(group_mode, engine=None, rows=24, batch=8, widths=((16, [5], 3), (20, [8, 4], 1), (10, [6, 3], 1))):
    """configs[4]'s three modalities placed MODALITY-PARALLEL (MDBN.train_modalities): rank i % N trains modality i alone,
    parameters are broadcast once, the joint layer follows on the concatenated outputs.  ``engine``: the CPU checker by
    default; tests/test_gpu_dp.py passes a HipEngine (device tensors through the same broadcasts)."""
    import mdbn_amd
    from mdbn_amd import MDBN
    if engine is None:
        from _oracle_engine import OracleEngine
        engine = OracleEngine()
    mdbn_amd.set_engine(engine)
    mdbn_amd.DBN.verbose = False
    rs = np.random.RandomState(0)
    rng = np.random.RandomState(123)
    specs = []
    for width, sizes, k in widths:
        x = rs.normal(size=(rows, width)).astype(np.float32)
        specs.append(dict(train_set=x, validation_set=x[:6], batch_size=batch, k=k, layers_sizes=sizes,
                          pretraining_epochs=[4] * len(sizes), pretrain_lr=[0.005] + [0.1] * (len(sizes) - 1),
                          lambda_1=0.01, lambda_2=0.1))
    trained = MDBN.train_modalities(specs, rng, group="auto" if group_mode else None, shuffle_seed=0)
    out = {}
    for m, (net, out_t, out_v) in enumerate(trained):
        for i, p in enumerate(net.params):
            out["m%d_p%d" % (m, i)] = p.get_value()
        for i, r in enumerate(net.rbm_layers):
            out["m%d_vb%d" % (m, i)] = r.vbias.get_value()
            out["m%d_ws%d" % (m, i)] = r.W_speed.get_value()
            out["m%d_ctr%d" % (m, i)] = np.array([r._rng_step, r.bit_i_idx])
        out["m%d_out" % m], out["m%d_val" % m] = out_t, out_v
    joint = np.concatenate([t[1] for t in trained], axis=1)
    top = mdbn_amd.DBN(numpy_rng=rng, n_ins=joint.shape[1], gauss=False, hidden_layers_sizes=[6], n_outs=3)
    top.shuffle_rng = np.random.RandomState(99)
    top.training(mdbn_amd.shared(joint), batch_size=batch, k=1, pretraining_epochs=[3, 3], pretrain_lr=[0.1, 0.1])
    for i, p in enumerate(top.params):
        out["top_p%d" % i] = p.get_value()
    out["classes"] = top.get_output(joint)
    return out


def run_interleaved(overlap, group_mode):
    """Two step functions of the SAME shape (two equal-sized modalities) called alternately: with the
    all-reduce of each deferred by one call, neither may see the other's pending statistics."""
    import mdbn_amd
    from _oracle_engine import OracleEngine
    eng = mdbn_amd.set_engine(OracleEngine())
    V, H, data, batches = make_problem(1)
    out = {}
    fns = []
    for name, seed in (("a", 5), ("b", 6)):
        rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(seed),
                            theano_rng=mdbn_amd.RandomStreams(11 + seed), engine=eng)
        _, up = rbm.get_cost_updates(lr=0.05, k=1, batch_size=10, lambda_2=0.1)
        fns.append((name, rbm, mdbn_amd.function(up, mdbn_amd.shared(data + seed, engine=eng),
                                                 data_parallel="auto" if group_mode else None, overlap=overlap)))
    costs = {"a": [], "b": []}
    for b in batches:
        for name, rbm, fn in fns:
            costs[name].append(fn(indexes=b, momentum=0.5))
    for name, rbm, fn in fns:
        fn.flush()
        out["W_" + name] = rbm.W.tensor.numpy().copy()
        out["Ws_" + name] = rbm.W_speed.tensor.numpy().copy()
        out["costs_" + name] = np.array([float(c) for c in costs[name]])
    return out


def worker(rank, world, port, outdir, gauss, overlap):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from mdbn_amd import dist
    dist.init_from_env(backend="gloo")
    if gauss == 6:                     # the opt-in bfloat16 wire format (not a parity path)
        os.environ["MDBN_WIRE_BF16"] = "1"
        out = run_steps(1, overlap, True)
    elif gauss == 3:
        out = run_pcd(True)
    elif gauss == 4:
        out = run_mdbn(True)
    elif gauss == 5:
        out = run_modalities(True)
    else:
        out = run_interleaved(overlap, True) if gauss == 2 else run_steps(gauss, overlap, True)
    np.savez(os.path.join(outdir, "rank%d_%d_%d.npz" % (rank, gauss, overlap)), **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.fixture(scope="module")
def dp_results():
    res = {}
    with tempfile.TemporaryDirectory() as d:
        # 2 = two step functions interleaved, 3 = PCD (synchronous only), 4 = MDBN stack (row-sharded), 5 = modality-parallel
        for gauss in (1, 0, 2, 3, 4, 5, 6):
            for overlap in ((0,) if gauss in (3, 4, 5) else (0, 1)):
                mp.spawn(worker, args=(2, free_port(), d, gauss, overlap), nprocs=2, join=True)
                res[(gauss, overlap)] = [dict(np.load(os.path.join(d, "rank%d_%d_%d.npz" % (r, gauss, overlap))))
                                         for r in range(2)]
    return res


@pytest.mark.parametrize("gauss", [1, 0])
def test_dp_equals_single_process(dp_results, gauss):
    sys.path.insert(0, HERE)
    single = run_steps(gauss, False, False)
    for overlap in (0, 1):
        r0, r1 = dp_results[(gauss, overlap)]
        for k in single:
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: %s" % k      # identical on every rank
            np.testing.assert_allclose(r0[k], single[k], rtol=1e-11, atol=1e-13, err_msg=k)


@pytest.mark.parametrize("gauss", [1, 0])
def test_overlap_is_bit_identical_to_sync(dp_results, gauss):
    sync, ovl = dp_results[(gauss, 0)][0], dp_results[(gauss, 1)][0]
    for k in sync:
        assert np.array_equal(sync[k], ovl[k]), k


def test_bf16_wire_format_is_an_opt_in_reporting_mode(dp_results):
    """MDBN_WIRE_BF16=1 (half the bytes per all-reduce): the statistics cross the wire as bfloat16 and come
    back widened.  Not a parity path -- the replicas still agree bit for bit with each other, the overlapped order with the
    synchronous one, and the run stays within bfloat16's 8 bits of the float wire's; it must NOT equal it."""
    exact = dp_results[(1, 0)][0]
    for overlap in (0, 1):
        r0, r1 = dp_results[(6, overlap)]
        for k in exact:
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: %s" % k
            assert np.abs(r0[k] - exact[k]).max() <= 2e-2 * max(1e-3, np.abs(exact[k]).max()), k
        assert not np.array_equal(r0["Ws"], exact["Ws"]), "the wire format was not used"
    for k in exact:
        assert np.array_equal(dp_results[(6, 0)][0][k], dp_results[(6, 1)][0][k]), k


def test_two_step_functions_of_equal_shape_do_not_share_pending_statistics(dp_results):
    sys.path.insert(0, HERE)
    single = run_interleaved(False, False)
    sync, ovl = dp_results[(2, 0)], dp_results[(2, 1)]
    for k in single:
        assert np.array_equal(sync[0][k], ovl[0][k]), k                 # overlapped == synchronous, bit for bit
        assert np.array_equal(ovl[0][k], ovl[1][k]), k                  # replicas agree
        np.testing.assert_allclose(ovl[0][k], single[k], rtol=1e-11, atol=1e-13, err_msg=k)


def test_pcd_under_data_parallelism(dp_results):
    sys.path.insert(0, HERE)
    single = run_pcd(False)
    r0, r1 = dp_results[(3, 0)]
    for k in ("W", "Ws", "costs", "bit"):
        assert np.array_equal(r0[k], r1[k]), k                        # replicas agree
        np.testing.assert_allclose(r0[k], single[k], rtol=1e-11, atol=1e-13, err_msg=k)
    # every rank holds the true chain on the rows it owns
    for r in (r0, r1):
        lo, hi = r["span"]
        assert hi > lo and np.array_equal(r["chain"][lo:hi], single["chain"][lo:hi])


def test_mdbn_stack_under_data_parallelism(dp_results):
    """BASELINE configs[4]'s structure on two ranks: identical replicas, equal to the single-process run."""
    sys.path.insert(0, HERE)
    single = run_mdbn(False)
    r0, r1 = dp_results[(4, 0)]
    for k in single:
        assert np.array_equal(r0[k], r1[k]), k
        np.testing.assert_allclose(r0[k], single[k], rtol=2e-5, atol=1e-7, err_msg=k)      # float32 get_value round trips


def test_modality_parallel_placement_equals_sequential(dp_results):
    """SURVEY 8e's alternative for configs[4]: three modalities over two ranks (rank 0 trains modalities 0 and 2, rank 1
    modality 1), no collective per step, one parameter broadcast per modality -- every array, counter and output equals
    the single-process sequential run with the same per-modality shuffle streams BIT FOR BIT; the joint layer on top
    (row-sharded over the ranks again) agrees to summation order."""
    sys.path.insert(0, HERE)
    single = run_modalities(False)
    r0, r1 = dp_results[(5, 0)]
    for k in single:
        assert np.array_equal(r0[k], r1[k]), "replicas differ: " + k
        if k.startswith("m"):
            assert np.array_equal(r0[k], single[k]), k
        else:
            np.testing.assert_allclose(r0[k], single[k], rtol=2e-5, atol=1e-7, err_msg=k)


def test_shard_bounds():
    from mdbn_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------------------------------------------------------------
# Which collective a group uses (VERDICT r4 #5a): the C-ABI RCCL collective by default ONCE IT HAS PROVEN
# ALIVE on every rank, torch.distributed's otherwise; the bfloat16 wire flag must agree between the ranks.
# ---------------------------------------------------------------------------------------------------------------------
class _FakeLib(object):
    """Stands in for libmdbn_hip.so's communicator entry points (no GPU here)."""

    def __init__(self):
        self.destroyed = 0

    def mdbn_comm_unique_id(self, buf):
        return 0

    def mdbn_comm_destroy(self, ctx):
        self.destroyed += 1
        return 0


class _FakeEngine(object):
    def __init__(self):
        self.ctx, self.lib, self.device = object(), _FakeLib(), torch.device("cpu")


def _collective_worker(rank, world, port, outdir, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("MDBN_DP_COLLECTIVE", None)
    torch.set_num_threads(1)
    from mdbn_amd import dist
    dist.init_from_env(backend="gloo")
    out = {}
    if case == "fallback":
        class G(dist.Group):                         # a group that believes the C-ABI collective is possible here ...
            def _native_possible(self, engine):
                return engine is not None

            def _build_native(self, engine, ident):  # ... whose smoke all-reduce fails on rank 1 only
                if self.rank == 1:
                    raise RuntimeError("stubbed: the small all-reduce did not complete")
                return "side-stream-of-rank-0"
        g, eng = G(), _FakeEngine()
        assert g.native is None                      # auto
        x = torch.full((8,), float(rank + 1), dtype=torch.float64)
        g.all_reduce_sum(x, eng)                     # probes once, agrees to fall back, reduces through torch.distributed
        y = torch.full((8,), 1.0, dtype=torch.float64)
        g.all_reduce_sum_async(y, eng).wait()
        out = dict(sum=x.numpy(), sum2=y.numpy(), torch_collective=np.array("torch.distributed" in g.collective),
                   has_error=np.array(g.native_error is not None), destroyed=np.array(eng.lib.destroyed),
                   installed=np.array(g._comm_engine is not None), auto_off=np.array(g._auto_off))
    elif case == "alive":
        class G(dist.Group):
            def _native_possible(self, engine):
                return engine is not None

            def _build_native(self, engine, ident):
                return "side"
        g, eng = G(), _FakeEngine()
        ok = g.probe_native(eng)
        out = dict(ok=np.array(ok), capi=np.array("C-ABI" in g.collective), err=np.array(g.native_error is None))
    elif case == "wire_mismatch":
        if rank == 0:
            os.environ["MDBN_WIRE_BF16"] = "1"
        else:
            os.environ.pop("MDBN_WIRE_BF16", None)
        try:
            dist.Group()
            out = dict(raised=np.array(False))
        except RuntimeError as exc:
            out = dict(raised=np.array("differs between the ranks" in str(exc)))
    elif case == "wire_with_capi":
        os.environ["MDBN_WIRE_BF16"] = "1"
        try:
            dist.Group(native=True)
            out = dict(raised=np.array(False))
        except RuntimeError as exc:
            out = dict(raised=np.array("cannot be combined" in str(exc)))
    np.savez(os.path.join(outdir, "coll_%s_%d.npz" % (case, rank)), **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("case", ["fallback", "alive", "wire_mismatch", "wire_with_capi"])
def test_collective_choice(case):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_collective_worker, args=(2, free_port(), d, case), nprocs=2, join=True)
        r = [dict(np.load(os.path.join(d, "coll_%s_%d.npz" % (case, k)))) for k in range(2)]
    if case == "fallback":
        for k in range(2):
            np.testing.assert_array_equal(r[k]["sum"], np.full(8, 3.0))          # 1 + 2 through the fallback collective
            np.testing.assert_array_equal(r[k]["sum2"], np.full(8, 2.0))
            assert r[k]["torch_collective"] and r[k]["has_error"] and r[k]["auto_off"] and not r[k]["installed"]
        assert int(r[0]["destroyed"]) == 1 and int(r[1]["destroyed"]) == 0       # the rank that had come up tore its communicator down
    elif case == "alive":
        assert all(bool(x["ok"]) and bool(x["capi"]) and bool(x["err"]) for x in r)
    else:
        assert all(bool(x["raised"]) for x in r)
