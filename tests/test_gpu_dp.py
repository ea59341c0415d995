"""Data-parallel rehearsal on the one-GPU box: two processes share cuda:0 and exchange the
statistics over gloo (RCCL needs one GPU per rank), running the real HIP kernels through the
same StepFunction code the 8-GPU bench uses."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHAPES = {"small": (320, 192, 1024, 256),
          "c2": (4096, 1024, 4096, 1024),      # BASELINE configs[2] at two ranks: 512 rows per rank
          "thin": (1200, 340, 512, 40)}        # 20 rows per rank: the thin-batch path (csrc/mdbn_thin.hip) under data parallelism


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_steps(group_mode, overlap, shape="small", resident="device"):
    import mdbn_amd
    V, H, N, BG = SHAPES[shape]
    eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
    rs = np.random.RandomState(0)
    data = rs.normal(size=(N, V)).astype(np.float32)
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123),
                        theano_rng=mdbn_amd.RandomStreams(3), engine=eng)
    _, up = rbm.get_cost_updates(lr=0.002, k=1, lambda_2=0.1, batch_size=BG)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng, resident=resident),
                           data_parallel="auto" if group_mode else None, overlap=overlap)
    costs = []
    batches = [rs.permutation(N)[:BG] for _ in range(7)]
    if resident == "host":          # the epoch's order, as the trainers announce it: every rank's feeder gathers ITS shard
        fn.announce(batches[:6])
    for t in range(6):
        # (with the hint of the next minibatch: the overlapped order gathers it inside its statistics kernel)
        costs.append(fn(indexes=batches[t], momentum=0.3, next_indexes=batches[t + 1]))
    costs = [float(c) for c in costs]
    fn.flush()
    return dict(W=rbm.W.get_value(), Ws=rbm.W_speed.get_value(), vb=rbm.vbias.get_value(),
                hbs=rbm.hbias_speed.get_value(), costs=np.array(costs))


def shadow_worker(rank, world, port, outdir, shape):
    """Synchronous data-parallel steps on a ShadowEngine: every shard's CD step is replayed by the float64 oracle along
    the device's recorded chain, the oracle's shard statistics are summed over the ranks, and each rank's shadow applies
    the global update -- no dependence on which near-tie draw falls which way."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", MDBN_COMM_CUS="0")
    import mdbn_amd
    from mdbn_amd import dist
    from _shadow import ShadowEngine
    from oracle import rbm_np
    dist.init_from_env(backend="gloo")
    V, H, N, BG = SHAPES[shape]
    eng = mdbn_amd.set_engine(ShadowEngine())
    rs = np.random.RandomState(0)
    data = rs.normal(size=(N, V)).astype(np.float32)
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123),
                        theano_rng=mdbn_amd.RandomStreams(3), engine=eng)
    _, up = rbm.get_cost_updates(lr=0.002, k=1, lambda_2=0.1, batch_size=BG)
    fn = mdbn_amd.function(up, mdbn_amd.shared(data, engine=eng), data_parallel="auto", overlap=False)
    for t in range(4):
        fn(indexes=rs.permutation(N)[:BG], momentum=0.3)
    eng.synchronize()
    np.savez(os.path.join(outdir, "shadow%d.npz" % rank), stat_err=eng.stat_err, param_err=eng.param_err(rbm),
             flips=eng.flips, steps=eng.steps, flip_gap=rbm_np.FLIP_GAP["max"], W=rbm.W.get_value())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("shape", ["small", "c2", "thin"])
def test_two_ranks_equal_oracle_on_global_batch(built_lib, shape):
    """configs[2] at two ranks (512 rows per rank at c2) against the float64 oracle on the GLOBAL minibatch, teacher-forced:
    per-shard statistics <= 1e-5 of max, parameters after 4 all-reduced updates within the one-step update
    bound; replicas bitwise equal."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, HERE)
    from _margins import check
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(shadow_worker, args=(2, free_port(), d, shape), nprocs=2, join=True)
        r = [dict(np.load(os.path.join(d, "shadow%d.npz" % k))) for k in range(2)]
    assert np.array_equal(r[0]["W"], r[1]["W"]), "replicas diverged"
    for k in range(2):
        assert int(r[k]["steps"]) == 4
        check("DP 2 ranks %s: shard S / s_h / s_v rel-to-max" % shape, float(r[k]["stat_err"]), 1e-5, "stats")
        check("DP 2 ranks %s: parameters vs global-batch oracle, rel-to-max" % shape, float(r[k]["param_err"]), 2e-6, "update")
        check("DP 2 ranks %s: |u - p| of a flipped draw" % shape, float(r[k]["flip_gap"]), 1e-6, "tie")


def worker(rank, world, port, outdir, overlap, shape, comm_cus=0, fused=1):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      MDBN_COMM_CUS=str(comm_cus), MDBN_DP_FUSED_UPDATE=str(fused))
    from mdbn_amd import dist
    dist.init_from_env(backend="gloo")
    out = run_steps(True, overlap, shape)
    np.savez(os.path.join(outdir, "rank%d_%d.npz" % (rank, overlap + (2 if comm_cus else 0) + (0 if fused else 4))), **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def wire_worker(rank, world, port, outdir, overlap):
    os.environ["MDBN_WIRE_BF16"] = "1"
    worker(rank, world, port, outdir, overlap, "small")


def test_bf16_wire_format_on_device(built_lib):
    """MDBN_WIRE_BF16=1 (opt-in reporting mode, never a parity path): the statistics of the HIP engine cross the wire as
    bfloat16.  Replicas agree bit for bit, overlapped == synchronous, the run stays within bfloat16's 8 bits of the
    float-wire run and differs from it."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(2, free_port(), d, 0, "small"), nprocs=2, join=True)
        exact = dict(np.load(os.path.join(d, "rank0_0.npz")))
        res = {}
        for overlap in (0, 1):
            mp.spawn(wire_worker, args=(2, free_port(), d, overlap), nprocs=2, join=True)
            res[overlap] = [dict(np.load(os.path.join(d, "rank%d_%d.npz" % (r, overlap)))) for r in range(2)]
    for overlap in (0, 1):
        r0, r1 = res[overlap]
        for k in exact:
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
            assert np.abs(r0[k] - exact[k]).max() <= 2e-2 * max(1e-3, np.abs(exact[k]).max()), k
    assert any(not np.array_equal(res[0][0][k], exact[k]) for k in exact), "the wire format was not used"
    for k in exact:
        assert np.array_equal(res[0][0][k], res[1][0][k]), k


@pytest.mark.parametrize("shape", ["small", "c2", "thin"])
def test_two_ranks_equal_one_process_on_device(built_lib, shape):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    single = run_steps(False, False, shape)
    with tempfile.TemporaryDirectory() as d:
        res = {}
        # 0: synchronous; 1: overlapped; 3: overlapped with 32 CUs left to the collective (the default of a real
        # data-parallel job: the plane GEMMs then run balanced on 224 workgroups, another summation grouping)
        # 5 / 7: the overlapped orders with the deferred update as its own launch (MDBN_DP_FUSED_UPDATE=0) instead of inside
        # the statistics GEMM's loader waves (one workgroup per tile: its tile; balanced: a flat share)
        for mode, (overlap, cus, fused) in {0: (0, 0, 1), 1: (1, 0, 1), 3: (1, 32, 2), 5: (1, 0, 0), 7: (1, 32, 0)}.items():
            mp.spawn(worker, args=(2, free_port(), d, overlap, shape, cus, fused), nprocs=2, join=True)
            res[mode] = [dict(np.load(os.path.join(d, "rank%d_%d.npz" % (r, mode)))) for r in range(2)]
    for k in single:                     # the update inside the statistics GEMM == the update launch, bit for bit
        assert np.array_equal(res[1][0][k], res[5][0][k]), k
        assert np.array_equal(res[3][0][k], res[7][0][k]), k
    for overlap in (0, 1, 3):
        r0, r1 = res[overlap]
        for k in single:
            assert np.array_equal(r0[k], r1[k]), "replicas diverged: " + k
            # two K-halves summed by the all-reduce instead of one K-long fp32 chain: rounding of order eps * sqrt(K)
            tol = (2e-6 if shape != "c2" else 5e-5) * max(1.0, np.abs(single[k]).max())
            over = np.abs(r0[k] - single[k]) > tol
            if shape == "small":
                assert not over.any(), (k, np.abs(r0[k] - single[k]).max())
                continue
            # c2: "two ranks == the global batch" is held by test_two_ranks_equal_oracle_on_global_batch (teacher-forced,
            # 1e-5 of max); the comparison with a free-running single process depends on which near-tie draw falls which
            # way (one flip moves a column of S by 1e-3), so here the c2 case keeps only the BITWISE statements:
            # replicas above, overlapped == synchronous below.
    for k in single:                     # overlapped == synchronous, bit for bit (same launch geometry)
        assert np.array_equal(res[0][0][k], res[1][0][k]), k


def test_split_update_phases_equal_fused(hip_engine):
    e = hip_engine
    rs = np.random.RandomState(1)
    from mdbn_amd.engine import padded_ld
    Vv, Hh = 130, 70
    ldh, ldv = padded_ld(Hh), padded_ld(Vv)
    def fresh():
        r = np.random.RandomState(2)
        W = e.to_device(r.normal(0, 0.1, (Vv, Hh)).astype(np.float32))
        Ws = e.to_device(r.normal(0, 0.01, (Vv, Hh)).astype(np.float32))
        vecs = [e.to_device(r.normal(size=n).astype(np.float32)) for n in (Hh, Hh, Vv, Vv)]
        return W, Ws, vecs
    stats = torch.from_numpy(rs.normal(size=Vv * ldh + ldh + ldv + 4).astype(np.float32)).to(e.device)
    stats.view(-1)[:Vv * ldh].view(Vv, ldh)[:, Hh:] = 0
    W, Ws, (hb, hbs, vb, vbs) = fresh()
    c0 = e.apply_update(W, Ws, None, hb, hbs, vb, vbs, stats, 0.05, 0.0, 0.1, 0.0, 0.6, 20.0, 17.0, 0.5, phase=0)
    W2, Ws2, (hb2, hbs2, vb2, vbs2) = fresh()
    e.apply_update(W2, Ws2, None, hb2, hbs2, vb2, vbs2, stats, 0.05, 0.0, 0.1, 0.0, 0.6, 20.0, 17.0, 0.5, phase=2)
    c1 = e.apply_update(W2, Ws2, None, hb2, hbs2, vb2, vbs2, stats, 0.05, 0.0, 0.1, 0.0, 0.6, 20.0, 17.0, 0.5, phase=1)
    for a, b in ((W, W2), (Ws, Ws2), (hb, hb2), (hbs, hbs2), (vb, vb2), (vbs, vbs2)):
        assert torch.equal(a, b)
    assert float(c0) == float(c1)
    with pytest.raises(Exception):       # lambda_1 != 0 cannot be split
        e.apply_update(W, Ws, None, hb, hbs, vb, vbs, stats, 0.05, 0.01, 0.1, 0.0, 0.6, 20.0, 17.0, 0.5, phase=1)


def host_worker(rank, world, port, outdir, resident):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", MDBN_COMM_CUS="0")
    from mdbn_amd import dist
    dist.init_from_env(backend="gloo")
    out = run_steps(True, 1, "c2", resident)
    np.savez(os.path.join(outdir, "rank%d_%s.npz" % (rank, resident)), **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_fed_from_a_host_table(built_lib):
    """Data-parallel step (overlapped order, c2 shape, two ranks on the one GPU) with the training table in pinned host
    memory: every rank's row feeder gathers and uploads the rows of ITS shard of the announced minibatches -- bit for bit the
    device-resident run, on both ranks."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with tempfile.TemporaryDirectory() as d:
        res = {}
        for resident in ("device", "host"):
            mp.spawn(host_worker, args=(2, free_port(), d, resident), nprocs=2, join=True)
            res[resident] = [dict(np.load(os.path.join(d, "rank%d_%s.npz" % (r, resident)))) for r in range(2)]
    for r in range(2):
        for k in res["device"][r]:
            assert np.array_equal(res["device"][r][k], res["host"][r][k]), (r, k)


def capped_worker(rank, world, port, outdir):
    """One-rank RCCL job (the box has one GPU): a channel-capped communicator as bench.py's second sweep stage builds it,
    installed in a data-parallel step function."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as td
    import mdbn_amd
    import bench
    torch.cuda.set_device(0)
    td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
    pg, abandoned = bench.capped_process_group(td, eng, 8, world)
    assert pg is not None and not abandoned, "a one-rank communicator with max_ctas = 8 must come up"
    buf = torch.full((4096 * 1024 + 5124,), 2.0, device=eng.device)
    work = td.all_reduce(buf, group=pg, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    ok = bool((buf == 2.0 * world).all())
    td.destroy_process_group()
    np.save(os.path.join(outdir, "ok.npy"), np.array([ok]))


def test_channel_capped_rccl_communicator(built_lib):
    """bench.py's second sweep stage (N > 1 on RCCL): ProcessGroupNCCL.Options().config.max_ctas -> a second communicator,
    proven with a deadline-guarded all-reduce.  One rank here (RCCL wants one GPU per rank)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(capped_worker, args=(1, free_port(), d), nprocs=1, join=True)
        assert bool(np.load(os.path.join(d, "ok.npy"))[0])


# ---------------------------------------------------------------------------------- modality-parallel placement on the device
MODALITY_SHAPES = dict(rows=96, batch=32, widths=((512, [40], 2), (260, [64, 24], 1), (128, [48, 12], 1)))


def modality_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", MDBN_COMM_CUS="0")
    import mdbn_amd
    from mdbn_amd import dist
    from test_dp_gloo import run_modalities
    dist.init_from_env(backend="gloo")
    out = run_modalities(True, engine=mdbn_amd.HipEngine(), **MODALITY_SHAPES)
    np.savez(os.path.join(outdir, "mod%d.npz" % rank), **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_modality_parallel_placement_on_device(built_lib):
    """MDBN.train_modalities (SURVEY 8e's second placement of configs[4]) with the HIP engine under two ranks: rank 0 trains
    modalities 0 and 2, rank 1 modality 1, each alone on the device; the broadcasts move the PADDED device storages and the
    host counters, W's bf16 planes are invalidated by the version bump.  Every modality's parameters, speeds, counters and
    outputs equal the sequential single-process run BIT FOR BIT on both ranks; the joint layer (row-sharded over the ranks
    again) agrees to summation order."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, HERE)
    import mdbn_amd
    from test_dp_gloo import run_modalities
    single = run_modalities(False, engine=mdbn_amd.HipEngine(), **MODALITY_SHAPES)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(modality_worker, args=(2, free_port(), d), nprocs=2, join=True)
        r0, r1 = [dict(np.load(os.path.join(d, "mod%d.npz" % k))) for k in range(2)]
    for k in single:
        assert np.array_equal(r0[k], r1[k]), "replicas differ: " + k
        if k.startswith("m"):
            assert np.array_equal(r0[k], single[k]), k
        else:
            scale = max(1.0, np.abs(single[k]).max())
            assert np.abs(r0[k] - single[k]).max() <= 5e-5 * scale, k
