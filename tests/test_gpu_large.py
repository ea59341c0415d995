"""Maximum sizes: a training table beyond 2^31 elements / 8 GiB (MI355X holds 288 GB; the reference's tables are a few
hundred rows, dbn.py:307 gathers them with a symbolic index).  Every way the step reads minibatch rows -- the gather
launch, the gather folded into the statistics kernel's loader waves (next_indexes), the f32-operand path, the lower
layers' forward pass and the pinned host table read over PCIe -- must address rows with 64-bit offsets: each is
compared BIT FOR BIT with the same step on a small table holding just the rows used."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

V, H, B = 4096, 1024, 512
N_BIG = 540_000          # 2.21e9 elements (> 2^31), 8.85 GB


def _batches(n_rows, rs, n_batches=4):
    """Minibatches drawn from both ends of the table: the last 4096 rows (offsets beyond 2^31 elements) and the first 64."""
    pool = np.concatenate([np.arange(64), np.arange(n_rows - 4096, n_rows)])
    return [rs.permutation(pool)[:B].astype(np.int64) for _ in range(n_batches)]


def _train(eng, table, batches, hints, planes=1):
    import mdbn_amd
    eng.set_option("gemm_planes", planes)
    keep, eng.keep_f32 = eng.keep_f32, False          # the product default (the session fixture keeps float32 copies)
    try:
        rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123), theano_rng=mdbn_amd.RandomStreams(3),
                            engine=eng)
        _, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
        fn = mdbn_amd.function(up, table, data_parallel=None)
        idx = [eng.index_tensor(b) for b in batches]
        costs = []
        for t in range(len(idx) - 1):
            kw = dict(next_indexes=idx[t + 1]) if hints else {}
            costs.append(float(fn(indexes=idx[t], momentum=0.3, **kw)))
        if hints and planes and not hasattr(table, "host"):
            assert eng.last_scratch.ahead is not None, "the gather-ahead path did not run"
        return costs, rbm.W.get_value(), rbm.vbias.get_value()
    finally:
        eng.set_option("gemm_planes", 1)
        eng.keep_f32 = keep


def test_rows_beyond_two_to_the_31_elements(hip_engine):
    import mdbn_amd
    eng = hip_engine
    g = torch.Generator(device=eng.device).manual_seed(5)
    big = torch.empty((N_BIG, V), dtype=torch.float32, device=eng.device)
    for r0 in range(0, N_BIG, 60_000):                       # bounded temporaries
        big[r0:r0 + 60_000].normal_(generator=g)
    assert big.numel() > 2 ** 31
    batches = _batches(N_BIG, np.random.RandomState(2))
    used = np.unique(np.concatenate(batches))
    small = big[torch.from_numpy(used).to(eng.device)].clone()
    remap = [np.searchsorted(used, b) for b in batches]
    # the gather itself
    got = eng.gather_rows(big, eng.index_tensor(batches[0]))
    assert torch.equal(got[:, :V], big[torch.from_numpy(batches[0]).to(eng.device)])
    t_small = mdbn_amd.shared(small, engine=eng)
    t_big = mdbn_amd.shared(big, engine=eng)
    del big
    for hints, planes in ((True, 1), (False, 1), (False, 0)):
        want = _train(eng, t_small, remap, hints, planes)
        got = _train(eng, t_big, batches, hints, planes)
        assert want[0] == got[0], (hints, planes, want[0], got[0])
        assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2]), (hints, planes)
    # the same table in pinned host memory: rows gathered over PCIe one minibatch ahead
    want = _train(eng, t_small, remap, True)
    host = t_big.tensor[:, :V].cpu()
    del t_big
    torch.cuda.empty_cache()
    t_host = mdbn_amd.shared(host, engine=eng, resident="host")
    del host
    assert t_host.host.numel() > 2 ** 31 and t_host._mirror is None
    got = _train(eng, t_host, batches, True)
    assert t_host._mirror is None, "training must not upload a host-resident table"
    assert want[0] == got[0]
    assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2])
