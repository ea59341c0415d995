#!/usr/bin/env python
"""What the data-parallel step pays on ONE GPU when a collective-shaped kernel runs beside it.

The overlapped DP code path (cd_step -> async all-reduce -> phase-3 update) runs with a stand-in group whose
"all-reduce" launches scripts/experiments/occupier.hip on the side stream: R workgroups with the register / LDS
footprint of RCCL's gfx950 all-reduce kernel, holding R CUs for D microseconds.  No inter-GPU traffic: this measures
CU sharing only (the thing a 256-workgroup GEMM grid is sensitive to), not xGMI.

    python scripts/dp_contention_probe.py [--blocks 0,16,32,64] [--us 120] [--opt name=value ...]
"""
import argparse, ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, mdbn_amd

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", default="0,16,32,64")
ap.add_argument("--us", type=float, default=120.0)
ap.add_argument("--threads", type=int, default=512)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--events", default="torch", help="torch | hip:<flags hex> | value -- who owns the fork / join: torch events, HIP events with flags, or stream memory operations")
ap.add_argument("--reserve", default="0", help="comma list of comm_cus settings to try for every occupier size")
args = ap.parse_args()

src = os.path.join(ROOT, "scripts", "experiments", "occupier.hip")
so = os.path.join(ROOT, "scripts", "experiments", "liboccupier.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", src, "-o", so])
occ = C.CDLL(so)
occ.occupier_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_double]
occ.occupier_fork.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_double]
occ.occupier_join.argtypes = [C.c_void_p, C.c_int]
occ.occupier_fork_value.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_double]
occ.occupier_join_value.argtypes = [C.c_void_p, C.c_uint]

eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
for o in args.opt:
    k, v = o.split("=")
    eng.set_option(k, int(v))
side = torch.cuda.Stream(device=eng.device)
buf_a = torch.zeros(16 << 20, dtype=torch.uint8, device=eng.device)      # the size of the c2 statistics buffer
buf_b = torch.zeros(16 << 20, dtype=torch.uint8, device=eng.device)


class _Work:
    def __init__(self, ev): self.ev = ev
    def wait(self):
        if isinstance(self.ev, tuple):
            assert occ.occupier_join_value(C.c_void_p(torch.cuda.current_stream().cuda_stream), self.ev[0]) == 0
        elif isinstance(self.ev, int):
            assert occ.occupier_join(C.c_void_p(torch.cuda.current_stream().cuda_stream), self.ev) == 0
        elif self.ev is not None:
            torch.cuda.current_stream().wait_event(self.ev)


value_ops = args.events == "value"           # stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32)
if value_ops:
    rc = occ.occupier_values_init()
    assert rc == 0, "stream value operations unavailable (%d)" % rc
hip_events = args.events.startswith("hip:")
if hip_events:
    assert occ.occupier_events(C.c_uint(int(args.events[4:], 16))) == 0


class OccupierGroup:
    rank, world_size = 0, 2
    blocks = 0
    def shard(self, n): return 0, n                # this "rank" owns every row: per-rank work of a 512-row shard
    def all_reduce_sum(self, t, engine=None): return t
    def all_reduce_sum_async(self, t, engine=None):
        if not self.blocks:
            return _Work(None)
        if value_ops:
            self.n = getattr(self, "n", 0) + 1
            rc = occ.occupier_fork_value(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(side.cuda_stream), self.n,
                                         self.blocks, args.threads, C.c_void_p(buf_a.data_ptr()), C.c_void_p(buf_b.data_ptr()),
                                         C.c_longlong(buf_a.numel()), C.c_double(args.us))
            assert rc == 0, rc
            return _Work((self.n,))
        if hip_events:
            self.n = getattr(self, "n", 0) + 1
            rc = occ.occupier_fork(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(side.cuda_stream), self.n,
                                   self.blocks, args.threads, C.c_void_p(buf_a.data_ptr()), C.c_void_p(buf_b.data_ptr()),
                                   C.c_longlong(buf_a.numel()), C.c_double(args.us))
            assert rc == 0, rc
            return _Work(self.n)
        side.wait_stream(torch.cuda.current_stream())
        rc = occ.occupier_launch(C.c_void_p(side.cuda_stream), self.blocks, args.threads, C.c_void_p(buf_a.data_ptr()),
                                 C.c_void_p(buf_b.data_ptr()), buf_a.numel(), args.us)
        assert rc == 0, rc
        ev = torch.cuda.Event(); ev.record(side)
        return _Work(ev)


V, H, B, N = 4096, 1024, 512, 32768
g = torch.Generator(device="cpu").manual_seed(0)
data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
grp = OccupierGroup()


def make(group):
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
    return mdbn_amd.function(up, data, data_parallel=group)


def run(fn, n):
    for it in range(n):
        mb, nb = it % (N // B), (it + 1) % (N // B)
        fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0, next_indexes=perm[nb * B:(nb + 1) * B])


host_us = [0.0]


def median_us(fn):
    run(fn, 20); eng.synchronize()
    out, host = [], []
    for _ in range(5):
        t0 = time.perf_counter(); run(fn, 100)
        host.append((time.perf_counter() - t0) * 1e4)      # time to ENQUEUE 100 steps (the queue is far from full)
        eng.synchronize()
        out.append((time.perf_counter() - t0) * 1e4)
    host_us[0] = float(np.median(host))
    return float(np.median(out))


single = make(None)
print("single-device fused step                         %.1f us/step  (host enqueue %.1f)" % (median_us(single), host_us[0]), flush=True)
for reserve in [int(x) for x in args.reserve.split(",")]:
    os.environ["MDBN_COMM_CUS"] = str(reserve)       # read by StepFunction, handed to every mdbn_cd_step call
    dp = make(grp)
    assert dp.comm_cus == reserve
    for blocks in [int(x) for x in args.blocks.split(",")]:
        grp.blocks = blocks
        t = median_us(dp)
        print("DP path, comm_cus=%-3d occupier %3d x %d thr, %3.0f us  %.1f us/step  (host enqueue %.1f)"
              % (reserve, blocks, args.threads, args.us, t, host_us[0]), flush=True)
    dp.flush()
