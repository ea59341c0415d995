#!/usr/bin/env python
"""Feeding a minibatch from a pinned host table WITHOUT a kernel on the CUs: CPU row gather into a pinned staging buffer,
then one H2D copy on a side stream (SDMA).  Prices each stage and what it costs the step to run beside them.
    python scripts/experiments/host_feed_probe.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, mdbn_amd
eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
V, H, B, N = 4096, 1024, 512, 32768
host = torch.randn((N, V), generator=torch.Generator().manual_seed(0)).pin_memory()
perm = torch.from_numpy(np.random.RandomState(1).permutation(N))
staging = [torch.empty((B, V)).pin_memory() for _ in range(3)]
dev = [torch.empty((B, V), device=eng.device) for _ in range(3)]
print("cpus visible: %d, torch threads %d" % (len(os.sched_getaffinity(0)), torch.get_num_threads()), flush=True)
for nt in (1, 2, 4, 8, 16):
    torch.set_num_threads(nt)
    for it in range(3):
        torch.index_select(host, 0, perm[it * B:(it + 1) * B], out=staging[0])
    t0 = time.perf_counter()
    for it in range(40):
        mb = it % (N // B)
        torch.index_select(host, 0, perm[mb * B:(mb + 1) * B], out=staging[it % 3])
    us = (time.perf_counter() - t0) / 40 * 1e6
    print("CPU gather, %2d threads: %.0f us per 8.4 MB minibatch = %.1f GB/s" % (nt, us, 4.0 * B * V / us / 1e3), flush=True)
side = torch.cuda.Stream(eng.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    for it in range(5):
        dev[0].copy_(staging[0], non_blocking=True)
    e0.record(side)
    for it in range(50):
        dev[it % 3].copy_(staging[it % 3], non_blocking=True)
    e1.record(side)
torch.cuda.synchronize()
print("H2D copy of 8.4 MB (pinned, side stream): %.1f us each = %.1f GB/s" % (e0.elapsed_time(e1) * 1e3 / 50, 4.0 * B * V / (e0.elapsed_time(e1) * 1e3 / 50) / 1e3), flush=True)

g = torch.Generator(device="cpu").manual_seed(0)
data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
_, up = rbm.get_cost_updates(lr=0.001, k=1, lambda_2=0.1, batch_size=B)
fn = mdbn_amd.function(up, data)
dperm = perm.to(eng.device)
def steps(n):
    for it in range(n):
        mb, nb = it % (N // B), (it + 1) % (N // B)
        fn(indexes=dperm[mb * B:(mb + 1) * B], momentum=0.0, next_indexes=dperm[nb * B:(nb + 1) * B])
def timed(n=300):
    steps(30); eng.synchronize()
    t0 = time.perf_counter(); steps(n); eng.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
alone = timed()
print("step alone: %.1f us" % alone, flush=True)
with torch.cuda.stream(side):
    e0.record(side)
    for it in range(400):
        dev[it % 3].copy_(staging[it % 3], non_blocking=True)
    e1.record(side)
both = timed()
torch.cuda.synchronize()
print("step beside free-running H2D copies: %.1f us (copies %.1f us each)" % (both, e0.elapsed_time(e1) * 1e3 / 400), flush=True)
# CPU gather threads beside the enqueueing thread
for nt in (4, 8):
    torch.set_num_threads(nt)
    stop = [False]; count = [0]
    def worker():
        it = 0
        while not stop[0]:
            mb = it % (N // B)
            torch.index_select(host, 0, perm[mb * B:(mb + 1) * B], out=staging[it % 3])
            it += 1
        count[0] = it
    th = threading.Thread(target=worker); t0 = time.perf_counter(); th.start()
    both = timed()
    stop[0] = True; th.join(); dt = time.perf_counter() - t0
    print("step beside a CPU gather thread (%d intra-op threads): %.1f us; gathers %.0f us each" % (nt, both, dt / max(1, count[0]) * 1e6), flush=True)
# all three together: gather thread feeding H2D copies on the side stream
torch.set_num_threads(8)
stop = [False]; count = [0]
def feeder():
    it = 0
    evs = [None] * 3
    while not stop[0]:
        mb = it % (N // B); s = it % 3
        if evs[s] is not None:
            evs[s].synchronize()
        torch.index_select(host, 0, perm[mb * B:(mb + 1) * B], out=staging[s])
        with torch.cuda.stream(side):
            dev[s].copy_(staging[s], non_blocking=True)
            evs[s] = torch.cuda.Event(); evs[s].record(side)
        it += 1
    count[0] = it
th = threading.Thread(target=feeder); t0 = time.perf_counter(); th.start()
both = timed(600)
stop[0] = True; th.join(); dt = time.perf_counter() - t0
print("step beside the whole feed (gather thread + H2D): %.1f us; feed delivers a minibatch every %.0f us" % (both, dt / max(1, count[0]) * 1e6), flush=True)
