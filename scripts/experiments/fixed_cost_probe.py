#!/usr/bin/env python
"""(Needs the diagnostic variants: `git apply scripts/experiments/diagnostic_variants.patch` first -- they no longer live in
the product kernels.)
Where the fixed cost of a forward GEMM launch sits OUTSIDE the kernel's own instructions.

In-kernel stamps (planes_stamps.py) see first-workgroup-start -> last-store-acknowledged = 21.9 us of the 25.6 us the
kernel trace reports for the propup launch.  This probe prices the rest with diagnostic builds that change only how many
bytes the launch leaves DIRTY in L2 when it ends (the end-of-kernel release writes them back before the dispatch
completes): the product build (16.8 MB of split-K partials), every split storing into slab 0 (2.1 MB), no stores.
Each build runs the c2 step under `rocprofv3 --kernel-trace --stats`; the propup / propdown averages are compared.

    python scripts/experiments/fixed_cost_probe.py            (on the GPU box; writes gpurun_out/r03_fixed_cost_*.csv)
"""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
src = [os.path.join(ROOT, "mdbn_amd", "csrc", f) for f in ("mdbn_kernels.hip", "mdbn_planes.hip", "mdbn_small.hip", "mdbn_capi.hip")]

if len(sys.argv) > 1 and sys.argv[1] == "--run":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from mdbn_amd import _lib
    _lib.use_diagnostic_library(sys.argv[2])
    import mdbn_amd
    eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
    V, H, B, N = 4096, 1024, 512, 32768
    g = torch.Generator(device="cpu").manual_seed(0)
    data = mdbn_amd.shared(torch.randn((N, V), generator=g).to(eng.device))
    rbm = mdbn_amd.GRBM(n_visible=V, n_hidden=H, numpy_rng=np.random.RandomState(123))
    _, up = rbm.get_cost_updates(lr=0.0, k=1, lambda_2=0.0, batch_size=B)          # lr 0: wrong partials cannot blow W up
    fn = mdbn_amd.function(up, data)
    perm = torch.from_numpy(np.random.RandomState(1).permutation(N)).to(eng.device)
    for it in range(400):
        mb = it % (N // B)
        fn(indexes=perm[mb * B:(mb + 1) * B], momentum=0.0)
    eng.synchronize()
    sys.exit(0)

rows = []
VARIANTS = (("product", []), ("slab0_only", ["-DPL_DIAG_DIRTY=1"]), ("no_stores", ["-DPL_DIAG_DIRTY=2"]))
if len(sys.argv) > 1 and sys.argv[1] == "stats":      # the tail of the statistics launch (speed epilogue: stores / everything)
    VARIANTS = (("product", []), ("stats_speed_not_stored", ["-DPL_DIAG_STATS=1"]), ("stats_no_speed_epilogue", ["-DPL_DIAG_STATS=2"]))
for tag, flags in VARIANTS:
    so = os.path.join(out, "libmdbn_fixedcost_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                           "-DMDBN_SRC_HASH=\"diag-%s\"" % tag] + flags + src + ["-o", so, "-ldl"])
    d = os.path.join(out, "r03_fixed_cost_%s" % tag)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--",
                           sys.executable, os.path.abspath(__file__), "--run", so], env=env,
                          stdout=open(os.path.join(out, "r03_fixed_cost_%s.log" % tag), "w"), stderr=subprocess.STDOUT)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    for r in csv.DictReader(open(stats)):
        if "mdbn::" in r["Name"]:
            rows.append((tag, r["Name"].replace("void ", "")[:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    # inter-kernel gaps of the steady state from the raw trace
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace))),
                key=lambda x: x[0])
    ev = ev[len(ev) // 2:]
    gaps = {}
    for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
        key = (n0.replace("void ", "")[:44], n1.replace("void ", "")[:44])
        gaps.setdefault(key, []).append((s1 - e0) / 1e3)
    import statistics
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
        rows.append((tag, "gap %s -> %s" % (a, b), len(v), statistics.median(v)))
    subprocess.call(["rm", "-rf", d])
with open(os.path.join(out, "r03_fixed_cost_probe%s.log" % ("_stats" if VARIANTS[1][0].startswith("stats") else "")), "w") as f:
    for tag, name, n, us in rows:
        line = "%-11s %-100s x%-6d %8.2f us" % (tag, name, n, us)
        print(line); f.write(line + "\n")
