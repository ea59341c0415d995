"""Balanced launches against one-workgroup-per-tile launches, GEMM by GEMM (same inputs, same draws)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mdbn_amd
from mdbn_amd import RngAddr
from mdbn_amd.engine import padded_ld

eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
eng.keep_f32 = True
V, H, B = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 1024, 512))]
cus = [int(x) for x in sys.argv[4:]] or [32]
rs = np.random.RandomState(4)
W = (rs.normal(0, 0.05, (V, H))).astype(np.float32)
hb, vb = rs.normal(0, 0.2, H).astype(np.float32), rs.normal(0, 0.2, V).astype(np.float32)
x = rs.normal(size=(B, V)).astype(np.float32)
dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, hb, vb, x)]


def run(c):
    eng.set_option("bal_blocks", c)
    stats, _ = eng.cd_step(dx, None, dW, dhb, dvb, True, 1, RngAddr(2, 0, 0, 0, 0))
    sc = eng.last_scratch
    torch.cuda.synchronize()
    out = dict(ph=sc.P2[:B].cpu().numpy().copy(), nh=sc.P2[B:2 * B].cpu().numpy().copy(), nv=sc.V2[B:2 * B].cpu().numpy().copy(),
               hs=sc.hs.cpu().numpy().copy(), S=stats[:V * padded_ld(H)].view(V, -1).cpu().numpy().copy())
    eng.set_option("bal_blocks", 0)
    return out


ref = run(0)
ph_np = 1.0 / (1.0 + np.exp(-(x.astype(np.float64) @ W.astype(np.float64) + hb)))
print("one workgroup per tile job: max |ph - float64| %.3e" % np.abs(ref["ph"] - ph_np).max())
for c in cus:
    got = run(c)
    print("bal_blocks=%d: max |ph - float64| %.3e;  ph[0, :4] = %s  (float64 %s)" % (c, np.abs(got["ph"] - ph_np).max(), got["ph"][0, :4], ph_np[0, :4]))
    for k in ("ph", "hs", "nv", "nh", "S"):
        d = np.abs(got[k] - ref[k])
        bad = d > 1e-4 * max(1.0, np.abs(ref[k]).max())
        msg = ""
        if bad.any():
            r, cidx = np.nonzero(bad)
            tiles = sorted(set(zip((r // 128).tolist(), (cidx // 128).tolist())))
            msg = "  bad tiles (tm, tn): %s%s" % (tiles[:24], " ..." if len(tiles) > 24 else "")
        if k == "ph" and bad.any():
            r, cidx = np.nonzero(~bad)
            print("   GOOD elements by row %% 4: %s  by (row %% 64) // 16: %s  by (col %% 64) // 16: %s  by row // 64 %% 2: %s  by col // 64 %% 2: %s"
                  % (np.bincount(r % 4, minlength=4), np.bincount((r % 64) // 16, minlength=4), np.bincount((cidx % 64) // 16, minlength=4),
                     np.bincount((r // 64) % 2, minlength=2), np.bincount((cidx // 64) % 2, minlength=2)))
        print("bal_blocks=%d %-3s max |diff| %.3e  elements off %d / %d%s" % (c, k, d.max(), int(bad.sum()), d.size, msg))
