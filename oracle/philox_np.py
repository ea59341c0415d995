"""numpy twin of the device RNG (Philox4x32-10, Salmon et al. SC'11).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws its randomness
from Theano's MRG_RandomStreams (rbm.py:40,92,210,237,656), whose stream layout is
not reproducible offline; parity is therefore defined on *identical uniforms*.
This module reproduces, bit for bit, the uniforms/normals the HIP kernels consume
(mdbn_amd/csrc/philox.h) so the oracle can be driven with them.

Draw addressing (the contract shared with the device code):

    counter = (col, global_row >> 2, draw, step)       key = (seed_lo, seed_hi ^ stream)
    word    = philox4x32_10(counter, key)[global_row & 3]
    uniform = ((word >> 8) + 0.5) * 2**-24             in the open interval (0, 1), exact in f32

``draw`` numbers the random matrices inside one step: 0 = positive-phase hidden
sample, 2t-1 = visible draw of Gibbs step t, 2t = hidden draw of Gibbs step t
(t = 1..k).  Normals (GRBM, error_free=False, rbm.py:656) are Box-Muller on the
pair (draw, draw | 0x80000000): z = sqrt(-2 ln u1) * cos(2 pi u2).
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
NORMAL_BIT = 0x80000000


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Counters are uint32 arrays (broadcastable),
    keys are Python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *(np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0                      # 32x32 -> 64 bit products
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _words(rows, cols, seed, stream, step, draw, row_offset):
    seed = int(seed)
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) & 0xFFFFFFFF) ^ (int(stream) & 0xFFFFFFFF)
    grow = np.arange(rows, dtype=np.uint64) + np.uint64(row_offset)
    col = np.arange(cols, dtype=np.uint64)
    out = philox4x32_10(col[None, :], (grow >> np.uint64(2))[:, None],
                        np.uint64(int(draw) & 0xFFFFFFFF),
                        np.uint64(int(step) & 0xFFFFFFFF), k0, k1)
    sel = (grow & np.uint64(3)).astype(np.int64)[:, None]
    stacked = np.stack(out, axis=0)                      # [4, rows, cols]
    return np.take_along_axis(stacked, sel[None, :, :].repeat(cols, axis=2), axis=0)[0]


def u32_to_uniform(w):
    """uint32 word -> float32 uniform in (0,1); every value is exact in f32."""
    return ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def uniform(rows, cols, seed, stream, step, draw, row_offset=0):
    """float32 [rows, cols] uniforms exactly as the device draws them."""
    return u32_to_uniform(_words(rows, cols, seed, stream, step, draw, row_offset))


def normal(rows, cols, seed, stream, step, draw, row_offset=0):
    """float32 [rows, cols] N(0,1) via Box-Muller; device uses the same f32 formula,
    so agreement is to a few ulp (libm vs ocml), not bit-exact."""
    u1 = uniform(rows, cols, seed, stream, step, draw, row_offset).astype(np.float64)
    u2 = uniform(rows, cols, seed, stream, step, int(draw) | NORMAL_BIT,
                 row_offset).astype(np.float64)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


class PhiloxDraws(object):
    """The sequence of random matrices one CD-k step consumes, addressed as above.

    Order mirrors the graph of rbm.py:303,318-336: hidden draw of the positive
    phase, then per Gibbs step the visible draw followed by the hidden draw.
    """

    def __init__(self, seed, stream, step, row_offset=0):
        self.seed, self.stream, self.step, self.row_offset = seed, stream, step, row_offset

    def u(self, draw, rows, cols):
        return uniform(rows, cols, self.seed, self.stream, self.step, draw, self.row_offset)

    def z(self, draw, rows, cols):
        return normal(rows, cols, self.seed, self.stream, self.step, draw, self.row_offset)
