"""Philox twins: published known answers, numpy == C == the library's host twin."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import philox_np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Random123 kat_vectors, philox4x32 with 10 rounds
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.fixture(scope="module")
def cref():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "_build", "libphilox_ref.so"))


def test_numpy_known_answers():
    for ctr, key, want in KAT:
        got = philox_np.philox4x32_10(*[np.uint64(c) for c in ctr], key[0], key[1])
        assert tuple(int(x) for x in got) == want


def test_c_known_answers(cref):
    for ctr, key, want in KAT:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        cref.philox_ref_block(c, k, o)
        assert tuple(o) == want


@pytest.mark.parametrize("rows,cols,offset", [(1, 1, 0), (7, 5, 0), (8, 33, 4), (9, 6, 3), (64, 17, 1022)])
def test_numpy_equals_c(cref, rows, cols, offset):
    seed, stream, step, draw = 0x123456789ABCDEF, 3, 41, 2
    want = philox_np.uniform(rows, cols, seed, stream, step, draw, offset)
    out = np.zeros((rows, cols), np.float32)
    cref.philox_ref_uniform(out.ctypes.data_as(C.c_void_p), C.c_int64(rows), C.c_int64(cols), C.c_int64(cols),
                            C.c_uint64(seed), C.c_uint32(stream), C.c_uint32(step), C.c_uint32(draw),
                            C.c_uint64(offset))
    assert np.array_equal(want, out)
    assert want.min() > 0.0 and want.max() < 1.0


def test_library_host_twin(built_lib):
    """mdbn_philox_host (no GPU needed) agrees bit for bit with the numpy twin."""
    from mdbn_amd import _lib
    lib = _lib.load()
    rows, cols, offset = 13, 9, 6
    r = _lib.Rng(77, 2, 5, 3, 0, offset)
    out = np.zeros((rows, cols), np.float32)
    _lib.check(lib.mdbn_philox_host(out.ctypes.data_as(C.c_void_p), rows, cols, cols, C.byref(r)), "host twin")
    assert np.array_equal(out, philox_np.uniform(rows, cols, 77, 2, 5, 3, offset))


def test_row_sharding_invariance():
    """Keyed by GLOBAL row: a shard's draws are the matching rows of the global matrix."""
    full = philox_np.uniform(20, 7, 5, 0, 9, 1, 0)
    for lo, hi in ((0, 10), (10, 20), (3, 11)):
        assert np.array_equal(full[lo:hi], philox_np.uniform(hi - lo, 7, 5, 0, 9, 1, lo))


def test_uniform_statistics():
    u = philox_np.uniform(512, 512, 2024, 0, 0, 0).astype(np.float64).ravel()
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12.0) < 1e-3
    hist, _ = np.histogram(u, bins=64, range=(0, 1))
    chi2 = ((hist - u.size / 64.0) ** 2 / (u.size / 64.0)).sum()
    assert chi2 < 120.0          # 63 dof: P(chi2 > 120) ~ 2e-5
    z = philox_np.normal(256, 256, 2024, 0, 0, 1).astype(np.float64).ravel()
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
