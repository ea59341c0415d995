"""The C-ABI library loads without a GPU and exports exactly what include/mdbn_hip.h
declares (no compute calls here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "mdbn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*int\s+(mdbn_\w+)\s*\(", text, flags=re.M)))


def test_header_matches_binding_table():
    from mdbn_amd import _lib
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(built_lib):
    lib = C.CDLL(built_lib)
    for name in header_functions():
        assert hasattr(lib, name), name


def test_sizes_and_errors_without_gpu(built_lib):
    from mdbn_amd import _lib
    lib = _lib.load()
    assert lib.mdbn_version() == 1
    n = C.c_int64()
    assert lib.mdbn_workspace_bytes(512, 4096, 1024, C.byref(n)) == 0 and n.value >= 8 * 512 * 1024 * 4
    assert lib.mdbn_stats_floats(4096, 4096, 1024, C.byref(n)) == 0
    assert n.value == 4096 * 1024 + 1024 + 4096 + 4
    assert lib.mdbn_stats_floats(6, 8, 8, C.byref(n)) == 0 and n.value == 6 * 8 + 8 + 8 + 4
    for cols, want in ((6, 8), (500, 500), (784, 784), (1024, 1024), (4096, 4096), (255, 256)):
        assert lib.mdbn_padded_ld(cols, C.byref(n)) == 0 and n.value == want, cols
    from mdbn_amd.engine import padded_ld
    assert all(padded_ld(c) == w for c, w in ((6, 8), (1024, 1024), (4096, 4096), (500, 500)))
    assert lib.mdbn_workspace_bytes(0, 1, 1, C.byref(n)) == -1
    assert "bad arguments" in _lib.last_error()


def test_struct_layouts_match_header():
    """ctypes mirrors of the by-pointer structs: field order/size as the header lays them out."""
    from mdbn_amd import _lib
    assert C.sizeof(_lib.Rng) == 32
    assert _lib.CdArgs.rng.offset % 8 == 0 and _lib.CdArgs.trace_h.offset == _lib.CdArgs.rng.offset + 32
    assert _lib.UpdateArgs.lr.offset == _lib.UpdateArgs.stats.offset + 8
    assert _lib.UpdateArgs.cost_out.offset % 8 == 0


def test_engine_fails_loudly_without_gpu(built_lib):
    import torch
    import pytest
    import mdbn_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mdbn_amd.MdbnError):
        mdbn_amd.HipEngine()
