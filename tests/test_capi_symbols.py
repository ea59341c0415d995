"""The C-ABI library loads without a GPU and exports exactly what include/mdbn_hip.h
declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

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
    assert lib.mdbn_version() == 2
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


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors of the by-pointer structs against the header AS A C COMPILER LAYS IT OUT: gcc compiles
    include/mdbn_hip.h (plain C) and prints sizeof / offsetof of every field; names, order, offsets and sizes must be
    those of mdbn_amd._lib.  Both structs start with `struct_size` (MDBN_VERSION 2) and the binding fills it in."""
    import shutil
    import subprocess
    from mdbn_amd import _lib
    assert C.sizeof(_lib.Rng) == 32
    a, u = _lib.CdArgs(), _lib.UpdateArgs()
    assert a.struct_size == C.sizeof(_lib.CdArgs) and u.struct_size == C.sizeof(_lib.UpdateArgs)
    assert _lib.CdArgs._fields_[0][0] == "struct_size" and _lib.UpdateArgs._fields_[0][0] == "struct_size"
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    text = open(os.path.join(ROOT, "include", "mdbn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "mdbn_hip.h"', 'int main(void) {']
    mirrors = {"mdbn_rng": _lib.Rng, "mdbn_cd_args": _lib.CdArgs, "mdbn_update_args": _lib.UpdateArgs}
    for cname, mirror in mirrors.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = [n.strip().lstrip("*").strip() for n in decl.split(",")]
            names[0] = re.split(r"[\s\*]+", names[0])[-1]
            fields += names
        assert fields == [f[0] for f in mirror._fields_], cname
        prog.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in fields:
            prog.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    prog.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = str(tmp_path / "layout")
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    for line in subprocess.check_output([exe]).decode().split("\n"):
        if not line:
            continue
        key, val = line.split()
        if "." in key:
            cname, f = key.split(".")
            assert getattr(mirrors[cname], f).offset == int(val), key
        else:
            assert C.sizeof(mirrors[key]) == int(val), key


def test_engine_fails_loudly_without_gpu(built_lib):
    import torch
    import pytest
    import mdbn_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mdbn_amd.MdbnError):
        mdbn_amd.HipEngine()


@pytest.mark.parametrize("tiles,stages,P", [(32, 128, 224), (128, 32, 224), (256, 32, 224), (256, 32, 255), (256, 32, 256),
                                            (144, 64, 199), (300, 24, 100), (8, 32, 64), (1, 40, 7), (257, 4, 31)])
def test_balanced_cut_covers_every_stage_once(built_lib, tiles, stages, P):
    """The cut behind mdbn_cd_args.comm_cus (csrc/mdbn_planes.hip, "BALANCED launches"), checked on the host: the
    segments of the P workgroups tile the (tile, stage) grid exactly once, a workgroup's stages differ by at most one,
    whole tiles are swept from stage 0, the workgroups that share a tile are consecutive and as many as the slab count
    the epilogue is told, and only a tile's LAST sharer holds the piece that ends it (what the fix-up kernel assumes)."""
    import ctypes as C
    from mdbn_amd import _lib
    lib = _lib.load()
    out = (C.c_int32 * 6)()
    seen = {}
    nts = []
    for w in range(P):
        _lib.check(lib.mdbn_bal_segment(tiles, stages, P, w, 0, out), "mdbn_bal_segment")
        nseg, nt = out[3], out[4]
        nts.append(nt)
        total = 0
        for k in range(nseg):
            _lib.check(lib.mdbn_bal_segment(tiles, stages, P, w, k, out), "mdbn_bal_segment")
            t, s0, s1, slabs = out[0], out[1], out[2], out[5]
            assert 0 <= t < tiles and 0 <= s0 < s1 <= stages
            total += s1 - s0
            for st in range(s0, s1):
                assert (t, st) not in seen, "stage covered twice"
                seen[(t, st)] = (w, s1 == stages, slabs)
        assert total == nt
    assert len(seen) == tiles * stages
    whole = (tiles // P) * P
    if tiles * stages - whole * stages >= P:
        assert max(nts) - min(nts) <= 1
    for t in range(tiles):
        owners = sorted({seen[(t, st)][0] for st in range(stages)})
        slabs = seen[(t, 0)][2]
        assert owners == list(range(owners[0], owners[-1] + 1)) and len(owners) == slabs
        if t < whole:
            assert slabs == 1
        assert seen[(t, stages - 1)][0] == owners[-1]
