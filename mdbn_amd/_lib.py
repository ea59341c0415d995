"""ctypes binding of libmdbn_hip.so (include/mdbn_hip.h).  No torch types cross this line:
callers pass tensor.data_ptr() and the raw hipStream_t."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmdbn_hip.so")


class MdbnError(RuntimeError):
    pass


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("stream_id", C.c_uint32), ("step", C.c_uint32),
                ("draw", C.c_uint32), ("reserved", C.c_uint32), ("row_offset", C.c_uint64)]


class _Sized(C.Structure):
    """By-pointer argument structs begin with their own size (mdbn_hip.h, MDBN_VERSION 2): set on construction."""

    def __init__(self, *a, **kw):
        super(_Sized, self).__init__(*a, **kw)
        self.struct_size = C.sizeof(type(self))


class CdArgs(_Sized):
    _fields_ = [("struct_size", C.c_uint64), ("data", C.c_void_p), ("n_data", C.c_int64), ("indexes", C.c_void_p),
                ("index_is_64", C.c_int32), ("gauss", C.c_int32), ("add_noise", C.c_int32),
                ("sample_stats", C.c_int32), ("keep_f32", C.c_int32), ("k", C.c_int32), ("B", C.c_int64), ("V", C.c_int64), ("H", C.c_int64),
                ("ldv", C.c_int64), ("ldh", C.c_int64),
                ("W", C.c_void_p), ("hbias", C.c_void_p), ("vbias", C.c_void_p),
                ("persistent", C.c_void_p),
                ("V2", C.c_void_p), ("P2", C.c_void_p), ("hs", C.c_void_p), ("vs", C.c_void_p),
                ("stats", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("rng", Rng), ("trace_h", C.c_void_p), ("trace_v", C.c_void_p),
                ("planes", C.c_void_p), ("planes_bytes", C.c_int64), ("W_planes", C.c_void_p),
                ("W_planes_valid", C.c_int32), ("comm_cus", C.c_int32),
                ("next_indexes", C.c_void_p), ("planes_alt", C.c_void_p), ("x_buffer", C.c_int32),
                ("v0_ready", C.c_int32), ("ahead_done", C.POINTER(C.c_int32))]


class UpdateArgs(_Sized):
    _fields_ = [("struct_size", C.c_uint64), ("W", C.c_void_p), ("W_speed", C.c_void_p), ("W0", C.c_void_p),
                ("hbias", C.c_void_p), ("hbias_speed", C.c_void_p),
                ("vbias", C.c_void_p), ("vbias_speed", C.c_void_p),
                ("V", C.c_int64), ("H", C.c_int64), ("ldv", C.c_int64), ("ldh", C.c_int64),
                ("stats", C.c_void_p),
                ("lr", C.c_float), ("lambda_1", C.c_float), ("lambda_2", C.c_float),
                ("weightcost", C.c_float), ("momentum", C.c_float),
                ("batch_size", C.c_float), ("n_rows", C.c_float),
                ("cost_scale", C.c_float), ("cost_out", C.c_void_p), ("W_planes", C.c_void_p),
                ("phase", C.c_int32), ("reserved", C.c_int32)]


_i64, _i32, _f32, _vp = C.c_int64, C.c_int, C.c_float, C.c_void_p
_rngp = C.POINTER(Rng)

# name -> argtypes; every function returns int.  Keep in step with include/mdbn_hip.h
# (tests/test_capi_symbols.py parses the header and checks both directions).
SIGNATURES = {
    "mdbn_version": [],
    "mdbn_last_error": [C.c_char_p, C.c_size_t],
    "mdbn_source_hash": [C.c_char_p, C.c_size_t],
    "mdbn_ctx_create": [C.POINTER(_vp), _i32],
    "mdbn_ctx_destroy": [_vp],
    "mdbn_set_option": [_vp, C.c_char_p, _i64],
    "mdbn_bal_segment": [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)],
    "mdbn_kernel_timing": [_vp, _i32],
    "mdbn_kernel_timing_read": [_vp, C.POINTER(_i64), C.POINTER(C.c_double)],
    "mdbn_kernel_timing_detail": [_vp, _i64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(_i64)],
    "mdbn_workspace_bytes": [_i64, _i64, _i64, C.POINTER(_i64)],
    "mdbn_workspace_bytes_ctx": [_vp, _i64, _i64, _i64, C.POINTER(_i64)],
    "mdbn_padded_ld": [_i64, C.POINTER(_i64)],
    "mdbn_planes_bytes": [_i64, _i64, _i64, C.POINTER(_i64)],
    "mdbn_planes_alt_bytes": [_i64, _i64, C.POINTER(_i64)],
    "mdbn_ahead_bytes_ctx": [_vp, _i64, _i64, _i64, _i64, _i64, C.POINTER(_i64)],
    "mdbn_planes_eligible": [_i64, _i64, _i64, _i64, _i64, C.POINTER(_i32)],
    "mdbn_planes_eligible_ctx": [_vp, _i64, _i64, _i64, _i64, _i64, C.POINTER(_i32)],
    "mdbn_split_planes": [_vp, _vp, _vp, _i64, _i64, _vp],
    "mdbn_stats_floats": [_i64, _i64, _i64, C.POINTER(_i64)],
    "mdbn_gather_rows": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i32, _i64, _vp, _i64],
    "mdbn_gather_rows_host": [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i32, _i64, _vp, _i64, _i32, _i32],
    "mdbn_host_gather_rows": [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _i32],
    "mdbn_feeder_create": [_vp, _vp, _i64, _i64, _i64, _i64, _i32, C.POINTER(_vp), _i64, _i32, C.POINTER(_vp)],
    "mdbn_feeder_submit": [_vp, _vp, _i64, C.POINTER(_i64)],
    "mdbn_feeder_acquire": [_vp, _i64, _vp, C.POINTER(_i32)],
    "mdbn_feeder_release": [_vp, _i64, _vp],
    "mdbn_feeder_stats": [_vp, C.POINTER(C.c_double)],
    "mdbn_feeder_cancel": [_vp],
    "mdbn_feeder_destroy": [_vp],
    "mdbn_propup_sample": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp,
                           _vp, _vp, _f32, _vp, _rngp, _vp, _i64],
    "mdbn_propdown_sample": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i32, _i32,
                             _vp, _vp, _vp, _rngp, _vp, _vp, _vp, _i64],
    "mdbn_comm_unique_id": [C.c_char_p],
    "mdbn_comm_init_rank": [_vp, C.c_char_p, _i32, _i32],
    "mdbn_allreduce_stats": [_vp, _vp, _vp, _i64],
    "mdbn_comm_destroy": [_vp],
    "mdbn_gibbs_chain": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i32, _i32, _i64,
                         _vp, _vp, _vp, _vp, _vp, _rngp, _vp, _i64],
    "mdbn_cd_stats": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _i64],
    "mdbn_apply_update": [_vp, _vp, C.POINTER(UpdateArgs)],
    "mdbn_cd_step": [_vp, _vp, C.POINTER(CdArgs)],
    "mdbn_cd_forward": [_vp, _vp, C.POINTER(CdArgs)],
    "mdbn_cd_statistics": [_vp, _vp, C.POINTER(CdArgs), C.POINTER(UpdateArgs)],
    "mdbn_cd_train_step": [_vp, _vp, C.POINTER(CdArgs), C.POINTER(UpdateArgs)],
    "mdbn_free_energy": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i32, _vp,
                         _vp, _i64],
    "mdbn_round_flip": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "mdbn_pl_cost": [_vp, _vp, _vp, _vp, _i64, _i64, _vp],
    "mdbn_recon_cost": [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64],
    "mdbn_tanh": [_vp, _vp, _vp, _i64, _i64, _i64],
    "mdbn_count_nonfinite": [_vp, _vp, _vp, _i64, _vp],
    "mdbn_f32_to_bf16": [_vp, _vp, _vp, _vp, _i64],
    "mdbn_bf16_to_f32": [_vp, _vp, _vp, _vp, _i64],
    "mdbn_rng_uniform": [_vp, _vp, _vp, _i64, _i64, _i64, _rngp],
    "mdbn_rng_normal": [_vp, _vp, _vp, _i64, _i64, _i64, _rngp],
    "mdbn_philox_host": [_vp, _i64, _i64, _i64, _rngp],
}

_lib = None
_diagnostic = False


def use_diagnostic_library(path):
    """Experiments only (scripts/): load ``path`` -- a library built with diagnostic -D flags from the sources on disk --
    instead of the in-tree one.  Its embedded hash is not compared with the sources (the flags differ by design); the
    ctypes layouts are those of the current header, so it must be built from the CURRENT sources."""
    global LIB_PATH, _diagnostic, _lib
    if _lib is not None:
        raise MdbnError("a library is already loaded in this process")
    LIB_PATH, _diagnostic = path, True



def load():
    """dlopen the in-tree library; raises MdbnError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build
    if not _diagnostic and _build.is_stale():
        # missing, or built from other sources than the ones on disk (content hash, not mtime)
        try:
            _build.build_lib(force=True)
        except Exception as exc:
            raise MdbnError(
                "%s is missing or was built from different sources, and rebuilding it failed (%r). Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or python -m mdbn_amd.build). "
                "There is no CPU fallback." % (LIB_PATH, exc))
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.mdbn_version() != 2:
        raise MdbnError("libmdbn_hip.so version mismatch (the ctypes layouts here are those of MDBN_VERSION 2)")
    buf = C.create_string_buffer(80)
    lib.mdbn_source_hash(buf, 80)
    if not _diagnostic and buf.value.decode() != _build.source_hash():
        # the mapped code is not what the ctypes layouts above describe (e.g. the path was dlopen'ed before a
        # rebuild in this process): refuse to run kernels that would be attributed to the wrong sources
        raise MdbnError("%s is loaded with source hash %s but the sources on disk hash to %s; rebuild "
                        "(python -m mdbn_amd.build) and restart the process"
                        % (LIB_PATH, buf.value.decode()[:16], _build.source_hash()[:16]))
    _lib = lib
    return lib


def last_error():
    buf = C.create_string_buffer(512)
    load().mdbn_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise MdbnError("%s failed (%d): %s" % (what, rc, last_error()))
