"""HipEngine: the device side of the RBM/DBN classes.

Owns nothing but a context handle and scratch tensors; every method enqueues work on
torch's current HIP stream through the C-ABI (mdbn_amd/_lib.py) and returns device
tensors.  PyTorch is used only as the device-array container (allocation, views,
H2D/D2H copies, streams).  There is no CPU fallback: constructing the engine without
the built library or without a gfx950 GPU raises.

Matrix convention: a device matrix is a torch view [rows, cols] over storage
[rows, ld] with ld = padded_ld(cols) (a multiple of 4) and zero padding (``alloc_matrix``).
"""
import ctypes as C
import os as _os

import numpy
import torch

from . import _lib


def round_up4(n):
    return (int(n) + 3) & ~3


def padded_ld(cols):
    """Leading dimension of a device matrix (same policy as mdbn_padded_ld in the library):
    round_up(cols, 4).  (Padding power-of-two rows was measured and rejected, see mdbn_capi.hip.)"""
    return round_up4(cols)


_default_engine = None


def get_engine():
    """Process-wide default engine (cuda:current device); created on first use."""
    global _default_engine
    if _default_engine is None:
        _default_engine = HipEngine()
    return _default_engine


def set_engine(engine):
    """Install ``engine`` as the default (tests install a CPU checker here)."""
    global _default_engine
    _default_engine = engine
    return engine


class RngAddr(object):
    """Address of one random matrix: see csrc/philox.h."""
    __slots__ = ("seed", "stream_id", "step", "draw", "row_offset")

    def __init__(self, seed, stream_id, step, draw=0, row_offset=0):
        self.seed, self.stream_id, self.step = int(seed), int(stream_id), int(step)
        self.draw, self.row_offset = int(draw), int(row_offset)

    def c(self):
        return _lib.Rng(self.seed & 0xFFFFFFFFFFFFFFFF, self.stream_id & 0xFFFFFFFF,
                        self.step & 0xFFFFFFFF, self.draw & 0xFFFFFFFF, 0, self.row_offset)


class CDScratch(object):
    """Per-(B, V, H) buffers of one CD step; see mdbn_cd_args in include/mdbn_hip.h."""

    def __init__(self, engine, B, V, H, need_vs, ldv, ldh):
        self.B, self.V, self.H = B, V, H
        self.V2 = engine.alloc_matrix(2 * B, V, ldv)       # same ld as the data matrix
        self.P2 = engine.alloc_matrix(2 * B, H, ldh)       # same ld as W
        self.hs = engine.alloc_matrix(B, H, ldh)
        self.vs = engine.alloc_matrix(B, V, ldv) if need_vs else None
        self.trace_h = self.trace_v = None      # chain taps (HipEngine.trace_chain), [k+1, B, ldh] / [k, B, ldv]
        # bf16 plane scratch (mdbn_cd_args.planes) for shapes made of whole 128-row / 128-column tiles
        self.planes = None
        if engine.plane_shape(B, V, H, ldv, ldh):
            n = C.c_int64()
            _lib.check(engine.lib.mdbn_planes_bytes(B, ldv, ldh, C.byref(n)), "mdbn_planes_bytes")
            self.planes = torch.empty(n.value // 2, dtype=torch.int16, device=engine.device)
        # gather-ahead (mdbn_cd_args.next_indexes): second X2-plane buffer (made on first use), the buffer the current
        # step uses, and what the last step gathered ahead: (data pointer, data version, index tensor kept alive, buffer)
        self.planes_alt = None
        self.x_buffer = 0
        self.ahead = None

    def alt_planes(self, engine, ldv):
        if self.planes_alt is None:
            n = C.c_int64()
            _lib.check(engine.lib.mdbn_ahead_bytes_ctx(engine.ctx, self.B, self.V, self.H, ldv, self.P2.stride(0), C.byref(n)),
                       "mdbn_ahead_bytes_ctx")
            self.planes_alt = torch.empty(n.value // 2, dtype=torch.int16, device=engine.device)
        return self.planes_alt


class RowFeeder(object):
    """``mdbn_feeder_*`` (include/mdbn_hip.h): minibatches of a host-resident table, gathered by CPU threads into pinned
    staging and moved by one copy each on the feeder's copy stream, ``slots`` deep.  ``submit`` returns a ticket at once;
    ``acquire(ticket)`` (submission order) makes the current stream wait for that minibatch's copy and returns its device
    rows; ``release(ticket)`` after the work reading them has been enqueued."""

    def __init__(self, engine, host, cols, max_rows, slots, threads):
        self.engine, self.host, self.cols, self.max_rows = engine, host, int(cols), int(max_rows)
        self.bufs = [engine.alloc_matrix(self.max_rows, self.cols, host.stride(0)) for _ in range(slots)]
        ptrs = (C.c_void_p * slots)(*[b.data_ptr() for b in self.bufs])
        self.handle = C.c_void_p()
        _lib.check(engine.lib.mdbn_feeder_create(engine.ctx, C.c_void_p(host.data_ptr()), host.shape[0], self.cols,
                                                 host.stride(0), self.max_rows, slots, ptrs, self.bufs[0].stride(0),
                                                 int(threads), C.byref(self.handle)), "mdbn_feeder_create")

    def submit(self, indexes, n=None):
        """``indexes``: CPU int64 tensor (or None with ``n``: the table's first n rows)."""
        t = C.c_int64()
        if indexes is None:
            _lib.check(self.engine.lib.mdbn_feeder_submit(self.handle, None, int(n), C.byref(t)), "mdbn_feeder_submit")
        else:
            assert indexes.dtype == torch.int64 and indexes.device.type == "cpu" and indexes.is_contiguous()
            _lib.check(self.engine.lib.mdbn_feeder_submit(self.handle, C.c_void_p(indexes.data_ptr()), indexes.numel(),
                                                         C.byref(t)), "mdbn_feeder_submit")
        return t.value

    def acquire(self, ticket, rows):
        slot = C.c_int32()
        _lib.check(self.engine.lib.mdbn_feeder_acquire(self.handle, int(ticket), self.engine._stream(), C.byref(slot)),
                   "mdbn_feeder_acquire")
        return self.bufs[slot.value][:rows]

    def release(self, ticket):
        _lib.check(self.engine.lib.mdbn_feeder_release(self.handle, int(ticket), self.engine._stream()), "mdbn_feeder_release")

    def stats(self):
        """Host-side time per stage since the last call (resets): dict of counts and mean microseconds."""
        out = (C.c_double * 5)()
        _lib.check(self.engine.lib.mdbn_feeder_stats(self.handle, out), "mdbn_feeder_stats")
        return {"fed": int(out[0]), "gather_us": out[1], "copy_enqueue_us": out[2], "acquired": int(out[3]),
                "acquire_wait_us": out[4]}

    def cancel(self):
        _lib.check(self.engine.lib.mdbn_feeder_cancel(self.handle), "mdbn_feeder_cancel")

    def close(self):
        if getattr(self, "handle", None):
            self.engine.lib.mdbn_feeder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipEngine(object):
    name = "hip"

    def __init__(self, device=None):
        self.lib = _lib.load()                      # raises if the extension is not built
        if not torch.cuda.is_available():
            raise _lib.MdbnError("no HIP device visible: mdbn_amd has no CPU fallback")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.MdbnError("HipEngine needs a HIP device, got %r" % (device,))
        # 'cuda' without an index means the current device: fixed ONCE here (the raw-stream lookup and mdbn_ctx_create both
        # take the index; an engine must not follow later torch.cuda.set_device calls)
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        ctx = C.c_void_p()
        _lib.check(self.lib.mdbn_ctx_create(C.byref(ctx), self.device.index), "mdbn_ctx_create")
        self.ctx = ctx
        self._workspace = None
        self._options_epoch = 0         # bumped by set_option: argument structs cached by step functions are then stale
        self._w_serial = 0              # bumped by every library call that writes parameters (what a positive phase computed
                                        # ahead of its step was computed FROM: _ahead_valid)
        self._ws_need = {}              # (B, V, H) -> bytes the library asks for (options that change it clear this)
        self._stats = {}
        self._scratch = {}
        self._cost_ring = torch.zeros(1024, dtype=torch.float32, device=self.device)
        self._cost_slot = 0
        self.last_scratch = None        # CDScratch of the most recent CD step (inspection / chain taps)
        # MDBN_OPTIONS="name=value,name=value": library knobs of every context this process creates (A/B runs of the
        # bench scripts without editing them; unknown names fail as mdbn_set_option does)
        for item in filter(None, _os.environ.get("MDBN_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            self.set_option(name.strip(), int(value))

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.mdbn_ctx_destroy(self.ctx)
        except Exception:
            pass

    # ------------------------------------------------------------------ arrays
    def alloc_matrix(self, rows, cols, ld=None):
        ld = padded_ld(cols) if ld is None else int(ld)
        return torch.zeros((rows, ld), dtype=torch.float32, device=self.device)[:, :cols]

    def alloc_vector(self, n):
        return torch.zeros(int(n), dtype=torch.float32, device=self.device)

    def to_device(self, value):
        """numpy / tensor -> padded device matrix (2-D) or vector (1-D), float32."""
        if isinstance(value, torch.Tensor):
            t = value.to(device=self.device, dtype=torch.float32)
        else:
            t = torch.from_numpy(numpy.ascontiguousarray(value, dtype=numpy.float32)).to(self.device)
        if t.dim() == 2:
            if t.device == self.device and self.is_matrix(t) and t is value:
                return t
            out = self.alloc_matrix(t.shape[0], t.shape[1])
            out.copy_(t)
            return out
        return t.contiguous()

    @staticmethod
    def is_matrix(t):
        return (t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1
                and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0)

    def as_matrix(self, x):
        """Accept numpy arrays, tensors or SharedArray-likes; return a device matrix."""
        t = getattr(x, "tensor", x)
        if isinstance(t, torch.Tensor) and t.device == self.device and self.is_matrix(t):
            return t
        return self.to_device(t)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    trace_chain = False         # True: every CD step also records the samples its Gibbs chain feeds onward
    keep_f32 = False            # True: the plane path also stores the float32 copies of ph / nh / nv / samples in the
                                # CD scratch (CDScratch.P2, V2[B:], hs, vs) for inspection; nothing on the path reads them
    host_feed_threads = 8           # CPU threads of a RowFeeder's gather (85 us per 8.4-MB minibatch on the GPU box's host; 16: 67)
    host_gather_workgroups = 32     # workgroups x threads of the PCIe gather of a host-resident table (mdbn_gather_rows_host)
    host_gather_threads = 256
    check_indexes = False       # True: also range-check index lists that already live on the device (one sync)

    def index_tensor(self, indexes, n_rows=None):
        """Minibatch indices -> device int32/int64 tensor (no copy if already there).  With ``n_rows``
        host index lists are range-checked like numpy fancy indexing (IndexError; negative indices
        count from the end) -- the gather kernel itself clamps instead of faulting."""
        if isinstance(indexes, torch.Tensor):
            if indexes.dtype not in (torch.int32, torch.int64):
                indexes = indexes.to(torch.int64)
            if n_rows is not None and indexes.numel() and (self.check_indexes or indexes.device.type == "cpu"):
                lo, hi = int(indexes.min()), int(indexes.max())
                if lo < -n_rows or hi >= n_rows:
                    raise IndexError("minibatch index out of range for %d rows: [%d, %d]" % (n_rows, lo, hi))
            return indexes.to(self.device).contiguous()
        a = numpy.asarray(indexes)
        if a.dtype not in (numpy.int32, numpy.int64):
            a = a.astype(numpy.int64)
        if n_rows is not None and a.size and (a.min() < -n_rows or a.max() >= n_rows):
            raise IndexError("minibatch index out of range for %d rows: [%d, %d]" % (n_rows, a.min(), a.max()))
        return torch.from_numpy(numpy.ascontiguousarray(a)).to(self.device)

    # ------------------------------------------------------------------ scratch
    def _stream(self):
        # (torch.cuda.current_stream() builds a Stream object: 5 us per call -- a sixth of a small layer's step)
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))

    def workspace(self, B, V, H):
        """One shared workspace, grown to the largest requirement seen.  The requirement is NOT monotone in (B, V, H) (a
        smaller minibatch takes more split-K slabs), so every shape asks the library (cached per shape)."""
        need = self._ws_need.get((B, V, H))
        if need is None:
            n = C.c_int64()
            _lib.check(self.lib.mdbn_workspace_bytes_ctx(self.ctx, B, V, H, C.byref(n)), "mdbn_workspace_bytes_ctx")
            need = self._ws_need[(B, V, H)] = max(n.value, 8 << 20)
        if self._workspace is None or self._workspace.numel() * 4 < need + 256:
            self._workspace = torch.empty(need // 4 + 64, dtype=torch.float32, device=self.device)
        return self._workspace

    def new_stats_buffer(self, V, H, ldv=None, ldh=None):
        """A packed [S | s_h | s_v | cost] buffer owned by the caller (a step function whose statistics
        must survive until a deferred update reads them keeps its own, never the engine's shared one)."""
        ldv = padded_ld(V) if ldv is None else ldv
        ldh = padded_ld(H) if ldh is None else ldh
        n = C.c_int64()
        _lib.check(self.lib.mdbn_stats_floats(V, ldv, ldh, C.byref(n)), "mdbn_stats_floats")
        return torch.zeros(n.value, dtype=torch.float32, device=self.device)

    def stats_buffer(self, V, H, slot=0, ldv=None, ldh=None):
        """Engine-wide scratch statistics buffer (consumed before the call that filled it returns)."""
        ldv = padded_ld(V) if ldv is None else ldv
        ldh = padded_ld(H) if ldh is None else ldh
        key = (V, H, slot, ldv, ldh)
        if key not in self._stats:
            self._stats[key] = self.new_stats_buffer(V, H, ldv, ldh)
        return self._stats[key]

    def cd_scratch(self, B, V, H, need_vs, ldv=None, ldh=None):
        ldv = padded_ld(V) if ldv is None else ldv
        ldh = padded_ld(H) if ldh is None else ldh
        key = (B, V, H, bool(need_vs), ldv, ldh)
        if key not in self._scratch:
            if len(self._scratch) > 8:
                self._scratch.clear()
            self._scratch[key] = CDScratch(self, B, V, H, need_vs, ldv, ldh)
        return self._scratch[key]

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    # ------------------------------------------------------------------ bf16 planes of W
    # The plane path pays for itself on big layers only (measured, scripts/step_ab.py gemm_planes 0 1 with
    # MDBN_AB_SHAPE: 4096->1024 at B = 512 158.7 -> 150 us, 2048->1024 122.9 -> 118.9; but 1024->256 69.2 -> 77.3,
    # 1024->512 at B = 256 68.1 -> 74.9, 2048->512 at B = 128 72.0 -> 80.6, 1024->1024 at B = 1024 112.0 -> 114.0):
    # B * V * H >= planes_min_work and V * H >= 2^21.  Mirrors mdbn_set_option("planes_min_work"); tests set 0.
    planes_min_work = 1 << 30

    def weight_ld(self, V, H):
        """Leading dimension of a [V, H] weight matrix.  A big layer whose hidden width is not a multiple of 128 (the
        reference's 2048 -> 400 gene-expression layer, AMLsm2.py / MDBN.py presets) gets its rows padded to the next
        multiple: the plane path (bf16 planes + LDS-DMA GEMMs, whole 128-column tiles) then serves it on the padded
        width, the pad columns holding exact zeros -- 2048 -> 400 CD-5 at B = 512: 393 us on the exact-f32 tile kernels,
        which is what a ragged width falls back to.  None: the default (padded_ld)."""
        if V % 128 == 0 and H % 128 != 0 and H > 128:
            He = (H + 127) // 128 * 128
            if V * He >= self.weight_ld_min_elems:
                return He
        return None

    weight_ld_min_elems = int(_os.environ.get("MDBN_WEIGHT_LD_MIN", 1 << 21))      # smallest V * padded H that is padded

    def plane_shape(self, B, V, H, ldv, ldh):
        """Shapes the plane path of the library takes under THIS engine's options: asked of the library itself
        (mdbn_planes_eligible_ctx), so the host's buffers and the library's choice of path cannot disagree."""
        ok = C.c_int32()
        _lib.check(self.lib.mdbn_planes_eligible_ctx(self.ctx, B, V, H, ldv, ldh, C.byref(ok)), "mdbn_planes_eligible_ctx")
        return bool(ok.value)

    def set_planes_min_work(self, work):
        """Smallest B * V * H the plane path serves (0: every whole-tile shape).  Library options belong to the context:
        this changes the rule for this engine only."""
        self.planes_min_work = int(work)
        self.set_option("planes_min_work", int(work))

    def set_option(self, name, value):
        """Library tuning knob (mdbn_set_option), e.g. ``set_option('gemm_bk', 32)``; of this engine's context only."""
        _lib.check(self.lib.mdbn_set_option(self.ctx, name.encode(), int(value)), "mdbn_set_option")
        self._options_epoch += 1         # (cached argument structs of step functions: stale)
        self._scratch.clear()            # options decide which scratch a shape needs (planes, slabs)
        self._ws_need.clear()            # ... and how many split-K slabs its workspace holds

    def w_planes(self, W, create=False):
        """(planes, valid) for a weight matrix: the [3, V, ldh] bf16 planes the library keeps in step with W, and
        whether they hold the split of the CURRENT W.  A torch-side write to W (set_value, checkpoint load) bumps
        the tensor's version and invalidates them; the library's own updates rewrite them."""
        # kept ON the tensor object (the one a SharedArray holds for its lifetime), never keyed by address: a new
        # matrix that reuses a freed one's memory must not inherit its planes
        ent = getattr(W, "_mdbn_planes", None)
        if ent is None or tuple(ent[0].shape[1:]) != (W.shape[0], W.stride(0)) or ent[2] != W.data_ptr():
            if not create:
                return None, False
            ent = [torch.empty((3, W.shape[0], W.stride(0)), dtype=torch.int16, device=self.device), None, W.data_ptr()]
            W._mdbn_planes = ent
        return ent[0], ent[1] == W._version

    def _w_planes_written(self, W):
        ent = getattr(W, "_mdbn_planes", None)
        if ent is not None:
            ent[1] = W._version

    # ------------------------------------------------------------------ propagation
    def propup(self, v, W, hbias, rng=None, want_pre=True, want_mean=True, want_sample=True):
        """[pre, mean, sample] of rbm.py:187-213; entries not wanted are None."""
        v = self.as_matrix(v)
        B, V = v.shape
        H = W.shape[1]
        assert W.shape[0] == V, "visible size mismatch: %d vs %d" % (V, W.shape[0])
        ldh = W.stride(0)                               # the C-ABI writes outputs with W's ld
        pre = self.alloc_matrix(B, H, ldh) if want_pre else None
        mean = self.alloc_matrix(B, H, ldh) if want_mean else None
        sample = self.alloc_matrix(B, H, ldh) if want_sample else None
        if B == 0:
            return pre, mean, sample
        ws = self.workspace(min(B, 4096), V, H)
        r = rng.c() if rng is not None else None
        _lib.check(self.lib.mdbn_propup_sample(
            self.ctx, self._stream(), self._p(v), B, v.stride(0), self._p(W), V, H, W.stride(0),
            self._p(hbias), self._p(pre), self._p(mean), 1.0, self._p(sample),
            C.byref(r) if r is not None else None, self._p(ws), ws.numel() * 4), "mdbn_propup_sample")
        return pre, mean, sample

    def propdown(self, h, W, vbias, gauss=False, add_noise=False, rng=None, v0=None):
        """[pre, mean, sample] of rbm.py:215-240 (RBM) / rbm.py:647-660 (GRBM).
        With ``v0`` also returns the un-normalised reconstruction-cost sum (device scalar)."""
        h = self.as_matrix(h)
        B, H = h.shape
        V = W.shape[0]
        assert W.shape[1] == H, "hidden size mismatch"
        if h.stride(0) != W.stride(0):                  # the C-ABI reads h with W's ld
            h2 = self.alloc_matrix(B, H, W.stride(0))
            h2.copy_(h)
            h = h2
        mean = self.alloc_matrix(B, V)
        sample = self.alloc_matrix(B, V)
        pre = mean if gauss else self.alloc_matrix(B, V)     # GRBM: "pre" is the mean (rbm.py:660)
        if B == 0:
            return (pre, mean, sample) if v0 is None else (pre, mean, sample, torch.zeros((), device=self.device))
        ws = self.workspace(min(B, 4096), V, H)
        r = rng.c() if rng is not None else None
        cost = None
        if v0 is not None:
            v0 = self.as_matrix(v0)
            if v0.stride(0) != mean.stride(0):          # target is read with the output's ld
                t = self.alloc_matrix(B, V, mean.stride(0))
                t.copy_(v0)
                v0 = t
            cost = torch.zeros(4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mdbn_propdown_sample(
            self.ctx, self._stream(), self._p(h), B, h.stride(0), self._p(W), V, H, mean.stride(0),
            self._p(vbias), int(bool(gauss)), int(bool(add_noise)),
            None if gauss else self._p(pre), self._p(mean), self._p(sample),
            C.byref(r) if r is not None else None, self._p(v0), self._p(cost),
            self._p(ws), ws.numel() * 4), "mdbn_propdown_sample")
        if cost is not None:
            return pre, mean, sample, cost[0]
        return pre, mean, sample

    def gibbs_chain(self, v, W, hbias, vbias, gauss, n_steps, rng, add_noise=False, want_pre=True):
        """n_steps of gibbs_vhv from the visible state ``v`` in ONE library call (mdbn_gibbs_chain): returns
        [pre_h, h_mean, h_sample, pre_v, v_mean, v_sample] of the last step; ``v`` itself is not modified."""
        v = self.as_matrix(v)
        B, V = v.shape
        H = W.shape[1]
        ldh = W.stride(0)
        state = self.alloc_matrix(B, V, v.stride(0))
        state.copy_(v)
        h_mean, h_sample = self.alloc_matrix(B, H, ldh), self.alloc_matrix(B, H, ldh)
        v_mean = self.alloc_matrix(B, V, v.stride(0))
        pre_h = self.alloc_matrix(B, H, ldh) if want_pre else None
        pre_v = self.alloc_matrix(B, V, v.stride(0)) if want_pre else None
        if B == 0:
            return [pre_h, h_mean, h_sample, pre_v, v_mean, state]
        ws = self.workspace(min(B, 4096), V, H)
        r = rng.c()
        _lib.check(self.lib.mdbn_gibbs_chain(
            self.ctx, self._stream(), self._p(state), B, state.stride(0), self._p(W), V, H, ldh, self._p(hbias),
            self._p(vbias), int(bool(gauss)), int(bool(add_noise)), int(n_steps), self._p(pre_h), self._p(h_mean),
            self._p(h_sample), self._p(pre_v), self._p(v_mean), C.byref(r), self._p(ws), ws.numel() * 4),
            "mdbn_gibbs_chain")
        return [pre_h, h_mean, h_sample, pre_v, v_mean, state]

    def free_energy(self, x, W, hbias, vbias, gauss):
        x = self.as_matrix(x)
        N, V = x.shape
        H = W.shape[1]
        out = self.alloc_vector(N)
        if N == 0:
            return out
        ws = self.workspace(min(N, 4096), V, H)
        _lib.check(self.lib.mdbn_free_energy(
            self.ctx, self._stream(), self._p(x), N, x.stride(0), self._p(W), V, H, W.stride(0),
            self._p(hbias), self._p(vbias), int(bool(gauss)), self._p(out),
            self._p(ws), ws.numel() * 4), "mdbn_free_energy")
        return out

    def gather_rows(self, src, indexes):
        src = self.as_matrix(src)
        idx = self.index_tensor(indexes, src.shape[0])
        out = self.alloc_matrix(idx.numel(), src.shape[1])
        if idx.numel() == 0:
            return out
        _lib.check(self.lib.mdbn_gather_rows(
            self.ctx, self._stream(), self._p(src), src.shape[0], src.shape[1], src.stride(0),
            self._p(idx), int(idx.dtype == torch.int64), idx.numel(), self._p(out), out.stride(0)),
            "mdbn_gather_rows")
        return out

    def gather_host_rows(self, host, cols, indexes, out=None):
        """``host[indexes]`` for a PINNED host matrix [n, ld] (shared.HostTable): the gather kernel reads the rows through
        their device-accessible host address, over PCIe, on the current stream.  ``out``: a device matrix to fill."""
        if not host.is_pinned():
            raise _lib.MdbnError("gather_host_rows needs pinned host memory (torch.Tensor.pin_memory)")
        idx = self.index_tensor(indexes, host.shape[0])
        if out is None:
            out = self.alloc_matrix(idx.numel(), cols, host.stride(0))
        assert out.shape[0] == idx.numel() and out.shape[1] == cols
        if idx.numel():
            _lib.check(self.lib.mdbn_gather_rows_host(
                self.ctx, self._stream(), C.c_void_p(host.data_ptr()), host.shape[0], cols, host.stride(0),
                self._p(idx), int(idx.dtype == torch.int64), idx.numel(), self._p(out), out.stride(0),
                int(self.host_gather_workgroups), int(self.host_gather_threads)), "mdbn_gather_rows_host")
        self._keep_alive = (host, idx)           # until the next call: the kernel reads them asynchronously
        return out

    def row_feeder(self, host, cols, max_rows, slots=3, threads=None):
        """A ``RowFeeder`` over the host matrix ``host`` [n, ld] (``mdbn_feeder_*``: CPU gather threads + one SDMA copy per
        minibatch on the feeder's own stream)."""
        return RowFeeder(self, host, cols, max_rows, slots, self.host_feed_threads if threads is None else threads)

    # ------------------------------------------------------------------ CD-k
    @staticmethod
    def _same_index_tensor(a, b):
        return (a is b) or (a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.numel() == b.numel()
                            and a.dtype == b.dtype)

    def _ahead_valid(self, ahead, data, idx, W, thin):
        """Was THIS minibatch prepared by the previous step (CDScratch.ahead)?  Same matrix, unchanged since, same index
        list; thin-batch path, whose record also covers x W: the same parameters, written by nobody since that step
        (torch-side writes move W._version, library-side ones _w_serial) under the same options."""
        if ahead is None or ahead[0].data_ptr() != data.data_ptr() or ahead[1] != data._version or \
                not self._same_index_tensor(ahead[2], idx) or ahead[4] != idx._version:
            return False
        return not thin or (len(ahead) == 9 and ahead[5:] == (W.data_ptr(), W._version, self._w_serial, self._options_epoch))

    def _ahead_record(self, announce, W):
        return announce + (W.data_ptr(), W._version, self._w_serial, self._options_epoch)

    def _cd_args(self, data, indexes, W, hbias, vbias, gauss, k, rng, persistent, add_noise, stats_slot,
                 sample_stats=False, stats=None, comm_cus=0, next_indexes=None):
        data = self.as_matrix(data)
        V, H = W.shape
        assert data.shape[1] == V, "data has %d columns, RBM has %d visibles" % (data.shape[1], V)
        idx = self.index_tensor(indexes, data.shape[0]) if indexes is not None else None
        B = idx.numel() if idx is not None else data.shape[0]
        ldv, ldh = data.stride(0), W.stride(0)
        sc = self.cd_scratch(B, V, H, not gauss, ldv, ldh)
        self.last_scratch = sc
        if stats is None:
            stats = self.stats_buffer(V, H, stats_slot, ldv, ldh)
        if persistent is not None and persistent.stride(0) != ldh:
            raise _lib.MdbnError("persistent chain must share W's leading dimension")
        ws = self.workspace(B, V, H)
        a = _lib.CdArgs()
        a.data, a.n_data = data.data_ptr(), data.shape[0]
        a.indexes = idx.data_ptr() if idx is not None else None
        a.index_is_64 = int(idx is not None and idx.dtype == torch.int64)
        a.gauss, a.add_noise, a.k = int(bool(gauss)), int(bool(add_noise)), int(k)
        a.sample_stats = int(bool(sample_stats))
        a.keep_f32 = int(bool(self.keep_f32))
        a.B, a.V, a.H = B, V, H
        a.ldv, a.ldh = ldv, ldh
        a.W, a.hbias, a.vbias = W.data_ptr(), hbias.data_ptr(), vbias.data_ptr()
        a.persistent = persistent.data_ptr() if persistent is not None else None
        a.V2, a.P2, a.hs = sc.V2.data_ptr(), sc.P2.data_ptr(), sc.hs.data_ptr()
        a.vs = sc.vs.data_ptr() if sc.vs is not None else None
        a.stats = stats.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        a.rng = rng.c()
        # W's planes travel with W whenever they exist (the update that follows rewrites them, and a step that finds them
        # stale re-splits them, whichever path it takes); the plane scratch only for shapes the library serves on planes
        wp, valid = self.w_planes(W, create=sc.planes is not None)
        if wp is not None:
            a.W_planes, a.W_planes_valid = wp.data_ptr(), int(valid)
        if sc.planes is not None:
            a.planes, a.planes_bytes = sc.planes.data_ptr(), sc.planes.numel() * 2
        a.comm_cus = int(comm_cus)
        # gather-ahead: was this minibatch gathered by the previous step's statistics kernel?  (same matrix, unchanged since,
        # same index list: the announced tensor has been kept alive, so an equal address means the same list)
        ahead, sc.ahead = sc.ahead, None
        keep = [data, idx, ws]
        # ... on the plane path; on the thin-batch path (B <= 32, ldh <= 512: the library decides and reports) the previous
        # step's update kernel gathered the rows AND left the partials of x W, so the parameters must be the ones it wrote
        thin = sc.planes is None and B <= 32 and ldh <= 512 and persistent is None and not sample_stats
        if (sc.planes is not None or thin) and idx is not None and not self.keep_f32 and not self.trace_chain:
            # (the announcing step keeps the matrix and the index tensor alive, so an equal address is the same object)
            # ... and unchanged: an index buffer refilled IN PLACE between the announcing call and this one has another
            # version counter, and the rows are gathered afresh
            if self._ahead_valid(ahead, data, idx, W, thin):
                sc.x_buffer = ahead[3]
                a.v0_ready = 1
            if sc.planes_alt is not None or next_indexes is not None:
                a.planes_alt = sc.alt_planes(self, ldv).data_ptr()
                a.x_buffer = sc.x_buffer
            if next_indexes is not None:
                nxt = self.index_tensor(next_indexes, data.shape[0])
                if nxt.numel() == B and nxt.dtype == idx.dtype:
                    a.next_indexes = nxt.data_ptr()
                    self._ahead_flag = C.c_int32(0)
                    a.ahead_done = C.pointer(self._ahead_flag)
                    keep.append(nxt)
                    sc._announce = (data, data._version, nxt, 0 if thin else 1 - sc.x_buffer, nxt._version)
        if self.trace_chain:
            if sc.trace_h is None or sc.trace_h.shape[0] != k + 1:
                sc.trace_h = torch.zeros((k + 1, B, ldh), dtype=torch.float32, device=self.device)
                sc.trace_v = None if gauss else torch.zeros((k, B, ldv), dtype=torch.float32, device=self.device)
            a.trace_h = sc.trace_h.data_ptr()
            a.trace_v = sc.trace_v.data_ptr() if sc.trace_v is not None else None
        return a, stats, sc, keep                   # keep the tensors alive until enqueued

    def cd_step(self, data, indexes, W, hbias, vbias, gauss, k, rng, persistent=None, add_noise=False,
                stats_slot=0, sample_stats=False, stats=None, comm_cus=0):
        """gather + positive phase + k Gibbs steps + statistics (rbm.py:303-345,374).
        Returns (stats, scratch): the packed [S | s_h | s_v | cost_sum] buffer and the
        CDScratch holding ph_mean / nv_mean / nh_mean for inspection.  ``comm_cus`` > 0 (data-parallel
        mode): CUs left to the collective that runs beside this step (mdbn_cd_args.comm_cus)."""
        a, stats, sc, _keep = self._cd_args(data, indexes, W, hbias, vbias, gauss, k, rng, persistent,
                                            add_noise, stats_slot, sample_stats, stats, comm_cus)
        _lib.check(self.lib.mdbn_cd_step(self.ctx, self._stream(), C.byref(a)), "mdbn_cd_step")
        if a.W_planes:
            self._w_planes_written(W)            # (split on entry if they were stale)
        return stats, sc

    def cd_forward(self, data, indexes, W, hbias, vbias, gauss, k, rng, add_noise=False, sample_stats=False, stats=None,
                   comm_cus=0, next_indexes=None):
        """The first half of ``cd_step`` (mdbn_cd_forward): gather, positive phase and the Gibbs chain -- everything that
        reads the parameters.  Returns the token ``cd_statistics`` completes the step with."""
        a, stats, sc, keep = self._cd_args(data, indexes, W, hbias, vbias, gauss, k, rng, None, add_noise, 0,
                                           sample_stats, stats, comm_cus, next_indexes=next_indexes)
        sc._announce, announce = None, getattr(sc, "_announce", None)
        keep.append(announce)
        _lib.check(self.lib.mdbn_cd_forward(self.ctx, self._stream(), C.byref(a)), "mdbn_cd_forward")
        if a.W_planes:
            self._w_planes_written(W)
            a.W_planes_valid = 1
        return (a, stats, sc, keep, W)

    def cd_statistics(self, token, deferred=None):
        """The second half (mdbn_cd_statistics): bias statistics + statistics GEMM of the step ``cd_forward`` began.
        ``deferred``: the arguments of ``apply_update(..., phase=3)`` for the PREVIOUS step (whose all-reduced statistics
        the current stream has just been made to wait for) -- the library applies that update inside the statistics GEMM
        when it can, as its own launch otherwise; returns (stats, scratch, cost of the deferred update or None)."""
        a, stats, sc, keep, W = token
        u, cost = (None, None)
        if deferred is not None:
            u, cost = self._update_args(*deferred)
        _lib.check(self.lib.mdbn_cd_statistics(self.ctx, self._stream(), C.byref(a), C.byref(u) if u is not None else None),
                   "mdbn_cd_statistics")
        if u is not None:
            self._w_planes_written(W)
            self._w_serial += 1
        announce = keep[-1]
        if announce is not None and a.ahead_done and self._ahead_flag.value:
            sc.ahead = self._ahead_record(announce, W)              # the next call finds its rows in the other X2 buffer
        return stats, sc, cost

    def _update_args(self, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr, lambda_1,
                     lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale, phase, ldv):
        V, H = W.shape
        # Step costs are 0-d views into a block of 1024 floats.  A block is never reused: when it is full a fresh one is
        # allocated (4 KB per 1024 steps) and the old one lives as long as a caller still holds one of its views, so a
        # cost kept unread for any number of steps can never show another step's value.
        cost = self._next_cost_slot()
        u = _lib.UpdateArgs()
        u.W, u.W_speed = W.data_ptr(), W_speed.data_ptr()
        u.W0 = W0.data_ptr() if W0 is not None else None
        u.hbias, u.hbias_speed = hbias.data_ptr(), hbias_speed.data_ptr()
        u.vbias, u.vbias_speed = vbias.data_ptr(), vbias_speed.data_ptr()
        u.V, u.H, u.ldv, u.ldh = V, H, (padded_ld(V) if ldv is None else ldv), W.stride(0)
        u.stats = stats.data_ptr()
        u.lr, u.lambda_1, u.lambda_2 = float(lr), float(lambda_1), float(lambda_2)
        u.weightcost, u.momentum = float(weightcost), float(momentum)
        u.batch_size, u.n_rows, u.cost_scale = float(batch_size), float(n_rows), float(cost_scale)
        u.cost_out = cost.data_ptr()
        u.phase = int(phase)
        wp, _ = self.w_planes(W)
        u.W_planes = wp.data_ptr() if wp is not None else None    # kept in step with W by every update that writes W
        return u, cost

    def _next_cost_slot(self):
        # Step costs are 0-d views into a block of 1024 floats.  A block is never reused: when it is full a fresh one is
        # allocated (4 KB per 1024 steps) and the old one lives as long as a caller still holds one of its views.
        if self._cost_slot >= self._cost_ring.numel():
            self._cost_ring = torch.zeros(1024, dtype=torch.float32, device=self.device)
            self._cost_slot = 0
        slot = self._cost_slot
        self._cost_slot = slot + 1
        return self._cost_ring[slot]

    def cd_train_step_cached(self, cache, data, idx, rng_step, lr, momentum, next_indexes=None):
        """The step of ``cd_train_step`` through argument structs a step function keeps from its previous call
        (``cache`` = what ``cd_train_step(..., cache_out=)`` left): for small layers, whose step is two short launches, the
        host side -- building two ctypes structs of ~70 fields, resolving scratch, statistics and workspace buffers -- was
        32 us per call, more than the GPU's time.  Only what changes from call to call is set: the index list, the Philox
        step, lr / momentum and the cost slot.  Returns None when the cache does not apply (the caller takes the full path)."""
        a, u, a_ref, u_ref, key, sc, keep = cache[:7]
        W, W_speed, W0 = keep[2], keep[3], keep[4]
        # the structs were captured for a W WITHOUT bf16 planes: if another consumer of the same W has created some since
        # (another batch size of the same RBM, planes_min_work = 0), these steps must not go on updating W behind them
        if getattr(W, "_mdbn_planes", None) is not None:
            return None
        if key != (data.data_ptr(), data.shape[0], data.stride(0), idx.dtype, idx.numel(),
                   self._workspace.data_ptr() if self._workspace is not None else 0, self.keep_f32, self.trace_chain,
                   self._options_epoch, a.rng.seed, a.rng.stream_id, torch._C._cuda_getCurrentRawStream(self.device.index),
                   W.data_ptr(), W_speed.data_ptr(), W0.data_ptr() if W0 is not None else 0):
            return None
        a.indexes = idx.data_ptr()
        a.rng.step = rng_step & 0xFFFFFFFF
        u.lr, u.momentum = float(lr), float(momentum)
        cost = self._next_cost_slot()
        u.cost_out = cost.data_ptr()
        self.last_scratch = sc
        # thin-batch path: positive phase prepared by the previous step / prepare the next one's (see _cd_args)
        ahead, sc.ahead = sc.ahead, None
        announce = None
        if a.planes_alt:
            a.v0_ready = int(self._ahead_valid(ahead, data, idx, W, True))
            a.next_indexes = None
            if next_indexes is not None:
                nxt = self.index_tensor(next_indexes, data.shape[0])
                if nxt.numel() == idx.numel() and nxt.dtype == idx.dtype:
                    a.next_indexes = nxt.data_ptr()
                    announce = (data, data._version, nxt, 0, nxt._version)
        _lib.check(self.lib.mdbn_cd_train_step(self.ctx, self._stream(), a_ref, u_ref), "mdbn_cd_train_step")
        self._w_serial += 1
        if announce is not None and a.ahead_done and a.ahead_done[0]:
            sc.ahead = self._ahead_record(announce, W)
        return cost

    def cd_train_step(self, data, indexes, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, gauss, k,
                      rng, lr, lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale,
                      sample_stats=False, next_indexes=None, cache_out=None):
        """The whole single-device step function (mdbn_cd_train_step): cd_step + update, with the
        finalize / parameter half of the update overlapped under the statistics GEMM.  Returns the
        monitoring cost (0-d device tensor).  ``cache_out``: a list that receives the argument structs of this call when
        the next call of the same step function may reuse them (``cd_train_step_cached``): index list given, no plane
        buffers (the plane path's arguments change from step to step: gather-ahead), no chain taps."""
        a, stats, sc, _keep = self._cd_args(data, indexes, W, hbias, vbias, gauss, k, rng, None, False, 0,
                                            sample_stats, next_indexes=next_indexes)
        sc._announce, announce = None, getattr(sc, "_announce", None)
        u, cost = self._update_args(W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr,
                                    lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale,
                                    0, a.ldv)
        _lib.check(self.lib.mdbn_cd_train_step(self.ctx, self._stream(), C.byref(a), C.byref(u)),
                   "mdbn_cd_train_step")
        self._w_planes_written(W)
        self._w_serial += 1
        if announce is not None and a.ahead_done and self._ahead_flag.value:
            sc.ahead = self._ahead_record(announce, W)              # the next call finds its rows ready
        if cache_out is not None:
            del cache_out[:]
            idx = _keep[1]
            if idx is not None and sc.planes is None and not a.W_planes and not self.trace_chain and not sample_stats:
                dm = _keep[0]
                if a.B <= 32 and a.ldh <= 512 and not self.keep_f32:
                    # thin-batch candidates: the cached calls hand the next minibatch to the update kernel (their own flag)
                    flag = C.c_int32(0)
                    a.planes_alt, a.x_buffer, a.ahead_done = sc.alt_planes(self, a.ldv).data_ptr(), 0, C.pointer(flag)
                key = (dm.data_ptr(), dm.shape[0], dm.stride(0), idx.dtype, idx.numel(), self._workspace.data_ptr(), self.keep_f32,
                       self.trace_chain, self._options_epoch, a.rng.seed, a.rng.stream_id,
                       torch._C._cuda_getCurrentRawStream(self.device.index), W.data_ptr(), W_speed.data_ptr(),
                       W0.data_ptr() if W0 is not None else 0)
                cache_out.extend([a, u, C.byref(a), C.byref(u), key, sc, (dm, stats, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed)])
        return cost

    def apply_update(self, W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats,
                     lr, lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale, phase=0,
                     ldv=None):
        """rbm.py:347-365; returns the monitoring cost as a 0-d device tensor.
        phase: 0 = whole rule, 1 = speeds (+cost) only, 2 = parameters only (mdbn_update_args)."""
        u, cost = self._update_args(W, W_speed, W0, hbias, hbias_speed, vbias, vbias_speed, stats, lr,
                                    lambda_1, lambda_2, weightcost, momentum, batch_size, n_rows, cost_scale,
                                    phase, ldv)
        _lib.check(self.lib.mdbn_apply_update(self.ctx, self._stream(), C.byref(u)), "mdbn_apply_update")
        self._w_serial += 1
        if phase != 1:
            self._w_planes_written(W)
        return cost

    # ------------------------------------------------------------------ monitoring helpers (all HIP)
    def round_flip(self, x, flip_col=-1):
        """tensor.round(x) with column ``flip_col`` replaced by 1 - round(x) (rbm.py:428-436)."""
        x = self.as_matrix(x)
        out = self.alloc_matrix(x.shape[0], x.shape[1], x.stride(0))
        if x.shape[0]:
            _lib.check(self.lib.mdbn_round_flip(self.ctx, self._stream(), self._p(x), x.shape[0], x.shape[1],
                                                x.stride(0), int(flip_col), self._p(out)), "mdbn_round_flip")
        return out

    def pl_cost(self, fe, fe_flip, n_visible):
        """-mean(n_visible * softplus(fe - fe_flip)) (rbm.py:442) as a 0-d device tensor."""
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mdbn_pl_cost(self.ctx, self._stream(), self._p(fe), self._p(fe_flip), fe.numel(),
                                         int(n_visible), self._p(out)), "mdbn_pl_cost")
        return out[0]

    def recon_cost(self, pre, target, gauss):
        """get_reconstruction_cost on given arrays (rbm.py:449-482, :690-699): 0-d device tensor."""
        pre, target = self.as_matrix(pre), self.as_matrix(target)
        assert pre.shape == target.shape
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        ws = self.workspace(1, 1, 1)
        _lib.check(self.lib.mdbn_recon_cost(self.ctx, self._stream(), self._p(pre), pre.stride(0), self._p(target),
                                            target.stride(0), pre.shape[0], pre.shape[1], int(bool(gauss)),
                                            self._p(out), self._p(ws), ws.numel() * 4), "mdbn_recon_cost")
        return out[0]

    def tanh_(self, x):
        """In-place tanh on a device matrix (HiddenLayer's default activation, mlp.py:36-110)."""
        x = self.as_matrix(x)
        _lib.check(self.lib.mdbn_tanh(self.ctx, self._stream(), self._p(x), x.shape[0], x.shape[1], x.stride(0)),
                   "mdbn_tanh")
        return x

    def count_nonfinite(self, *tensors):
        """Number of NaN / Inf values in the given device tensors (synchronises): the check the
        reference's commented-out NanGuardMode would make (rbm.py:542-543, dbn.py:311)."""
        count = torch.zeros(1, dtype=torch.int32, device=self.device)
        for t in tensors:
            base = t._base if t._base is not None else t           # a padded matrix: scan the whole storage
            _lib.check(self.lib.mdbn_count_nonfinite(self.ctx, self._stream(), self._p(base), base.numel(),
                                                     self._p(count)), "mdbn_count_nonfinite")
        return int(count.item())

    # ------------------------------------------------------------------ RNG (tests / utilities)
    def rng_uniform(self, rows, cols, rng, normal=False):
        out = self.alloc_matrix(rows, cols)
        r = rng.c()
        fn = self.lib.mdbn_rng_normal if normal else self.lib.mdbn_rng_uniform
        _lib.check(fn(self.ctx, self._stream(), self._p(out), rows, cols, out.stride(0), C.byref(r)),
                   "mdbn_rng_*")
        return out

    def kernel_timing(self, enable):
        """Bracket every GEMM launch with HIP events (measurement only; bench.py)."""
        _lib.check(self.lib.mdbn_kernel_timing(self.ctx, int(bool(enable))), "mdbn_kernel_timing")

    def kernel_timing_read(self):
        n, ms = C.c_int64(), C.c_double()
        _lib.check(self.lib.mdbn_kernel_timing_read(self.ctx, C.byref(n), C.byref(ms)),
                   "mdbn_kernel_timing_read")
        return n.value, ms.value

    def kernel_timing_detail(self, cap=8192):
        """Per recorded GEMM launch: (ms, algorithmic FLOPs, FLOPs issued on its matrix pipe, kind)."""
        ms, alg, pipe = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
        kind, n = (C.c_int32 * cap)(), C.c_int64()
        _lib.check(self.lib.mdbn_kernel_timing_detail(self.ctx, cap, ms, alg, pipe, kind, C.byref(n)),
                   "mdbn_kernel_timing_detail")
        m = min(cap, n.value)
        return [(ms[i], alg[i], pipe[i], kind[i]) for i in range(m)]

    # ------------------------------------------------------------------ bfloat16 wire format (data-parallel reporting mode)
    def narrow_bf16(self, src, out=None):
        """bfloat16 copy of a float32 buffer (round to nearest even; mdbn_f32_to_bf16): the wire form of the packed statistics
        under MDBN_WIRE_BF16=1.  ``out``: a bfloat16 tensor of the same size to fill (kept by the caller across steps)."""
        if out is None or out.numel() != src.numel():
            out = torch.empty(src.numel(), dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.mdbn_f32_to_bf16(self.ctx, self._stream(), self._p(src), C.c_void_p(out.data_ptr()), src.numel()),
                   "mdbn_f32_to_bf16")
        return out

    def widen_bf16(self, wire, dst):
        """dst (float32) = wire (bfloat16), exactly (mdbn_bf16_to_f32)."""
        _lib.check(self.lib.mdbn_bf16_to_f32(self.ctx, self._stream(), C.c_void_p(wire.data_ptr()), self._p(dst), dst.numel()),
                   "mdbn_bf16_to_f32")
        return dst

    def cost_values(self, costs):
        """Python floats of a list of 0-d step costs: ONE device-to-host copy per 1024-cost ring block (costs are views into
        such blocks, `_next_cost_slot`; a block is never reused), no device arithmetic.  Synchronises."""
        out, blocks = [0.0] * len(costs), {}
        for i, c in enumerate(costs):
            base = c._base if getattr(c, "_base", None) is not None else c
            blocks.setdefault(id(base), (base, []))[1].append((i, c.storage_offset() - base.storage_offset()))
        for base, items in blocks.values():
            host = base.detach().reshape(-1).cpu().numpy()
            for i, off in items:
                out[i] = float(host[off])
        return out

    def synchronize(self):
        torch.cuda.synchronize(self.device)
