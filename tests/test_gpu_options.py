"""Library options and timing records belong to their context (include/mdbn_hip.h, "Tuning knobs, PER CONTEXT"): two engines
of one process -- modality-parallel placement in-process, a test toggling a knob -- never see each other's settings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _kinds(eng, V, H, B, gauss=False):
    from mdbn_amd import RngAddr
    rs = np.random.RandomState(0)
    W = (rs.uniform(-0.1, 0.1, size=(V, H))).astype(np.float32)
    data = rs.normal(size=(B, V)).astype(np.float32) if gauss else (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
    dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, np.zeros(H, np.float32), np.zeros(V, np.float32), data)]
    eng.kernel_timing(True)
    try:
        eng.cd_step(dx, None, dW, dhb, dvb, gauss, 1, RngAddr(1, 0, 0, 0, 0))
        eng.synchronize()
        return [k for _, _, _, k in eng.kernel_timing_detail()]
    finally:
        eng.kernel_timing(False)


def test_two_contexts_keep_their_own_options(built_lib):
    import mdbn_amd
    a, b = mdbn_amd.HipEngine(), mdbn_amd.HipEngine()
    # knob 1: the one-launch step of LDS-resident layers off on `a` only
    a.set_option("small_fused", 0)
    ka, kb = _kinds(a, 100, 24, 512), _kinds(b, 100, 24, 512)
    assert len(ka) > 0 and any(k >= 1000 for k in ka), ka          # the multi-launch path: register-streaming GEMMs (family 1)
    assert kb == [], kb                                            # b: the one-launch path records no GEMM launch
    # knob 2: the plane path for a small whole-tile shape on `b` only (the default rule keeps it off planes)
    b.set_planes_min_work(0)
    ka, kb = _kinds(a, 1024, 512, 256, gauss=True), _kinds(b, 1024, 512, 256, gauss=True)
    assert ka and all(k < 2000 for k in ka), ka                    # a: f32-operand kernels (families 0 / 1)
    assert kb and all(k >= 2000 for k in kb), kb                   # b: bf16 plane kernels (family 2)
    # ... and the sizing calls answer per context
    assert b.plane_shape(256, 1024, 512, 1024, 512) and not a.plane_shape(256, 1024, 512, 1024, 512)
    # the timing records are per context too: enabling on one does not record the other's launches
    a.kernel_timing(True)
    try:
        _ = _kinds(b, 1024, 512, 256, gauss=True)
        a.synchronize()
        assert a.kernel_timing_detail() == []
    finally:
        a.kernel_timing(False)


@pytest.mark.parametrize("shape", [(100, 24, 512), (784, 500, 20), (300, 130, 100)], ids=["one_launch", "thin", "streaming"])
def test_a_whole_step_between_forward_and_statistics_ends_the_hand_over(hip_engine, shape):
    """mdbn_cd_forward leaves partials in the workspace all shapes share; a whole step in between overwrites them, so the
    statistics half must then be refused (MDBN_EINVAL) on every path instead of summing foreign partials."""
    from mdbn_amd import RngAddr, _lib
    eng = hip_engine
    V, H, B = shape
    rs = np.random.RandomState(1)
    W = rs.uniform(-0.1, 0.1, size=(V, H)).astype(np.float32)
    data = (rs.uniform(size=(B, V)) < 0.3).astype(np.float32)
    dW, dhb, dvb, dx = [eng.to_device(a) for a in (W, np.zeros(H, np.float32), np.zeros(V, np.float32), data)]
    token = eng.cd_forward(dx, None, dW, dhb, dvb, False, 1, RngAddr(1, 0, 0, 0, 0))
    eng.cd_step(dx, None, dW, dhb, dvb, False, 1, RngAddr(1, 0, 1, 0, 0), stats_slot=1)        # a whole step in between
    with pytest.raises(_lib.MdbnError, match="must follow mdbn_cd_forward"):
        eng.cd_statistics(token)
    # the regular order still works afterwards
    token = eng.cd_forward(dx, None, dW, dhb, dvb, False, 1, RngAddr(1, 0, 2, 0, 0))
    eng.cd_statistics(token)
    eng.synchronize()


def test_environment_knobs_reach_every_new_context(built_lib, monkeypatch):
    """MDBN_OPTIONS="name=value,..." (A/B runs of the bench scripts): applied by every HipEngine created afterwards, to its own
    context only; an unknown name fails as mdbn_set_option does."""
    import mdbn_amd
    from mdbn_amd import _lib
    plain = mdbn_amd.HipEngine()
    monkeypatch.setenv("MDBN_OPTIONS", "stream_x6=0, small_fused=0")
    tuned = mdbn_amd.HipEngine()
    monkeypatch.delenv("MDBN_OPTIONS")
    ka, kb = _kinds(plain, 1000, 300, 200), _kinds(tuned, 1000, 300, 200)
    assert ka and all(1100 <= k < 2000 for k in ka), ka            # the default: the streaming bf16x6 kernel
    assert kb and not any(1100 <= k < 2000 for k in kb), kb        # stream_x6 = 0 on this context: LDS-tiled / exact kernels
    monkeypatch.setenv("MDBN_OPTIONS", "no_such_knob=1")
    with pytest.raises(_lib.MdbnError):
        mdbn_amd.HipEngine()
