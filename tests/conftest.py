import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions leave the measured tolerance margins behind (tests/_margins.py)."""
    try:
        import _margins
        _margins.dump(os.path.join(ROOT, "gpurun_out", "tolerance_margins.json"))
    except Exception:
        pass


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree libmdbn_hip.so (built on demand; hipcc cross-compiles without a GPU)."""
    from mdbn_amd.build import build_lib
    return build_lib()


@pytest.fixture(scope="session")
def hip_engine(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mdbn_amd
    eng = mdbn_amd.set_engine(mdbn_amd.HipEngine())
    eng.keep_f32 = True          # the parity tests inspect ph / nh / nv in the CD scratch (the product default skips those copies)
    eng.set_planes_min_work(0)   # the plane path at every whole-tile shape (the product default keeps small layers off it)
    return eng


@pytest.fixture()
def oracle_engine():
    """CPU checker engine (tests/_oracle_engine.py) installed as the default engine."""
    import mdbn_amd
    from _oracle_engine import OracleEngine
    import mdbn_amd.engine as E
    prev = E._default_engine
    eng = mdbn_amd.set_engine(OracleEngine())
    yield eng
    E._default_engine = prev
