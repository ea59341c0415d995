"""ThreadSanitizer over the host threads of the row feeder: mdbn_amd/csrc/row_pool.h is plain C++, so
its spin / sleep hand-off is compiled alone with g++ -fsanitize=thread and driven by tests/tsan_row_pool.cc -- no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("spin_us", ["2000", "0"])
def test_row_pool_under_thread_sanitizer(tmp_path, spin_us):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "tsan_row_pool")
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
                            "-I", os.path.join(ROOT, "mdbn_amd", "csrc"), os.path.join(ROOT, "tests", "tsan_row_pool.cc"),
                            "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in (build.stderr or "").lower():
        pytest.skip("libtsan is not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, MDBN_FEED_SPIN_US=spin_us, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-4000:]
    assert run.returncode == 0, (run.returncode, run.stdout[-500:], run.stderr[-2000:])
    assert "row pool: 0 bad" in run.stdout
