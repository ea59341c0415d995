"""bench.py's watchdog never reports a stall as a clean run: the parked line is printed marked provisional,
with the stalled phase, and the process leaves with a non-zero code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_watchdog_prints_the_parked_line_and_exits_nonzero():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "dog = bench.Watchdog(0)\n"
            "dog.park({'metric': 'm', 'value': 1.0})\n"
            "dog.arm(0.2, 'a phase that hangs')\n"
            "time.sleep(30)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert p.returncode == 3, (p.returncode, p.stderr)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["provisional"] is True and out["watchdog_phase"] == "a phase that hangs" and out["value"] == 1.0
    assert "stalled" in p.stderr


def test_watchdog_finish_prints_once_and_exits_zero():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "dog = bench.Watchdog(0)\n"
            "dog.park({'value': 1.0}); dog.arm(0.2, 'x'); dog.finish({'value': 2.0}); time.sleep(1.5)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines == [json.dumps({"value": 2.0})]
