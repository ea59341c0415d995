"""Tolerance bookkeeping of the GPU parity tests (TEST-ONLY).

``check(name, measured, tol)`` asserts ``measured <= tol`` and remembers the worst value seen under ``name``
together with the tolerance and SURVEY 8d's value for that quantity; at the end of a GPU session conftest.py
writes the table to gpurun_out/tolerance_margins.json (committed per round under profiles/ and quoted in
DESIGN.md section 4): with an unpinned oracle the tolerance table is the contract, so every entry carries the
margin that was actually measured on the box."""
import json
import os

_records = {}

# fp32 device vs float64 oracle on identical inputs and injected uniforms
SURVEY = {"prob": 2e-6, "nv_mean": 2e-6, "stats": 1e-5, "update": 1e-6, "free_energy": 1e-4, "drift100": 1e-4,
          "tie": 1e-6}


def check(name, measured, tol, kind=None, msg=None):
    measured, tol = float(measured), float(tol)
    r = _records.setdefault(name, {"worst": 0.0, "tol": tol, "survey": SURVEY.get(kind), "kind": kind, "n": 0})
    r["worst"] = max(r["worst"], measured)
    r["tol"] = max(r["tol"], tol)
    r["n"] += 1
    assert measured <= tol, "%s: %.3e > %.3e%s" % (name, measured, tol, (" (%s)" % msg) if msg else "")


def dump(path):
    if not _records:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = {}
    if os.path.exists(path):
        try:
            with open(path) as f:
                old = json.load(f)
        except Exception:
            old = {}
    for k, r in _records.items():
        if k in old:
            r["worst"] = max(r["worst"], old[k].get("worst", 0.0))
            r["n"] += old[k].get("n", 0)
        r["margin"] = (r["tol"] / r["worst"]) if r["worst"] > 0 else None
        old[k] = r
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)
