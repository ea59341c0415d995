"""CPU oracle for the CD-k hot path of glgerard/MDBN (src/rbm.py, src/dbn.py).

TEST INFRASTRUCTURE ONLY.  Nothing under ``mdbn_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference has no tests, golden vectors or fixtures, and its
arithmetic lives in Theano (un-vendored, version not pinned, not installable
offline), so this restatement cannot be checked against reference outputs.  It is
pinned instead by algebraic known-answer tests (brute-force partition function,
finite differences, hand-computed update-rule cases) and by the numpy legacy
``RandomState`` known answers listed in SURVEY.md section 0.
"""
