"""CPU oracle for the VLSA language-guided patch-aggregation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vlsa_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker /
the timed CPU baseline -- never as the thing shipped.

Parity status: PINNED.  ``oracle/vlsa_oracle.py`` is checked (a) against the upstream reference's own
Python (imported from /root/reference in the build container, ``tests/test_oracle_vs_reference.py``) and
(b) against the committed golden vectors under ``tests/golden/*.npz`` that were produced by that import
(``tests/golden/make_golden.py``).  The reference ships no tests or golden vectors of its own
(SURVEY.md section 4), so those generated fixtures are the pin.
"""
