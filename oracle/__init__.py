"""CPU oracle for the SemiReward hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.  The product path (``semireward_amd``)
never imports it and raises if ``libsrhip.so`` is missing.

It is an independent restatement (torch-CPU fp32 for floating point, numpy for the
integer/byte arithmetic) of the reference's algorithm; every function cites the
reference ``file:line`` it follows.  Parity pinning: the reference ships no tests or
golden vectors, so the oracle is pinned against outputs of the
reference itself, imported headless in the build container by ``oracle/gen_golden.py``
and committed as fixtures under ``tests/golden/`` (``tests/test_oracle_golden.py``).
"""
