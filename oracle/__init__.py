"""CPU oracle of the steering-coefficient hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg import this package.  Parity status: "parity unpinned" (see bf_oracle.h).
"""
