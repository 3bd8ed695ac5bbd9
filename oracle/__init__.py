"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (and, under ``_ref/``, the reference's own vendored
OSQP compiled where it lies).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import, link or execute anything in here, and only as the checker -- never
as the thing measured or shipped.  The product path (``rl-mpc-locomotion_amd/``) never touches it.
"""
