"""CPU oracle for the COMBAT alternated-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``combat_amd/`` or the entry scripts may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / the timed CPU baseline.
"""
