"""TEST INFRASTRUCTURE ONLY — CPU oracle for the hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``hydrium_amd/`` does (tests/test_layout.py enforces that).
"""
