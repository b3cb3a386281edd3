"""CPU oracle for the SuperSLAM deep-feature front-end hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``superslam_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / reported baseline, never as the product path.
"""
