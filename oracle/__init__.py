"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference DGMR training-step arithmetic.

Nothing under ``skillful_nowcasting_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, as the checker.
"""
