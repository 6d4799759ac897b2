"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the RangeDet inference hot path used as the parity checker.  Nothing under
``rangedet_amd/`` may import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do.
"""
