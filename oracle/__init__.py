"""oracle/ — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's algorithm for the hot path).

Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it, as the checker / CPU baseline, never as the thing shipped or measured.
Parity status: PINNED — checked against golden vectors generated from the reference itself
(tests/golden/make_golden.py, run in the build container where /root/reference exists) and against the
reference tests' closed-form known answers (tests/common.py:29-44, tests/cpu/core/test_functional.py:15-36).
"""
