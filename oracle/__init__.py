"""ORACLE — test infrastructure only (CPU restatement of the reference's hot path).

Importable ONLY from `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg.
PARITY UNPINNED: the reference ships no golden vectors and cannot run here (see model_ref.py header).
"""
