"""CPU oracle for the TDVC P-frame hot path — TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
anything under this directory; the product (`tdvc_amd/`) never does.
"""
