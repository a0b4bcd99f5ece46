"""Parity oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under myfm_amd/ imports it (tests/test_no_oracle_in_product.py enforces that).
"""
