"""CPU oracle for the optical-flow -> ego-velocity hot path.

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg.  The product package never imports this (tests/test_no_oracle_in_product.py
greps for it).
"""
