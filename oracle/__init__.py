"""CPU oracle of the hot path -- TEST INFRASTRUCTURE (see fx_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
