"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/scan_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
