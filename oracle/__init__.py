"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's hot-path algorithms, used as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
dcd_amd/ may import this package.
"""
