"""ORACLE package (test infrastructure, not product code).

ctypes bindings for oracle/liboracle.so (this repo's CPU restatement of the GraphChainer hot path) and
oracle/_ref/libref_units.so (the reference's own directly-compilable units). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .binding import Oracle, RefUnits, load_oracle_lib, build  # noqa: F401
