"""TEST INFRASTRUCTURE: CPU oracle for the tick path (see oracle/vds_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product package ``vehicles_dispatch_simulator_amd`` never does.
"""
