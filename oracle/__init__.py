"""CPU oracle of the Point-DAE hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; point_dae_amd/ never does (see oracle/pdae_oracle.c header)."""
