"""CPU oracle for the CSPN hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (cspn_amd/) never does.
"""
from .oracle import cspn2d_oracle, cspn2d_gate_wb_oracle, cspn3d_oracle, guidance_head_oracle, guidance_head_backward_oracle, build, oracle_threads, set_oracle_threads  # noqa: F401
