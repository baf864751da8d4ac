#!/usr/bin/env python3
"""What do the ten Ruiz passes cost under load (three workgroups per CU)?  Two runs with exactly the same iteration work — 75
iterations, three termination checks, one factorisation (adaptive rho off, tolerances unreachable) — with 10 and with 0 scaling
passes; the difference is the loaded cost of the scaling phase.  Parameters only, product library."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_shapes as bs
from decentralized_ekf_mhe_amd import go1_params

for passes in (10, 0, 10, 0):
    bs.run(f"go1 75 forced iterations, scaling passes = {passes}", go1_params, 4096, 60, scaling_iters=passes, adapt_rho=0, max_qp_iter=75,
           abs_tol=1e-30, rel_tol=1e-30)
