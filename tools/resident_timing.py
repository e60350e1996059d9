"""Resident (one launch) against multi-kernel GCR on small Poisson systems: us per iteration (the solver object is built
once; each timed solve uploads nothing)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import problems

mg.init(0)
for n, restart in ((64, 10), (64, 5), (48, 10), (32, 10), (16, 10)):
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    A = mg.Sparse(N, ncol, rowptr, col, val)
    dims = (n, n, n)
    rhs = mg.Field(dims).fill_rhs(0)
    its = 400
    for resident in (1, 0):
        mg.set_option("resident_solver", resident)
        g = mg.GCR(A, mg.GCR_Param(0, restart, its, 1e-300, False))
        x = mg.Field(dims)
        g.solve(rhs, x)
        mg.lib().mgcr_synchronize()
        best = 1e9
        for _ in range(5):
            x.set_zero()
            mg.lib().mgcr_synchronize()
            t0 = time.perf_counter()
            g.solve(rhs, x)
            mg.lib().mgcr_synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%d^3 GCR(%d) %s: %d iterations, %.2f us per iteration, final %.3e" % (n, restart, "resident    " if resident else "multi-kernel", g.last_iterations, best * 1e6 / g.last_iterations, g.last_history[-1]), flush=True)
mg.set_option("resident_solver", 1)
