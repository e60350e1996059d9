"""Does a direct coarsest solve (MG_Param(coarse_direct=...)) pay for BASELINE configs[2]?  Poisson 256^3, flexible GCR(5) to
1e-8: the 3-level hierarchy of the bench (coarsest 64^3: too large for a dense inverse) against deeper hierarchies whose
coarsest level (16^3 ... 4^3) is solved by GCR (tol 1e-2 / 50) or directly.   python tools/direct_coarse_experiment.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import Field, GCR, GCR_Param, MG, MG_Param, Mesh, Sparse, problems

mg.init(0)
n = 256
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = Sparse(N, ncol, rowptr, col, val)
del rowptr, col, val
dims = (n, n, n)
rhs, x, y = Field(dims).fill_rhs(0), Field(dims), Field(dims)
ones = np.ones((1, N), np.complex128)
for levels, direct in ((2, 0), (4, 0), (4, 4096), (5, 0), (5, 1024), (6, 0), (6, 64)):
    t0 = time.perf_counter()
    M = MG(A, MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                       levels, None, None, null_vectors=ones, coarse_direct=direct))
    mg.lib().mgcr_synchronize()
    setup = time.perf_counter() - t0
    M(rhs, out=y)
    mg.lib().mgcr_synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        M(rhs, out=y)
    mg.lib().mgcr_synchronize()
    vc = (time.perf_counter() - t0) / 10
    g = GCR(A, GCR_Param(0, 5, 300, 1e-8, False, None, M, flexible=True, check_every=2))
    x.set_zero()
    g.solve(rhs, x)
    x.set_zero()
    mg.lib().mgcr_synchronize()
    t0 = time.perf_counter()
    g.solve(rhs, x)
    mg.lib().mgcr_synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"coarse_levels": levels, "coarsest_rows": M.level_info(levels)["dim"], "coarse_direct": direct, "setup_s": round(setup, 3),
                      "vcycle_ms": round(vc * 1e3, 3), "outer_iterations": g.last_iterations, "seconds_to_1e-8": round(dt, 4)}), flush=True)
    del g, M
