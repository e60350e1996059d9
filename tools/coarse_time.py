import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
mg.init(0)
n = 256
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = Sparse(N, ncol, rowptr, col, val)
dims = (n, n, n)
prm = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)), 2, None, None, null_vectors=np.ones((1, N), np.complex128))
M = MG(A, prm)
for lvl in (2, 1, 0):
    Ac = M.level_operator(lvl) if lvl else A
    nc = Ac.get_dim()
    b = Field((nc,)).fill_rhs(3); x = Field((nc,))
    for R in (10, 8, 5):
        prm_c = GCR_Param(0, R, 50, 1e-2, False) if lvl == 2 else GCR_Param(0, 10, 2, 1e-30, False)
        g = GCR(Ac, prm_c)
        best = 1e9
        for _ in range(4):
            x.set_zero(); mg.lib().mgcr_synchronize(); t = time.perf_counter(); g.solve(b, x); mg.lib().mgcr_synchronize(); best = min(best, time.perf_counter() - t)
        print("level", lvl, "rows", nc, "restart", R, "its", g.last_iterations, "ms", best * 1e3, "fmt", Ac.storage_format())
        if lvl != 2: break
