"""One MG-preconditioned flexible GCR solve at 256^3 (bench.py's mg256 workload) for a kernel trace:
   rocprofv3 --kernel-trace -d gpurun_out/mgs -o mgs -- python3 tools/mg_solve_prof.py ; python tools/mg_solve_prof.py --report gpurun_out/mgs/mgs_results.db"""
import sys
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import re
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    n0 = max(r[3] for r in rows if "expand_add" in r[0])
    ex = [i for i, r in enumerate(rows) if "expand_add" in r[0] and r[3] == n0]
    a, b = ex[-3], ex[-2]          # one outer iteration: from one fine-level prolongation to the next
    t0 = rows[a][1]
    print("one outer iteration = %d kernels, %.1f us wall" % (b - a, (rows[b][1] - t0) / 1e3))
    for r in rows[a:b]:
        nm = r[0].replace("void mgcr::", "").replace("mgcr::", "")
        nm = re.sub(r"\(.*", "", nm)[:60]
        print("%9.1f  %8.1f us  grid %8d  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], nm))
    raise SystemExit(0)
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
mg.init(0)
n = 256
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = Sparse(N, ncol, rowptr, col, val)
dims = (n, n, n)
prm = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)), 2, None, None, null_vectors=np.ones((1, N), np.complex128))
M = MG(A, prm)
b = Field(dims).fill_rhs(0)
x = Field(dims)
outer = GCR(A, GCR_Param(0, 5, 200, 1e-8, False, None, M, flexible=True, check_every=2))
for _ in range(2):
    x.set_zero()
    outer.solve(b, x)
mg.lib().mgcr_synchronize()
print("iterations", outer.last_iterations, "converged", outer.last_converged)
