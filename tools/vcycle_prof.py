import sys, numpy as np
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
rhs = Field(dims).fill_rhs(0); y = Field(dims)
for _ in range(6): M(rhs, out=y)
mg.lib().mgcr_synchronize()
