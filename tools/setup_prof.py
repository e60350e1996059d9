"""Where the set-up of configs[2] goes (Sparse creation, MG hierarchy): wall times; under `rocprofv3 --kernel-trace --hip-trace`
the kernels and HIP calls behind them (found sten_planes_kernel's 2.3 M serialised atomics: 25 ms per 256^3 operator)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
mg.init(0)
n = 256
t0=time.perf_counter()
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
t1=time.perf_counter()
A = Sparse(N, ncol, rowptr, col, val)
mg.lib().mgcr_synchronize()
t2=time.perf_counter()
dims = (n, n, n)
nv = np.ones((1, N), np.complex128)
for rep in range(3):
    ta=time.perf_counter()
    prm = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)), 2, None, None, null_vectors=nv)
    M = MG(A, prm)
    mg.lib().mgcr_synchronize()
    tb=time.perf_counter()
    print("MG set-up %.1f ms" % ((tb-ta)*1e3))
print("host matrix %.2f s, Sparse create %.3f s" % (t1-t0, t2-t1))
