"""GCR(5) iterations/s on n^3 Poisson grids (one process per size and variant):  python tools/gcr_size_sweep.py 192 320 [ENV[=a,b]]
   Each size runs in child processes with ENV (default MGCR_PLANE_WALK) = 0 and 1, interleaved, twice."""
import os
import subprocess
import sys
import time

CHILD = r'''
import sys, time
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
n = int(sys.argv[1])
its = 25
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = Sparse(N, ncol, rowptr, col, val)
del rowptr, col, val
b = Field((n, n, n)).fill_rhs(0)
x = Field((n, n, n))
g = GCR(A, GCR_Param(0, 5, its, 1e-30, False))
best = 0.
for rep in range(4):
    x.set_zero()
    mg.lib().mgcr_synchronize()
    t = time.perf_counter()
    g.solve(b, x)
    mg.lib().mgcr_synchronize()
    dt = time.perf_counter() - t
    if rep: best = max(best, its / dt)
y = Field((n, n, n))
ms = A.bench_apply(b, y, 10)
print("n %d it/s %.1f  final %.6e  kind %d  apply_ms %.4f" % (n, best, g.last_history[-1], A.xr_fuse_kind(), ms))
'''
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()]
var = [a for a in sys.argv[1:] if not a.isdigit()]
var = var[0] if var else "MGCR_PLANE_WALK"
vals = ("0", "1")
if "=" in var:
    var, v = var.split("=")
    vals = tuple(v.split(","))
for n in sizes:
    for rep in range(1 if n >= 448 else 2):
        for v in vals:
            t = time.time()
            p = subprocess.run([sys.executable, "-c", CHILD, str(n)], env=dict(os.environ, **{var: v}), capture_output=True, text=True, timeout=900)
            out = p.stdout.strip().splitlines()[-1] if p.returncode == 0 and p.stdout.strip() else "FAILED " + p.stderr[-600:]
            print("%s=%s  %s  (%.0f s)" % (var, v, out, time.time() - t), flush=True)
