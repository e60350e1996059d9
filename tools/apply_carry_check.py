import os, subprocess, sys, tempfile
import numpy as np
CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
n, nz = 256, 16
N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
A = Sparse(N, ncol, rowptr, col, val)
rng = np.random.default_rng(5)
xv = rng.standard_normal(N) + 1j * rng.standard_normal(N)
x = Field((nz, n, n), xv)
y = A(x)
D = DiracOp(A, 0.13 + 0.02j)
z = D(x)
np.save(sys.argv[1], np.concatenate([y.to_numpy().ravel(), z.to_numpy().ravel()]).view(np.float64))
'''
outs = []
for v in ("0", "1"):
    f = os.path.join(tempfile.mkdtemp(), "o.npy")
    p = subprocess.run([sys.executable, "-c", CHILD, f], env=dict(os.environ, MGCR_APPLY_CARRY=v), capture_output=True, text=True, timeout=200)
    assert p.returncode == 0, p.stderr[-2000:]
    outs.append(np.load(f))
print("apply carry on/off identical:", np.array_equal(outs[0], outs[1]), "nonzero:", float(np.abs(outs[0]).max()))
