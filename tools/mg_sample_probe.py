"""Adaptive-aggregation MG on the reference's 4x4 sample operator: how close is the GPU cycle to the oracle's in DEVICE order?"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
from oracle import oracle as orc
mg.init()
mg.lib().mgcr_set_small_solve_rows(0)
path = os.path.join("tests", "golden", "4x4parsed.txt")
if not os.path.exists(path):
    import gzip, shutil
    with gzip.open(path + ".gz", "rb") as f, open("/tmp/4x4parsed.txt", "wb") as g:
        shutil.copyfileobj(f, g)
    path = "/tmp/4x4parsed.txt"
DIMS = (4, 4, 4, 4, 4, 3)
D = read_data(os.path.basename(path), directory=os.path.dirname(path))
k = 0.19
dirac = DiracOp(D, k)
prm = MG_Param(Mesh(DIMS), 2, 2, GCR_Param(0, 10, 10, 1e-8, False), GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)), 1, None, None)
M = MG(dirac, prm)
pv, agg = M.prolongator(0)
nrow, ncol, rowptr, col, val = orc.read_text_csr(path)
Do = orc.csr(nrow, ncol, rowptr, col, val)
Ao = orc.dirac(Do, k)
vecs = np.ascontiguousarray(pv.T)
Mo = orc.MG(Ao, rowptr, col, val, DIMS, (1, 1, 1, 1, 0, 0), 2, vecs, 2, orc.gcr_param(restart=10, max_iter=2, tol=1e-30), orc.gcr_param(restart=10, max_iter=50, tol=1e-2), shift=k, vectors_are_prolongator=True)
pvo, aggo = Mo.prolongator(0)
print("prolongator: equal", np.array_equal(pv.reshape(pvo.shape), pvo), "max dev", np.abs(pv.reshape(pvo.shape) - pvo).max())
lay = dirac.ell_layout()
print("layout", lay, "levels", [M.level_info(l) for l in range(2)])
lanes = lay["lanes"] > 1 or lay["tail_rows"] > 0
b = problems.rhs_grid(3072, 4)
y = M(Field(DIMS, b)).to_numpy().ravel()
with orc.device_order(ell_width=lay["ell_width"] if lanes else -1, ell_lanes=lay["lanes"], tail_cap=lay["tail_chunk_cap"], lean=True, recurrence_residual=True):
    yo = Mo(b)
print("cycle: equal", np.array_equal(y, yo), "max rel dev", np.abs(y - yo).max() / np.abs(yo).max())
nc = M.level_info(1)["dim"]
w = problems.rhs_grid(nc, 4)
print("coarse apply equal", np.array_equal(M.level_operator(1)(Field((nc,), w)).to_numpy(), Mo.level_op(1)(w)))
outer = GCR(dirac, GCR_Param(0, 5, 200, 1e-10, False, None, M, flexible=True))
x = Field(DIMS).set_zero()
outer.solve(Field(DIMS, b), x)
with orc.device_order(ell_width=lay["ell_width"] if lanes else -1, ell_lanes=lay["lanes"], tail_cap=lay["tail_chunk_cap"], lean=True, recurrence_residual=True):
    xo, ho, ito, co = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=200, tol=1e-10, right=Mo, flexible=True), b)
print("MG-GCR: its", outer.last_iterations, ito, "hist equal", np.array_equal(outer.last_history, ho), "x equal", np.array_equal(x.to_numpy().ravel(), xo))
