"""Where does a V-cycle with several near-null vectors per aggregate leave the oracle's bits?  (small case, in-process)"""
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
from oracle import oracle as orc
mg.init()
mg.lib().mgcr_set_small_solve_rows(0)
ne, levels = 2, 1
dims = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 16, 16)
N, ncol, rowptr, col, val = problems.poisson3d_box_csr(*dims)
A = Sparse(N, ncol, rowptr, col, val)
vecs = np.ones((ne, N), np.complex128)
vecs[1] = problems.rhs_grid(N, 11)
prm = MG_Param(Mesh(dims), 2, ne, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)), levels, None, None, null_vectors=vecs)
M = MG(A, prm)
Ao = orc.csr(N, ncol, rowptr, col, val)
lay = A.ell_layout()
band, per = orc.row_map(N, lay["reach"])
Ao.set_rowmap(band, per, orc.row_map_plane(N, lay["reach"]), init_banded=band > 0, xr_banded=band > 0 and A.xr_fuse_kind() in (1, 2))
print("level 0 map", band, per, "kind", A.xr_fuse_kind())
Mo = orc.MG(Ao, rowptr, col, val, dims, (1, 1, 1), 2, vecs, levels + 1, orc.gcr_param(restart=10, max_iter=2, tol=1e-30), orc.gcr_param(restart=10, max_iter=50, tol=1e-2))
pv_d, agg_d = M.prolongator(0)
pv_o, agg_o = Mo.prolongator(0)
print("aggregates equal", np.array_equal(agg_d, agg_o), "prolongator equal", np.array_equal(pv_d.reshape(pv_o.shape), pv_o), "max dev", np.abs(pv_d.reshape(pv_o.shape) - pv_o).max())
nc = M.level_info(1)["dim"]
w = problems.rhs_grid(nc, 4)
Ac_d = M.level_operator(1)(Field((nc,), w)).to_numpy()
Ac_o = Mo.level_op(1)(w)
print("coarse apply equal", np.array_equal(Ac_d, Ac_o), "max rel dev", np.abs(Ac_d - Ac_o).max() / np.abs(Ac_o).max())
v = problems.rhs_grid(N, 3)
Rd = M.restrict(Field(dims, v)).to_numpy()
print("restrict/expand available on oracle:", hasattr(Mo, "restrict"))
b = problems.rhs_grid(N, 0)
with orc.device_order(lean=True, recurrence_residual=True):
    yo = Mo(b)
yd = M(Field(dims, b)).to_numpy().ravel()
print("cycle equal", np.array_equal(yd, yo), "max rel dev", np.abs(yd - yo).max() / np.abs(yo).max())
# the MG-preconditioned outer solve, and the coarsest solver alone on several right-hand sides
outer = GCR(A, GCR_Param(0, 5, 8, 1e-30, False, None, M, flexible=True))
xs = Field(dims).set_zero()
outer.solve(Field(dims, b), xs)
with orc.device_order(lean=True, recurrence_residual=True):
    xo, ho, ito, _ = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=8, tol=1e-30, right=Mo, flexible=True), b)
h = outer.last_history
print("outer history equal", np.array_equal(h, ho), "first differing step", int(np.argmax(h != ho)) if not np.array_equal(h, ho) else -1)
Ac, Aco = M.level_operator(1), Mo.level_op(1)
for seed in range(6):
    w = problems.rhs_grid(nc, 20 + seed) * (1.0 if seed % 2 else 1e-3)
    g = GCR(Ac, GCR_Param(0, 10, 50, 1e-2, False))
    xc = Field((nc,)).set_zero()
    g.solve(Field((nc,), w), xc)
    with orc.device_order(lean=True):
        xco, hco, itco, cco = orc.gcr_solve(Aco, orc.gcr_param(restart=10, max_iter=50, tol=1e-2), w)
    print("coarse solve seed", seed, "its", g.last_iterations, itco, "hist equal", np.array_equal(g.last_history, hco), "x equal", np.array_equal(xc.to_numpy(), xco))
# the cycle on other right-hand sides
for seed in range(4):
    bb = problems.rhs_grid(N, 40 + seed)
    with orc.device_order(lean=True, recurrence_residual=True):
        yo2 = Mo(bb)
    yd2 = M(Field(dims, bb)).to_numpy().ravel()
    print("cycle on rhs", seed, "equal", np.array_equal(yd2, yo2), "max rel dev", np.abs(yd2 - yo2).max() / np.abs(yo2).max())
for mi in (1, 2):
    o2 = GCR(A, GCR_Param(0, 5, mi, 1e-30, False, None, M, flexible=True))
    x2 = Field(dims).set_zero()
    o2.solve(Field(dims, b), x2)
    with orc.device_order(lean=True, recurrence_residual=True):
        xo2, ho2, _, _ = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=mi, tol=1e-30, right=Mo, flexible=True), b)
    xg = x2.to_numpy().ravel()
    print("outer max_iter", mi, "hist", o2.last_history.tolist(), ho2.tolist(), "x equal", np.array_equal(xg, xo2), "x rel dev", np.abs(xg - xo2).max() / np.abs(xo2).max())
# the same outer solve with a ne = 1 hierarchy for comparison
prm1 = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)), levels, None, None, null_vectors=vecs[:1])
M1 = MG(A, prm1)
Mo1 = orc.MG(Ao, rowptr, col, val, dims, (1, 1, 1), 2, vecs[:1], levels + 1, orc.gcr_param(restart=10, max_iter=2, tol=1e-30), orc.gcr_param(restart=10, max_iter=50, tol=1e-2))
d1 = M1.level_operator(1).ell_layout()
b1, p1 = orc.row_map(M1.level_info(1)["dim"], d1["reach"])
keep = Mo1.level_op(1).set_layout(d1["ell_width"], d1["lanes"], d1["tail_chunk_cap"]).set_rowmap(b1, p1, 0, init_banded=b1 > 0, xr_banded=b1 > 0 and M1.level_operator(1).xr_fuse_kind() in (1, 2))
o3 = GCR(A, GCR_Param(0, 5, 2, 1e-30, False, None, M1, flexible=True))
x3 = Field(dims).set_zero()
o3.solve(Field(dims, b), x3)
with orc.device_order(lean=True, recurrence_residual=True):
    xo3, ho3, _, _ = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=2, tol=1e-30, right=Mo1, flexible=True), b)
print("ne = 1: hist equal", np.array_equal(o3.last_history, ho3), "x equal", np.array_equal(x3.to_numpy().ravel(), xo3))
