"""The reference's experiment / diagnostic functions (src/main.cpp) on the GPU path — the callers on
either side of the hot path — plus the MatrixMarket converter
(rank 2).  Each function returns its numbers (and prints what the reference prints).

    python -m mgpreconditionedgcr_amd.experiments kcritical  --dir data/sample_matrix
    python -m mgpreconditionedgcr_amd.experiments mg_property --dir data/sample_matrix
    python -m mgpreconditionedgcr_amd.experiments hermiticity --dir data/sample_matrix
    python -m mgpreconditionedgcr_amd.experiments parse  conf.mtx parsed.txt
"""
import argparse
import os

import numpy as np

from .api import DiracOp, Field, GCR, GCR_Param, MG, MG_Param, Mesh, Sparse, read_data

DIMS_4x4 = (4, 4, 4, 4, 4, 3)
K_CRITICAL = {"4x4parsed.txt": 0.20611, "8x8parsed.txt": 0.17865}  # src/main.cpp:699,722,845


def parse_data(file_loc, out_path):
    """parse_data (src/Parse.cpp:9-61): MatrixMarket `complex coordinate` -> the text-CSR format
    read_data consumes.  '%' comment lines skipped, then `rows cols elements`, then 1-based
    `row col re im` triplets in any order; duplicates are SUMMED (triplet constructor,
    src/Operator.h:250-294); output: `nrow ncol nnz`, the nrow row offsets, then `col (re,im)` per
    entry with the default 6 significant digits of operator<<."""
    with open(file_loc) as f:
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        rows, cols, elements = (int(t) for t in line.split())
        data = np.loadtxt(f, ndmin=2)
    if data.shape[0] != elements:
        raise ValueError("expected %d triplets, found %d" % (elements, data.shape[0]))
    from .hostalg import csr_from_triplets
    rowptr, cc, vals = csr_from_triplets(rows, cols, data[:, 0].astype(np.int64) - 1, data[:, 1].astype(np.int64) - 1,
                                         data[:, 2] + 1j * data[:, 3])
    with open(out_path, "w") as f:
        f.write("%d %d %d\n" % (rows, cols, cc.size))
        f.write(" ".join(str(int(x)) for x in rowptr[:rows]) + " ")
        for j, z in zip(cc, vals):
            f.write("\n%d (%s,%s)" % (j, _g6(z.real), _g6(z.imag)))
    return rows, cols, rowptr, cc, vals


def _g6(x):
    """operator<< of a double at the default precision 6 (what the reference's writer produces)."""
    s = "%.6g" % x
    return s


def test_hermiticity(mat, dims, seeds=(2, 5)):
    """src/main.cpp:541-570: <v, M w> vs <M v, w> (real parts), v, w deterministic random fields."""
    v = Field(dims).fill_rhs(seeds[0])
    w = Field(dims).fill_rhs(seeds[1])
    vmw = v.dot(mat(w)).real
    mvw = mat(v).dot(w).real
    hermitian = (vmw - mvw) < 1e-13  # the reference's (one-sided) test
    print("<v, Mw> = <Mv, w>: Matrix is Hermitian." if hermitian else "<v, Mw> != <Mv, w>: Matrix is NOT Hermitian!")
    return vmw, mvw, hermitian


def test_kcritical(D, dims, k_c, k_start, steps=5, restart=10, max_iter=50000, tol=1e-13, seed=42):
    """src/main.cpp:696-741: GCR iterations to tolerance as k approaches the critical hopping
    parameter k_c (the solve degrades and finally fails to converge there)."""
    field = Field(dims).fill_rhs(seed)
    out = []
    step = (k_c - k_start) / steps
    for i in range(steps):
        k = k_start + step * i
        dirac = DiracOp(D, k)
        sol = Field(dims).set_zero()
        gcr = GCR(dirac, GCR_Param(0, restart, max_iter, tol, False, check_every=50))
        gcr.solve(field, sol)
        print("k = %f: %s after %d steps, residual %.10e" % (k, "converged" if gcr.last_converged else "did not converge",
                                                             gcr.last_iterations, gcr.last_history[-1]))
        out.append((k, gcr.last_iterations, gcr.last_converged, float(gcr.last_history[-1])))
    return out


def test_MG_property(Dirac, dims, subblock=2, n_eigen=2, null_vectors=None, seed=42):
    """src/main.cpp:877-918 and MG::test_MG (src/MG.h:432-512): projector identities
    (R P R = R, P R P R = P R) and coarse-operator consistency P R A v = P A_c R v on span(P)."""
    eigen = GCR_Param(0, 10, 10, 1e-8, False)
    prm = MG_Param(Mesh(dims), subblock, n_eigen, eigen, GCR(GCR_Param(0, 10, 1, 1e-8, False)),
                   GCR(GCR_Param(0, 10, 1, 1e-8, False)), 1, None, None, null_vectors=null_vectors)
    mg = MG(Dirac, prm)
    rhs = Field(dims).fill_rhs(seed)
    inter1 = mg.restrict(rhs)
    inter2 = mg.expand(inter1)
    inter3 = mg.restrict(inter2)
    inter22 = mg.expand(mg.restrict(inter2))
    rt_id = (inter2 - inter22).norm()
    trtr = (inter3 - inter1).norm()
    print("RT - Id identity test difference = %.5e" % rt_id)
    print("TR TR - TR projector test difference = %.5e" % trtr)
    Ac = mg.level_operator(1)
    w = inter2  # in span(P)
    lhs = mg.expand(mg.restrict(Dirac(w)))
    rhs2 = mg.expand(Ac(mg.restrict(w)))
    rel = (lhs - rhs2).norm() / lhs.norm()
    print("Relative Difference between TRM and TmR = %.5e" % rel)
    return dict(rt_id=rt_id, trtr=trtr, trm_tmr=rel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["kcritical", "mg_property", "hermiticity", "parse"])
    ap.add_argument("args", nargs="*")
    ap.add_argument("--dir", default="../../data/sample_matrix/")
    ap.add_argument("--file", default="4x4parsed.txt")
    a = ap.parse_args()
    if a.what == "parse":
        parse_data(a.args[0], a.args[1])
        return
    D = read_data(a.file, directory=a.dir)
    dims = DIMS_4x4 if D.get_dim() == 3072 else (8, 8, 8, 8, 4, 3)
    if a.what == "hermiticity":
        test_hermiticity(D, dims)
    elif a.what == "kcritical":
        kc = K_CRITICAL.get(a.file, 0.20611)
        test_kcritical(D, dims, kc, kc - 0.00611 if a.file.startswith("4x4") else 0.174)
    else:
        test_MG_property(DiracOp(D, 0.1), dims)


if __name__ == "__main__":
    main()
