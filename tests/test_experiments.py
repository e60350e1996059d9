"""Callers / data formats either side of the hot path: the MatrixMarket
converter parse_data and the reference's experiment functions, on the GPU path."""
import gzip
import os

import numpy as np
import pytest

from oracle import oracle as orc


def _mtx_from_sample(sample_matrix_path, path, rng):
    nrow, ncol, rowptr, col, val = orc.read_text_csr(sample_matrix_path)
    rows = np.repeat(np.arange(nrow), np.diff(rowptr))
    # unordered triplets; every 50th entry split into two halves (duplicates must be summed)
    r, c, v = list(rows), list(col), list(val)
    for i in range(0, len(v), 50):
        r.append(r[i]); c.append(c[i]); v.append(v[i] * 0.5); v[i] = v[i] * 0.5
    perm = rng.permutation(len(v))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate complex general\n% comment line\n")
        f.write("%d %d %d\n" % (nrow, ncol, len(v)))
        for i in perm:
            f.write("%d %d %.17g %.17g\n" % (r[i] + 1, c[i] + 1, v[i].real, v[i].imag))
    return nrow, ncol, rowptr, col, val


def test_parse_data_roundtrip_cpu(sample_matrix_path, tmp_path):
    """MatrixMarket -> text CSR (src/Parse.cpp:9-61) -> read back: same structure, values to the
    6 significant digits the format keeps.  (No GPU: the converter is host code.)"""
    from mgpreconditionedgcr_amd.experiments import parse_data
    rng = np.random.default_rng(0)
    mtx, out = str(tmp_path / "m.mtx"), str(tmp_path / "parsed.txt")
    nrow, ncol, rowptr, col, val = _mtx_from_sample(sample_matrix_path, mtx, rng)
    parse_data(mtx, out)
    n2, c2, rp2, col2, val2 = orc.read_text_csr(out)
    assert (n2, c2) == (nrow, ncol) and np.array_equal(rp2, rowptr) and np.array_equal(col2, col)
    assert np.allclose(val2, val, rtol=2e-6, atol=1e-12)
    # header + row line + one line per entry, like the reference's writer
    with open(out) as f:
        lines = f.read().split("\n")
    assert lines[0] == "3072 3072 119808" and len(lines) == 2 + 119808


@pytest.mark.gpu
def test_experiments_on_gpu(sample_matrix_path):
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import DiracOp, read_data, experiments as ex, problems
    mg.init()
    D = read_data(os.path.basename(sample_matrix_path), directory=os.path.dirname(sample_matrix_path))
    dims = ex.DIMS_4x4
    # hermiticity check against the oracle on the same fields
    vmw, mvw, herm = ex.test_hermiticity(D, dims)
    nrow, ncol, rowptr, col, val = orc.read_text_csr(sample_matrix_path)
    Do = orc.csr(nrow, ncol, rowptr, col, val)
    v, w = problems.rhs_grid(3072, 2), problems.rhs_grid(3072, 5)
    assert abs(vmw - orc.dot(v, Do(w)).real) <= 1e-11 * abs(vmw) + 1e-11
    assert abs(mvw - orc.dot(Do(v), w).real) <= 1e-11 * abs(mvw) + 1e-11
    # k-sweep towards k_c = 0.20611 (src/main.cpp:699): iteration counts grow, the solve degrades
    res = ex.test_kcritical(D, dims, 0.20611, 0.17, steps=3, max_iter=3000)
    its = [r[1] for r in res]
    assert its[0] < its[1] < its[2] or not res[2][2]
    # projector identities: "machine precision" (report p.9; reference probe 2.2e-15 / 7.2e-16)
    out = ex.test_MG_property(DiracOp(D, 0.1), dims)
    assert out["rt_id"] < 1e-13 and out["trtr"] < 1e-13 and out["trm_tmr"] < 1e-12
