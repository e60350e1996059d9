"""Seeded random sweep of the GCR parameter space against the oracle (which tests/test_oracle_golden.py pins to the
real reference): every mode (restart 1..17, truncation 1..12, full), short and long solves, with and without the
DiracOp shift, with and without use_x0, on systems that take the one-workgroup solver, the slab layout, and the
row-pattern dictionary — i.e. every combination of the lean / classic, fused / chunked, aliased / copied code
paths of gcr.hip.  Same tolerances as tests/test_gpu_parity.py (module docstring there): residual history within
max(1e-9 relative, 8 x the reference algorithm's own re-association sensitivity), iteration count within its
spread, recurrence residual == true residual, x equal to the oracle's where the solve is well conditioned.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("mgpreconditionedgcr_amd")
from mgpreconditionedgcr_amd import DiracOp, Field, GCR, GCR_Param, Sparse, problems  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_gpu_parity import hist_close, its_close, x_close  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _init():
    mg.init()
    yield


def _system(rng):
    kind = rng.choice(["poisson-small", "poisson-slab", "poisson-pattern", "random"])
    if kind == "random":
        N = int(rng.integers(200, 3000))
        rowptr, col, val = problems.random_csr(N, N, rng, min_len=1, max_len=8)
        rows = np.repeat(np.arange(N), np.diff(rowptr))
        rowsum = np.bincount(rows, weights=np.abs(val), minlength=N)
        # make it diagonally dominant (GCR then converges): add a dominant diagonal entry at the end of each row
        newptr = rowptr + np.arange(N + 1)
        ncol_arr, nval = np.empty(newptr[-1], np.int64), np.empty(newptr[-1], np.complex128)
        for r in range(N):
            s, e = rowptr[r], rowptr[r + 1]
            ncol_arr[newptr[r]:newptr[r] + (e - s)] = col[s:e]
            nval[newptr[r]:newptr[r] + (e - s)] = val[s:e]
            ncol_arr[newptr[r + 1] - 1] = r
            nval[newptr[r + 1] - 1] = 1.5 * rowsum[r] + 1.0
        return kind, N, newptr, ncol_arr, nval
    n = {"poisson-small": int(rng.integers(4, 10)), "poisson-slab": int(rng.integers(11, 24)), "poisson-pattern": 33}[kind]
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    val = val * complex(1.0, float(rng.choice([0.0, 0.125, -0.25])))
    return kind, N, rowptr, col, val


@pytest.mark.parametrize("seed", range(96))
def test_random_parameters_vs_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    kind, N, rowptr, col, val = _system(rng)
    mode = rng.choice(["restart", "truncation", "full"], p=[0.6, 0.25, 0.15])
    kw = dict(max_iter=int(rng.choice([0, 1, 2, 3, 7, 20, 45])), tol=float(rng.choice([1e-30, 1e-5, 1e-9])))
    if mode == "restart":
        kw["restart"] = int(rng.integers(1, 18))
    elif mode == "truncation":
        kw["truncation"] = int(rng.integers(1, 13))
    shift = complex(rng.uniform(0.02, 0.1), rng.uniform(-0.05, 0.05)) if rng.random() < 0.35 else None
    use_x0 = bool(rng.random() < 0.3)
    b = problems.rhs_grid(N, int(rng.integers(0, 50)))
    x0 = problems.rhs_grid(N, 77) * 0.1 if use_x0 else None
    Ao = orc.csr(N, N, rowptr, col, val)
    A = Sparse(N, N, rowptr, col, val)
    keep = [A]
    if shift is not None:
        Ao, A = orc.dirac(Ao, shift), DiracOp(A, shift)
    po = orc.gcr_param(use_x0=use_x0, **kw)
    xo, ho, ito, co = orc.gcr_solve(Ao, po, b, x0)
    _, sens, its_rng = orc.gcr_reorder_sensitivity(Ao, po, b, x0)
    gcr = GCR(A, GCR_Param(kw.get("truncation", 0), kw.get("restart", 0), kw["max_iter"], kw["tol"], False, use_x0=use_x0,
                           check_every=int(rng.choice([0, 1, 3, 50]))))
    fb = Field((N,), b)
    x = Field((N,), x0) if use_x0 else Field((N,)).set_zero()
    small = int(rng.choice([0, 1024, 16384]))   # one-workgroup solver off / default / forced for everything that fits
    mg.lib().mgcr_set_small_solve_rows(small)
    try:
        gcr.solve(fb, x)
    finally:
        mg.lib().mgcr_set_small_solve_rows(1024)
    what = "%s N=%d %s shift=%s x0=%s small=%d" % (kind, N, kw, shift, use_x0, small)
    assert its_close(gcr.last_iterations, ito, its_rng), what
    if gcr.last_iterations == ito:
        assert gcr.last_converged == co, what
    hist_close(gcr.last_history, ho, what, sens)
    if np.isfinite(gcr.last_history[-1]):
        x_close(x, A, fb, gcr, xo if gcr.last_iterations == ito else None, sens)
    del keep
