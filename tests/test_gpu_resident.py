"""The one-launch resident GCR (csrc/gcr_resident.hip) against the multi-kernel path (csrc/gcr.hip): the same solve on the
same operator has to give the same iteration count, the same residual history and the same x, bit for bit — both fold
their reductions with the same tree in the same order — and both are checked against the CPU oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

pytestmark = pytest.mark.gpu


def _solve(A, dims, p, rhs, resident, x0=None):
    import mgpreconditionedgcr_amd as mg
    prev = mg.set_option("resident_solver", 1 if resident else 0)
    try:
        g = mg.GCR(A, p)
        b = mg.Field(dims, rhs)
        x = mg.Field(dims, x0 if x0 is not None else np.zeros(rhs.size, np.complex128))
        before = mg.stat("resident_solves")
        g.solve(b, x)
        assert mg.stat("resident_solves") - before == (1 if resident else 0), "the solve did not take the path under test"
        return x.to_numpy().copy(), g.last_history.copy(), g.last_iterations, g.last_converged
    finally:
        mg.set_option("resident_solver", prev)


def _poisson(n, shift=0.0):
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import problems
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    if shift:
        val = val.copy()
        val[col == np.repeat(np.arange(N), np.diff(rowptr))] += shift
    return mg.Sparse(N, ncol, rowptr, col, val), (n, n, n), (N, rowptr, col, val)


def _rhs(N, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(N) + 1j * rng.standard_normal(N)


@pytest.mark.parametrize("n,restart,max_it,tol", [(32, 10, 50, 1e-30), (32, 5, 23, 1e-30), (36, 10, 40, 1e-6), (32, 5, 50, 1e-5),
                                                  (40, 10, 7, 1e-30), (64, 10, 50, 1e-2), (64, 10, 30, 1e-30), (33, 10, 3, 1e-30),
                                                  (48, 7, 4, 1e-30), (63, 10, 21, 1e-30)])
def test_resident_equals_multi_kernel_path_bit_for_bit(n, restart, max_it, tol):
    import mgpreconditionedgcr_amd as mg
    A, dims, _ = _poisson(n, shift=0.05)
    rhs = _rhs(n ** 3, 100 + n)
    p = mg.GCR_Param(0, restart, max_it, tol, False)
    xr, hr, itr, cr = _solve(A, dims, p, rhs, True)
    xc, hc, itc, cc = _solve(A, dims, p, rhs, False)
    assert itr == itc and cr == cc
    assert np.array_equal(hr, hc)
    assert np.array_equal(xr, xc)
    assert np.all(np.isfinite(xr))


def test_resident_keeps_a_given_x_as_the_vector_it_adds_to():
    """use_x0 = False: the reference starts from r = rhs whatever x holds, and adds its updates to that x (src/GCR.h:189)."""
    import mgpreconditionedgcr_amd as mg
    n = 32
    A, dims, _ = _poisson(n, shift=0.05)
    rhs = _rhs(n ** 3, 7)
    x0 = _rhs(n ** 3, 8)
    p = mg.GCR_Param(0, 10, 25, 1e-30, False)
    xr, hr, itr, _ = _solve(A, dims, p, rhs, True, x0)
    xc, hc, itc, _ = _solve(A, dims, p, rhs, False, x0)
    assert itr == itc and np.array_equal(hr, hc) and np.array_equal(xr, xc)


def test_resident_against_the_oracle():
    import mgpreconditionedgcr_amd as mg
    from oracle import oracle as orc   # checker only
    n = 32
    A, dims, (N, rowptr, col, val) = _poisson(n, shift=0.1)
    rhs = _rhs(N, 3)
    p = mg.GCR_Param(0, 5, 30, 1e-10, False)
    xr, hr, itr, _ = _solve(A, dims, p, rhs, True)
    xo, ho, ito, _ = orc.gcr_solve(orc.csr(N, N, rowptr, col, val), orc.gcr_param(restart=5, max_iter=30, tol=1e-10), rhs)
    assert itr == ito
    # the oracle sums its dot products sequentially, the device in a tree: equal to rounding
    np.testing.assert_allclose(hr, ho[: itr + 1], rtol=1e-9, atol=0)
    assert np.linalg.norm(xr - xo) <= 1e-9 * np.linalg.norm(xo)


def test_resident_is_faster_per_iteration():
    """64^3, GCR(10): the point of the exercise (MI355X: 18.9 against 28.9 us per iteration when this test was written)"""
    import time
    import mgpreconditionedgcr_amd as mg
    n = 64
    A, dims, _ = _poisson(n, shift=0.0)
    b = mg.Field(dims).fill_rhs(0)
    t = {}
    for resident in (1, 0):
        prev = mg.set_option("resident_solver", resident)
        try:
            g = mg.GCR(A, mg.GCR_Param(0, 10, 400, 1e-300, False))
            x = mg.Field(dims)
            g.solve(b, x)
            best = 1e9
            for _ in range(3):
                x.set_zero()
                mg.lib().mgcr_synchronize()
                t0 = time.perf_counter()
                g.solve(b, x)
                mg.lib().mgcr_synchronize()
                best = min(best, time.perf_counter() - t0)
            t[resident] = best / g.last_iterations
        finally:
            mg.set_option("resident_solver", prev)
    print("64^3 GCR(10): resident %.1f us, multi-kernel %.1f us per iteration" % (t[1] * 1e6, t[0] * 1e6))
    assert t[1] * 1.2 < t[0]


def test_resident_gives_up_instead_of_hanging(tmp_path):
    """A workgroup that never shows up (here: told to leave after step 0) must not leave the others spinning: they give up
    after a bounded number of polls — and GCR::solve then REPEATS the solve on the multi-kernel path from the untouched
    right-hand side and the caller's x (re-zeroed when the Field was known to be zero, restored from a copy otherwise): the
    first call already returns the right answer, bit for bit what the multi-kernel path gives.  Runs in a child process: the
    switches are read from the environment once."""
    import subprocess
    code = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import problems
n = 32
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = mg.Sparse(N, ncol, rowptr, col, val)
g = mg.GCR(A, mg.GCR_Param(0, 10, 30, 1e-30, False))
b = mg.Field((n, n, n)).fill_rhs(0)
x0 = problems.rhs_grid(N, 3)
res = []
for case in ("zeroed", "x0"):
    x = mg.Field((n, n, n)).set_zero() if case == "zeroed" else mg.Field((n, n, n), x0)
    f0, r0 = mg.stat("one_launch_fallbacks"), mg.stat("resident_solves")
    g.solve(b, x)          # first call of the process: the resident launch gives up, the solve is repeated inside the library
    res.append((case, mg.stat("one_launch_fallbacks") - f0, mg.stat("resident_solves") - r0, x.to_numpy(), g.last_history.copy(), g.last_iterations))
print("SYNC-OK" if mg.lib().mgcr_synchronize() == 0 else "SYNC-FAILS")
# reference: the multi-kernel path (the one-launch paths have switched themselves off)
for case, fb, rs, xs, h, it in res:
    x = mg.Field((n, n, n)).set_zero() if case == "zeroed" else mg.Field((n, n, n), x0)
    g.solve(b, x)
    same = np.array_equal(xs, x.to_numpy()) and np.array_equal(h, g.last_history) and it == g.last_iterations == 30
    print(case, "fallbacks", fb, "resident", rs, "SAME" if same and np.isfinite(xs).all() else "DIFFERENT")
'''
    import os
    env = dict(os.environ, MGCR_TEST_RESIDENT_STALL="3", MGCR_TEST_RESIDENT_SPIN_LIMIT="20000")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120,
                         cwd=os.path.join(os.path.dirname(__file__), ".."))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "SYNC-OK" in out.stdout, out.stdout
    assert "zeroed fallbacks 1 resident 1 SAME" in out.stdout, out.stdout      # gave up once, repeated, right
    assert "x0 fallbacks 0 resident 0 SAME" in out.stdout, out.stdout          # the path is off from then on


def test_one_launch_paths_ask_the_runtime_for_co_residency():
    """The one-launch kernels' workgroups wait for each other, so whether the grid fits the chip at once is asked of the runtime
    (hipOccupancyMaxActiveBlocksPerMultiprocessor on the instantiation that would run, with its dynamic LDS), not derived from
    what the kernels are meant to need.  Here the answer is forced to 0 workgroups per CU: neither path may be taken, and the
    solves come out the same."""
    import subprocess
    code = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import problems
for n, re, its in ((32, 10, 30), (96, 5, 12)):
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    A = mg.Sparse(N, ncol, rowptr, col, val)
    g = mg.GCR(A, mg.GCR_Param(0, re, its, 1e-30, False))
    b = mg.Field((n, n, n)).fill_rhs(0)
    x = mg.Field((n, n, n)).set_zero()
    g.solve(b, x)
    print(n, "resident", mg.stat("resident_solves"), "step_build", mg.stat("step_build_launches"), "fallbacks", mg.stat("one_launch_fallbacks"),
          "FINITE" if np.isfinite(x.to_numpy()).all() and g.last_iterations == its else "BAD")
'''
    import os
    outs = {}
    for occ in ("0", None):
        env = dict(os.environ)
        if occ is not None:
            env["MGCR_TEST_OCCUPANCY"] = occ
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180,
                             cwd=os.path.join(os.path.dirname(__file__), ".."))
        assert out.returncode == 0, out.stderr[-2000:]
        outs[occ] = out.stdout
    assert "32 resident 0 step_build 0 fallbacks 0 FINITE" in outs["0"] and "96 resident 0 step_build 0 fallbacks 0 FINITE" in outs["0"], outs["0"]
    # ... and what the runtime really says on this chip lets both paths run
    assert "32 resident 1 step_build 0 fallbacks 0 FINITE" in outs[None], outs[None]
    l96 = [l for l in outs[None].splitlines() if l.startswith("96 ")][0]
    assert int(l96.split()[4]) > 0 and "fallbacks 0 FINITE" in l96, outs[None]


def _diag_dominant(N, rowptr, col, val, factor=2.0):
    """add to every diagonal entry enough to dominate its row (GCR then converges whatever the off-diagonal values)"""
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    rowsum = np.bincount(rows, weights=np.abs(val), minlength=N)
    val = val.copy()
    d = col == rows
    val[d] = factor * rowsum[rows[d]] + 1.0
    return val


@pytest.mark.parametrize("case", ["ell-small", "ell-random-columns", "dictionary-with-values", "dictionary-columns-only"])
@pytest.mark.parametrize("restart,max_it", [(10, 37), (5, 12), (10, 4)])
def test_resident_general_storage_bit_for_bit(case, restart, max_it):
    """every storage format of a Sparse the fused kernels read — plain ELL slab (matrices below 2^15 rows, or without
    repeating rows), row-pattern dictionary with values, dictionary for the columns + value slab — through the resident
    solver: the bits of the multi-kernel path"""
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import problems
    rng = np.random.default_rng(5)
    opts = {}
    if case == "ell-small":
        n = 20   # 8000 rows: below the dictionary's 2^15
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        val = val + 0.0j
        val[col == np.repeat(np.arange(N), np.diff(rowptr))] += 0.3 - 0.1j
        dims, want = (n, n, n), 0
    elif case == "ell-random-columns":
        N = 40000
        col = np.sort(rng.integers(0, N, size=(N, 6)), axis=1)
        col[:, 0] = np.arange(N)   # a diagonal entry per row (first in storage order is fine: CSR order is what it is)
        col = np.sort(col, axis=1).ravel()
        rowptr = np.arange(N + 1, dtype=np.int64) * 6
        val = rng.standard_normal(col.size) + 1j * rng.standard_normal(col.size)
        # duplicates of the diagonal column may occur: make every one of them big
        val = _diag_dominant(N, rowptr, col, val)
        ncol, dims, want = N, (N,), 0
    else:
        n = 34   # 39304 rows
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        dims = (n, n, n)
        if case == "dictionary-with-values":
            val = val * (0.75 - 0.5j)
            val[col == np.repeat(np.arange(N), np.diff(rowptr))] += 0.2
            opts, want = {"stencil_storage": 0}, 1
        else:
            val = rng.standard_normal(val.size) + 1j * rng.standard_normal(val.size)
            val = _diag_dominant(N, rowptr, col, val)
            want = 2
    prev = {k: mg.set_option(k, v) for k, v in opts.items()}
    try:
        A = mg.Sparse(N, ncol, rowptr, col, val)
    finally:
        for k, v in prev.items():
            mg.set_option(k, v)
    assert A.storage_format()[0] == want
    rhs = _rhs(N, 17)
    p = mg.GCR_Param(0, restart, max_it, 1e-11, False)
    xr, hr, itr, cr = _solve(A, dims, p, rhs, True)
    xc, hc, itc, cc = _solve(A, dims, p, rhs, False)
    assert itr == itc and cr == cc and itr >= min(max_it, 3)
    assert np.array_equal(hr, hc)
    assert np.array_equal(xr, xc)
    assert hr[-1] < hr[0]


def test_vcycle_with_resident_coarsest_solve_is_bit_identical():
    """a V-cycle whose coarsest solve takes the one-launch path returns the bits of the same cycle on the multi-kernel path, and
    the preconditioned outer solve the same history"""
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import problems
    n = 64
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    dims = (n, n, n)
    out = []
    for resident in (1, 0):
        prev = mg.set_option("resident_solver", resident)
        try:
            A = mg.Sparse(N, ncol, rowptr, col, val)
            prm = mg.MG_Param(mg.Mesh(dims), 2, 1, None, mg.GCR(mg.GCR_Param(0, 10, 50, 1e-2, False)),
                              mg.GCR(mg.GCR_Param(0, 10, 2, 1e-30, False)), 1, None, None, null_vectors=np.ones((1, N), np.complex128))
            M = mg.MG(A, prm)
            b = mg.Field(dims).fill_rhs(2)
            before = mg.stat("resident_solves")
            y = M(b).to_numpy()
            took = mg.stat("resident_solves") - before
            outer = mg.GCR(A, mg.GCR_Param(0, 5, 40, 1e-9, False, None, M, flexible=True))
            x = mg.Field(dims).set_zero()
            outer.solve(b, x)
            out.append((took, y, outer.last_history.copy(), x.to_numpy()))
        finally:
            mg.set_option("resident_solver", prev)
    assert out[0][0] >= 1 and out[1][0] == 0   # the 32^3 coarsest level (32768 rows) went through the resident solver, or not
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    assert out[0][2][-1] < 1e-9


@pytest.mark.parametrize("n,stencil", [(32, 1), (33, 0), (20, 1)])
def test_resident_dirac_operator(n, stencil):
    """DiracOp (y = x - k A x, src/Operator.h:569-575) through the resident solver: stencil view, dictionary and ELL slab"""
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import problems
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    prev = mg.set_option("stencil_storage", stencil)
    try:
        A = mg.Sparse(N, ncol, rowptr, col, val)
    finally:
        mg.set_option("stencil_storage", prev)
    D = mg.DiracOp(A, 0.07 - 0.03j)
    rhs = _rhs(N, 31)
    p = mg.GCR_Param(0, 10, 400, 1e-8, False)
    xr, hr, itr, cr = _solve(D, (n, n, n), p, rhs, True)
    xc, hc, itc, cc = _solve(D, (n, n, n), p, rhs, False)
    assert itr == itc and cr == cc and cr
    assert np.array_equal(hr, hc) and np.array_equal(xr, xc)
    # and it solves the system: || rhs - D x || small
    r = rhs - D(mg.Field((n, n, n), xr)).to_numpy()
    assert np.linalg.norm(r) <= 2e-8 * np.linalg.norm(rhs)


@pytest.mark.parametrize("max_it,tol,zero_rhs", [(1, 1e-30, False), (2, 1e-30, False), (50, 0.5, False), (11, 1e-30, False), (7, 1e-30, True)])
def test_resident_corner_cases(max_it, tol, zero_rhs):
    """one iteration; a solve that stops after its first step; one step past a closed cycle; a zero right-hand side (0 / 0: the
    reference's NaN, which ends the solve) — same iteration counts, histories and x as the multi-kernel path, NaNs included"""
    import mgpreconditionedgcr_amd as mg
    n = 32
    A, dims, _ = _poisson(n, shift=0.05)
    rhs = np.zeros(n ** 3, np.complex128) if zero_rhs else _rhs(n ** 3, 77)
    p = mg.GCR_Param(0, 10, max_it, tol, False)
    xr, hr, itr, cr = _solve(A, dims, p, rhs, True)
    xc, hc, itc, cc = _solve(A, dims, p, rhs, False)
    assert itr == itc and cr == cc
    assert np.array_equal(hr, hc, equal_nan=True) and np.array_equal(xr, xc, equal_nan=True)


def test_cross_xcd_coherence_of_sc1_accesses():
    """What the one-launch paths stand on, checked on the chip at every test run (it is hardware behaviour, not a promise of
    the HIP memory model): `buffer_store ... sc1` rows of one workgroup are read correctly by `buffer_load ... sc1` in
    workgroups of other XCDs across the library's fence-free exchange — 3 000 dependent steps, one workgroup per CU, 0 rows
    wrong; and the control with ordinary loads / stores does go wrong (the test can fail)."""
    import ctypes as C
    import mgpreconditionedgcr_amd as mg
    mg.init()
    bad = C.c_int64(-2)
    from mgpreconditionedgcr_amd._lib import check as _chk
    for steps in (1000, 1001, 3000):
        _chk(mg.lib().mgcr_selftest_coherence(steps, 1, C.byref(bad)))
        assert bad.value == 0, "%d rows wrong after %d steps" % (bad.value, steps)
    _chk(mg.lib().mgcr_selftest_coherence(1001, 0, C.byref(bad)))
    assert bad.value > 0
    # the solvers' paths are untouched by the self-test
    assert mg.lib().mgcr_synchronize() == 0


def test_nested_resident_solve_that_gives_up_is_repaired_by_the_outer_solve():
    """The coarsest solve of a V-cycle runs as one launch; when THAT launch gives up (a workgroup told to leave), the cycle's output
    is NaN and so is the outer solve — which the library then repeats on the multi-kernel paths: the caller's first call returns a
    converged, finite solution (here x was zeroed: no copy needed), the same as a process that never took the one-launch paths."""
    import subprocess
    code = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
n = 32
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = Sparse(N, ncol, rowptr, col, val)
prm = MG_Param(Mesh((n, n, n)), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)), 1, None, None,
               null_vectors=np.ones((1, N), np.complex128))
M = MG(A, prm)
b = Field((n, n, n)).fill_rhs(0)
x = Field((n, n, n)).set_zero()
outer = GCR(A, GCR_Param(0, 5, 100, 1e-9, False, None, M, flexible=True))
outer.solve(b, x)
r = b - A(x)
print("fallbacks", mg.stat("one_launch_fallbacks"), "resident", mg.stat("resident_solves"), "its", outer.last_iterations, "converged", outer.last_converged,
      "finite", bool(np.isfinite(x.to_numpy()).all()), "true_res_ok", bool(r.norm() / b.norm() <= 2e-9))
np.save(sys.argv[1], np.concatenate([outer.last_history, x.to_numpy().ravel().view(np.float64)]))
'''
    import os
    import tempfile
    d = tempfile.mkdtemp()
    outs = {}
    for tag, env_add in (("stalled", dict(MGCR_TEST_RESIDENT_STALL="3", MGCR_TEST_RESIDENT_SPIN_LIMIT="20000")), ("never", dict(MGCR_RESIDENT="0"))):
        env = dict(os.environ, MGCR_SMALL_SOLVE_ROWS="0", **env_add)
        f = os.path.join(d, tag + ".npy")
        out = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=200,
                             cwd=os.path.join(os.path.dirname(__file__), ".."))
        assert out.returncode == 0, out.stderr[-2000:]
        outs[tag] = (out.stdout, np.load(f))
    assert "fallbacks 1 " in outs["stalled"][0] and "converged True finite True true_res_ok True" in outs["stalled"][0], outs["stalled"][0]
    assert "fallbacks 0 resident 0 " in outs["never"][0], outs["never"][0]
    assert np.array_equal(outs["stalled"][1], outs["never"][1])     # history and x: the bits of the multi-kernel paths
