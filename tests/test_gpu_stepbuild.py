"""Apply + dot products + direction build as one launch (csrc/gcr_stepbuild.hip) against the two kernels it replaces: the same
solve must give the same iteration count, history and x bit for bit (same rows per thread, same accumulation order, same
fold tree), and the one-launch path must actually have run."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

pytestmark = pytest.mark.gpu


def _solve(A, dims, p, b, fused):
    import mgpreconditionedgcr_amd as mg
    prev = mg.set_option("step_build", 1 if fused else 0)
    try:
        g = mg.GCR(A, p)
        x = mg.Field(dims).set_zero()
        before = mg.stat("step_build_launches")
        g.solve(b, x)
        return x.to_numpy().copy(), g.last_history.copy(), g.last_iterations, mg.stat("step_build_launches") - before
    finally:
        mg.set_option("step_build", prev)


@pytest.mark.parametrize("n,restart,max_it,dirac", [(96, 5, 23, False), (128, 5, 20, False), (112, 10, 27, False), (100, 5, 12, True),
                                                    (128, 10, 14, True), (104, 3, 9, False)])
def test_step_build_equals_two_kernels_bit_for_bit(n, restart, max_it, dirac):
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import problems
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    A = mg.Sparse(N, ncol, rowptr, col, val)
    op = mg.DiracOp(A, 0.05 - 0.02j) if dirac else A
    dims = (n, n, n)
    b = mg.Field(dims).fill_rhs(n)
    p = mg.GCR_Param(0, restart, max_it, 1e-30, False)
    xf, hf, itf, nf = _solve(op, dims, p, b, True)
    xc, hc, itc, nc = _solve(op, dims, p, b, False)
    assert nc == 0 and nf > 0, (nf, nc)
    # every step but the solve's last that orthogonalises against at most 5 stored directions (the steps that close a cycle included)
    expect = sum(1 for it in range(1, max_it) if ((it - 1) % restart) + 1 <= 5)
    assert nf == expect, (nf, expect)
    assert itf == itc
    assert np.array_equal(hf, hc)
    assert np.array_equal(xf, xc)
    assert np.all(np.isfinite(xf)) and hf[-1] < hf[0]


def test_step_build_stops_like_the_two_kernels():
    """convergence inside a cycle: the stop predicate is raised by the build half, later launches of the solve return at once"""
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import problems
    n = 96
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    val = val.copy()
    val[col == np.repeat(np.arange(N), np.diff(rowptr))] += 2.0   # well conditioned: converges in a few steps
    A = mg.Sparse(N, ncol, rowptr, col, val)
    dims = (n, n, n)
    b = mg.Field(dims).fill_rhs(1)
    p = mg.GCR_Param(0, 5, 200, 1e-9, False)
    xf, hf, itf, nf = _solve(A, dims, p, b, True)
    xc, hc, itc, nc = _solve(A, dims, p, b, False)
    assert nf > 0 and itf == itc and itf < 60
    assert np.array_equal(hf, hc) and np.array_equal(xf, xc)
    assert hf[-1] <= 1e-9


def test_step_build_gives_up_instead_of_hanging():
    """a workgroup that never publishes (told to leave at once) must not leave the other 511 spinning: bounded polls, and
    GCR::solve repeats the solve on the three-kernel path inside the library — the FIRST call returns a finite, correct x
    (bit for bit the three-kernel result), for a Field known to be zero and for an x0 the library had to keep a copy of.
    Child process: the switches are read from the environment once."""
    import subprocess
    code = r'''
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import problems
n = 96
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = mg.Sparse(N, ncol, rowptr, col, val)
g = mg.GCR(A, mg.GCR_Param(0, 5, 12, 1e-30, False))
b = mg.Field((n, n, n)).fill_rhs(0)
x0 = problems.rhs_grid(N, 3)
x = mg.Field((n, n, n), x0)
f0, l0 = mg.stat("one_launch_fallbacks"), mg.stat("step_build_launches")
g.solve(b, x)              # the first one-launch step gives up; the solve is repeated from the copy of x0
first = (mg.stat("one_launch_fallbacks") - f0, mg.stat("step_build_launches") - l0, x.to_numpy(), g.last_history.copy(), g.last_iterations)
print("LIBRARY-USABLE" if mg.lib().mgcr_synchronize() == 0 else "STILL-FAILING")
x = mg.Field((n, n, n), x0)
l1 = mg.stat("step_build_launches")
g.solve(b, x)              # the three-kernel path (the one-launch paths switched themselves off)
print("fallbacks", first[0], "launches", first[1], "then", mg.stat("step_build_launches") - l1)
ok = np.isfinite(first[2]).all() and np.array_equal(first[2], x.to_numpy()) and np.array_equal(first[3], g.last_history) and first[4] == 12
r = b.to_numpy() - A(mg.Field((n, n, n), first[2] - x0)).to_numpy()      # r0 = b whatever x0 is (src/GCR.h:189): x - x0 solves A y = b
ok = ok and abs(np.linalg.norm(r) / np.linalg.norm(b.to_numpy()) - first[3][-1]) <= 1e-6 * first[3][-1]
print("FIRST-CALL-RIGHT" if ok else "FIRST-CALL-WRONG")
'''
    env = dict(os.environ, MGCR_TEST_STEPBUILD_STALL="7", MGCR_TEST_RESIDENT_SPIN_LIMIT="20000")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180,
                         cwd=os.path.join(os.path.dirname(__file__), ".."))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "LIBRARY-USABLE" in out.stdout and "FIRST-CALL-RIGHT" in out.stdout, out.stdout
    import re
    m = re.search(r"fallbacks (\d+) launches (\d+) then (\d+)", out.stdout)
    assert m and int(m.group(1)) == 1 and int(m.group(2)) >= 1 and int(m.group(3)) == 0, out.stdout


def test_vcycle_with_one_launch_smoother_steps_is_bit_identical():
    """inside a V-cycle at 128^3 the level-0 smoothers (2 sweeps: the first one an in-cycle step, the second the solve's last,
    with the V-cycle's deferred-residual / pending-x hand-overs) and the flexible outer solve: same cycle output and same outer
    history with the one-launch steps on and off"""
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import problems
    n = 128
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    dims = (n, n, n)
    out = []
    for on in (1, 0):
        prev = mg.set_option("step_build", on)
        try:
            A = mg.Sparse(N, ncol, rowptr, col, val)
            prm = mg.MG_Param(mg.Mesh(dims), 2, 1, None, mg.GCR(mg.GCR_Param(0, 10, 50, 1e-2, False)),
                              mg.GCR(mg.GCR_Param(0, 10, 2, 1e-30, False)), 2, None, None, null_vectors=np.ones((1, N), np.complex128))
            M = mg.MG(A, prm)
            b = mg.Field(dims).fill_rhs(4)
            before = mg.stat("step_build_launches")
            y = M(b).to_numpy()
            took = mg.stat("step_build_launches") - before
            outer = mg.GCR(A, mg.GCR_Param(0, 5, 30, 1e-8, False, None, M, flexible=True))
            x = mg.Field(dims).set_zero()
            outer.solve(b, x)
            out.append((took, y, outer.last_history.copy(), x.to_numpy()))
            del M, A
        finally:
            mg.set_option("step_build", prev)
    assert out[0][0] >= 2 and out[1][0] == 0
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    assert out[0][2][-1] <= 1e-8
