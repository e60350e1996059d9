"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors of the real
reference, against the CPU oracle on seeded inputs, and through size-independent properties at
the full BASELINE sizes.  Tolerances (stated here):

  * SpMV / element-wise Field algebra: component-wise |diff| <= 1e-13 * row scale (the kernels
    keep the reference's operation order; only multi-lane rows re-associate);
  * dots / norms: relative 1e-12 (reduction order differs from the reference's sequential sum);
  * GCR residual history h(k) = |r_k| / |b|:
        |h_gpu(k) - h_ref(k)| <= max(1e-9 * h_ref(k), 8 * s(k)) + 1e-17
    where s(k) is the REFERENCE ALGORITHM'S OWN re-association sensitivity for that solve: how far
    its history moves when its dot products are summed in reverse or pairwise order instead of
    index order (oracle.gcr_reorder_sensitivity; the sequential-order oracle is pinned bit for bit
    to the real reference).  The only thing the GPU path does differently from the reference is
    the order in which dot products are summed, and several of these recurrences amplify that
    (truncated GCR on Poisson 8^3: s(k)/h(k) reaches 0.4 at the last step; restart-2 on the 4x4
    sample 3e-5), so a fixed relative bound would be either meaningless or unattainable by ANY
    parallel sum.  Where s(k) is tiny the bound is nine digits.  Iterations to convergence: +-1.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("mgpreconditionedgcr_amd")
from mgpreconditionedgcr_amd import (Dense, DiracOp, Field, GCR, GCR_Param, HierarchicalSparse, MgcrError,  # noqa: E402
                                     Sparse, problems, read_data)
from oracle import oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module", autouse=True)
def _init():
    mg.init()  # raises loudly when no GPU / no library: there is no fallback
    yield


@pytest.fixture(params=["multi-kernel", "one-workgroup"], autouse=True)
def solver_path(request):
    """Every test runs twice: with the one-workgroup small-system solver (gcr_small.hip) switched
    off, so that small test systems exercise the fused multi-kernel loop that large systems use, and
    with it forced on for systems up to 16384 rows (default: 1024)."""
    mg.lib().mgcr_set_small_solve_rows(0 if request.param == "multi-kernel" else 16384)
    yield request.param
    mg.lib().mgcr_set_small_solve_rows(1024)


def its_close(it, it_ref, its_range=None):
    """Iterations to convergence: +-1, widened by the spread the reference algorithm itself shows
    under re-association of its dot products (see module docstring)."""
    lo, hi = (it_ref, it_ref) if its_range is None else its_range
    slack = max(1, 3 * (hi - lo))  # two alternative orders only sample the spread: allow three times what they show
    return lo - slack <= it <= hi + slack


def x_close(x, A, rhs, gcr, xo, sens):
    """Solution check.  Always: the recurrence residual the solver reports equals the true residual
    b - A x (size-independent property).  Additionally, where the solve is well conditioned with
    respect to re-association (sensitivity at the last step < 1e-12), x equals the oracle's."""
    b = rhs.to_numpy()
    r = rhs - A(x)
    hl = gcr.last_history[-1]
    assert abs(r.norm() / np.linalg.norm(b) - hl) <= 1e-6 * hl + 1e-13
    if xo is not None and sens is not None and sens.size == gcr.last_history.size and sens[-1] < 1e-12 * max(hl, 1e-300) * 1e3:
        assert np.abs(x.to_numpy() - xo).max() <= 1e-8 * np.abs(xo).max()


OBSERVED = {}   # golden tag -> observed deviation of the GPU history from the reference's (written out at module teardown)


def record_deviation(tag, path, h, ref):
    """Observed deviation per golden, committed as tests/golden/observed_r02.json: max relative deviation over the steps
    with h_ref >= 1e-9 and over those below, and the step at which the maximum occurs."""
    n = min(h.size, ref.size)
    rel = np.abs(h[1:n] - ref[1:n]) / ref[1:n]
    hi = ref[1:n] >= 1e-9
    k = int(np.argmax(rel)) + 1 if n > 1 else 0
    OBSERVED.setdefault(tag, {})[path] = {
        "steps_compared": int(n - 1), "iterations_gpu": int(h.size - 1), "iterations_reference": int(ref.size - 1),
        "max_rel_dev_where_ref_ge_1e-9": float(rel[hi].max()) if hi.any() else 0.0,
        "max_rel_dev_where_ref_lt_1e-9": float(rel[~hi].max()) if (~hi).any() else 0.0,
        "step_of_max": k, "ref_at_step_of_max": float(ref[k]) if n > 1 else None}


def fixed_bound_ok(h, ref, hi=1e-9, lo=1e-6):
    """SURVEY.md §8(c), the FIXED bound: |h_gpu(k) - h_ref(k)| / h_ref(k) <= 1e-9 while h_ref(k) >= 1e-9, <= 1e-6 below
    that.  Returns (ok, first offending step)."""
    n = min(h.size, ref.size)
    for k in range(1, n):
        if abs(h[k] - ref[k]) > (hi if ref[k] >= 1e-9 else lo) * ref[k]:
            return False, k
    return True, None


def hist_close(h, ref, what="", sens=None, path=None):
    n = min(h.size, ref.size)
    if path is not None:
        record_deviation(what, path, h, ref)
    for k in range(1, n):
        s = sens[k] if sens is not None and k < sens.size else 0.0
        tol = max(1e-9 * ref[k], 8 * s) + 1e-17
        assert abs(h[k] - ref[k]) <= tol, "%s step %d: %.12e vs ref %.12e (sensitivity %.2e)" % (what, k, h[k], ref[k], s)


@pytest.fixture(scope="module", autouse=True)
def _write_observed():
    yield
    if OBSERVED:
        import json
        import os
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "observed_deviation.json"), "w") as f:
            json.dump(OBSERVED, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module")
def sample_oracle(sample_matrix_path):
    nrow, ncol, rowptr, col, val = orc.read_text_csr(sample_matrix_path)
    return orc.csr(nrow, ncol, rowptr, col, val)


@pytest.fixture(scope="module")
def sample(sample_matrix_path):
    import os
    D = read_data(os.path.basename(sample_matrix_path), directory=os.path.dirname(sample_matrix_path))
    return D


DIMS = (4, 4, 4, 4, 4, 3)


def test_read_data_and_layout(sample):
    assert sample.get_dim() == 3072 and sample.get_nrow() == 3072 and sample.get_nnz() == 119808
    sb = sample.stored_bytes()
    assert sb["tail_nnz"] == 0 and 39 <= sb["ell_width"] <= 40  # perfectly regular: 39 nnz / row


def test_g1_spmv(sample, sample_gold):
    g = sample_gold
    x = Field(DIMS, g["g1_x"])
    y = sample(x).to_numpy()
    scale = np.abs(g["g1_Dx"]).max()
    assert np.abs(y - g["g1_Dx"]).max() <= 1e-13 * scale
    dirac = DiracOp(sample, 0.15)
    y = dirac(x).to_numpy()
    assert np.abs(y - g["g1_dirac_x"]).max() <= 1e-13 * np.abs(g["g1_dirac_x"]).max()


def test_g2_blas1(sample_gold):
    g = sample_gold
    a, b = Field(DIMS, g["g2_a"]), Field(DIMS, g["g2_b"])
    dot_ab, norms, alpha = g["g2_scalars"]
    assert abs(a.dot(b) - dot_ab) <= 1e-12 * abs(dot_ab)
    assert abs(a.squarednorm() - norms.real) <= 1e-12 * norms.real
    assert abs(b.squarednorm() - norms.imag) <= 1e-12 * norms.imag
    # element-wise: same operation order and no FMA contraction => identical bits
    assert np.array_equal(a.add_scaled(alpha, b).to_numpy(), g["g2_a_plus_alpha_b"])
    assert np.array_equal(a.add_scaled(-alpha, b).to_numpy(), g["g2_a_minus_alpha_b"])
    c = a.copy()
    c += b
    assert np.array_equal(c.to_numpy(), g["g2_a"] + g["g2_b"])


GCR_CASES = [
    ("g3_restart5", dict(re=5, max_it=4000, tau=1e-13), 118),
    ("g4_restart2", dict(re=2, max_it=4000, tau=1e-13), 116),
    ("g5_trunc8", dict(trunc=8, max_it=300, tau=1e-3), 46),
    ("g10_maxiter0", dict(re=10, max_it=0, tau=1e-8), 1),
]


def _okw(kw):
    m = dict(re="restart", trunc="truncation", max_it="max_iter", tau="tol")
    return {m[k]: v for k, v in kw.items()}


@pytest.mark.parametrize("tag,kw,iters", GCR_CASES)
def test_gcr_history_vs_reference(sample, sample_oracle, sample_gold, tag, kw, iters, solver_path):
    g = sample_gold
    dirac = DiracOp(sample, 0.15)
    rhs = Field(DIMS, g["gcr_rhs"])
    x = Field(DIMS).set_zero()
    gcr = GCR(dirac, GCR_Param(verb=False, **kw))
    gcr.solve(rhs, x)
    ref, sens, rng = orc.gcr_reorder_sensitivity(orc.dirac(sample_oracle, 0.15), orc.gcr_param(**_okw(kw)), g["gcr_rhs"])
    assert np.array_equal(ref[1:], g[tag + "_hist"][1:])  # the oracle IS the reference here
    assert its_close(gcr.last_iterations, iters, rng)
    hist_close(gcr.last_history, g[tag + "_hist"], tag, sens, path=solver_path)
    if tag in ("g3_restart5", "g4_restart2"):
        # The headline goldens additionally under a FIXED bound and +-1 iteration.  G3 (restart 5): SURVEY.md §8(c)'s own
        # bound, 1e-9 relative while h_ref >= 1e-9 and 1e-6 below (observed: 1.1e-11 / 5.5e-9, tests/golden/observed_r02.json).
        # G4 (restart 2) cannot meet it with ANY parallel sum: the reference's own history moves by up to 2e-4 relative at
        # step 112 when its dot products are merely summed in reverse or pairwise order (`sens`); observed here 9.5e-9 while
        # h_ref >= 1e-9 (first above 1e-9 at step 69) and 2.0e-5 below.  Its fixed bound is therefore 1e-7 / 1e-4.
        hi, lo = (1e-9, 1e-6) if tag == "g3_restart5" else (1e-7, 1e-4)
        ok, k = fixed_bound_ok(gcr.last_history, g[tag + "_hist"], hi, lo)
        assert ok, "%s: fixed bound broken at step %s" % (tag, k)
        assert abs(gcr.last_iterations - iters) <= 1
    x_close(x, dirac, rhs, gcr, g[tag + "_x"] if gcr.last_iterations == iters else None, sens)


def test_gcr_full_mode_prefix(sample, sample_gold):
    """Full GCR (no restart, no truncation) stalls on the non-Hermitian sample because of the
    reference's conjugation order and amplifies rounding; pin 60 steps."""
    g = sample_gold
    dirac = DiracOp(sample, 0.15)
    gcr = GCR(dirac, GCR_Param(0, 0, 60, 1e-13, False))
    x = Field(DIMS).set_zero()
    gcr.solve(Field(DIMS, g["gcr_rhs"]), x)
    assert gcr.last_iterations == 60 and not gcr.last_converged
    hist_close(gcr.last_history, g["g6_full_hist"][:61], "full")


def test_gcr_check_every_does_not_change_results(sample, sample_gold):
    g = sample_gold
    dirac = DiracOp(sample, 0.15)
    rhs = Field(DIMS, g["gcr_rhs"])
    res = []
    for ce in (1, 7, 50):
        x = Field(DIMS).set_zero()
        gcr = GCR(dirac, GCR_Param(0, 5, 4000, 1e-13, False, check_every=ce))
        gcr.solve(rhs, x)
        res.append((gcr.last_iterations, gcr.last_history.copy(), x.to_numpy()))
    for r in res[1:]:
        assert r[0] == res[0][0] and np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])


def test_gcr_complex_k_and_x0_quirk(sample, sample_gold):
    g = sample_gold
    rhs = Field(DIMS, g["gcr_rhs"])
    x = Field(DIMS).set_zero()
    gcr = GCR(DiracOp(sample, 0.12 + 0.05j), GCR_Param(0, 5, 40, 1e-13, False))
    gcr.solve(rhs, x)
    hist_close(gcr.last_history, g["g3b_complexk_hist"], "complex k")
    # r0 = b whatever x0 is (src/GCR.h:189): history independent of x0, x_final = x0 + x(zero start)
    dirac = DiracOp(sample, 0.15)
    x0 = problems.rhs_grid(3072, 11)
    xa, xb = Field(DIMS, x0), Field(DIMS).set_zero()
    ga, gb = GCR(dirac, GCR_Param(0, 5, 20, 1e-13, False)), GCR(dirac, GCR_Param(0, 5, 20, 1e-13, False))
    ga.solve(rhs, xa)
    gb.solve(rhs, xb)
    assert np.array_equal(ga.last_history, gb.last_history)
    hist_close(ga.last_history, g["g10_x0rand_hist"], "x0 quirk")
    assert np.abs(xa.to_numpy() - x0 - xb.to_numpy()).max() <= 1e-12


def test_precond_hooks_literal(sample, sample_gold):
    """r = M(r) / Ar = Ml(Ar) literal hooks (src/GCR.h:197-204,236-247) with M = 1 + kD."""
    g = sample_gold
    dirac = DiracOp(sample, 0.15)
    M = DiracOp(sample, -0.15)
    rhs = Field(DIMS, g["gcr_rhs"])
    x = Field(DIMS).set_zero()
    gcr = GCR(dirac, GCR_Param(0, 5, 20, 1e-13, False, None, M))
    gcr.solve(rhs, x)
    # this (reference-literal) right-preconditioned recurrence diverges: compare loosely in the tail
    ref = g["g11_right_neumann_hist"]
    assert np.allclose(gcr.last_history[1:], ref[1:], rtol=1e-8)
    x = Field(DIMS).set_zero()
    gcr = GCR(dirac, GCR_Param(0, 5, 60, 1e-13, False, M, None))
    gcr.solve(rhs, x)
    hist_close(gcr.last_history, g["g11_left_neumann_hist"], "left precond")


def test_poisson_goldens(poisson_gold, solver_path):
    g = poisson_gold
    for n, kw, tag in [(32, dict(re=5, max_it=10, tau=1e-13), "p32"),
                       (8, dict(trunc=4, max_it=300, tau=1e-10), "p8_trunc4"),
                       (8, dict(max_it=25, tau=1e-10), "p8_full"),
                       (16, dict(re=3, max_it=300, tau=1e-12), "p16_restart3")]:
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        A = Sparse(N, ncol, rowptr, col, val)
        assert A.stored_bytes()["ell_width"] == 7
        rhs = Field((n, n, n)).fill_rhs(0)
        assert np.array_equal(rhs.to_numpy(), problems.rhs_grid(N, 0))
        x = Field((n, n, n)).set_zero()
        gcr = GCR(A, GCR_Param(verb=False, **kw))
        gcr.solve(rhs, x)
        ref = g[tag + "_hist"]
        oref, sens, rng = orc.gcr_reorder_sensitivity(orc.csr(N, ncol, rowptr, col, val), orc.gcr_param(**_okw(kw)),
                                                      problems.rhs_grid(N, 0))
        assert np.array_equal(oref[1:], ref[1:])
        assert its_close(gcr.last_iterations, ref.size - 1, rng), (tag, gcr.last_iterations, ref.size - 1, rng)
        hist_close(gcr.last_history, ref, tag, sens, path=solver_path)
        if tag == "p32":
            ok, k = fixed_bound_ok(gcr.last_history, ref)
            assert ok, "%s: fixed bound broken at step %s" % (tag, k)
            assert gcr.last_iterations == ref.size - 1
        x_close(x, A, rhs, gcr, g.get(tag + "_x") if gcr.last_iterations == ref.size - 1 else None, sens)


def test_poisson128_first_steps_vs_reference(poisson_gold, solver_path):
    """BASELINE config 2 matrix: first 10 steps against the real reference run on the same RHS."""
    n = 128
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    A = Sparse(N, ncol, rowptr, col, val)
    del rowptr, col, val
    rhs = Field((n, n, n)).fill_rhs(0)
    x = Field((n, n, n)).set_zero()
    gcr = GCR(A, GCR_Param(0, 5, 10, 1e-13, False))
    gcr.solve(rhs, x)
    hist_close(gcr.last_history, poisson_gold["p128_hist"], "p128", path=solver_path)
    ok, k = fixed_bound_ok(gcr.last_history, poisson_gold["p128_hist"])
    assert ok, "p128: fixed bound (SURVEY 8(c)) broken at step %s" % k
    # size-independent property: the recurrence residual equals the true residual b - A x
    r = rhs - A(x)
    assert abs(r.norm() / rhs.norm() - gcr.last_history[-1]) <= 1e-10 * gcr.last_history[-1]


def test_g8_hsparse(hsparse_gold):
    g = hsparse_gold
    nb, bs = int(g["nb"]), int(g["bs"])
    H = HierarchicalSparse(nb, nb, g["rows"], g["cols"], g["blocks"])
    assert H.get_dim() == nb * bs
    y = H(Field((nb * bs,), g["x"])).to_numpy()
    assert np.abs(y - g["y"]).max() <= 1e-13 * np.abs(g["y"]).max()


@pytest.mark.parametrize("bs,nb", [(1, 50), (3, 40), (20, 64), (33, 10), (64, 5), (70, 3)])
def test_block_csr_vs_oracle(bs, nb):
    rng = np.random.default_rng(bs * 100 + nb)
    rows, cols = [], []
    for r in range(nb):
        k = int(rng.integers(1, min(nb, 9) + 1))
        cs = rng.choice(nb, size=k, replace=True)  # duplicates allowed (kept and summed at apply time)
        rows += [r] * k
        cols += list(cs)
    perm = rng.permutation(len(rows))
    rows, cols = np.array(rows, np.int32)[perm], np.array(cols, np.int32)[perm]
    blocks = rng.uniform(-1, 1, (rows.size, bs, bs)) + 1j * rng.uniform(-1, 1, (rows.size, bs, bs))
    x = problems.rhs_grid(nb * bs, 5)
    H = HierarchicalSparse(nb, nb, rows, cols, blocks)
    ref = orc.bcsr_from_triplets(nb, nb, bs, rows, cols, blocks)(x)
    y = H(Field((nb * bs,), x)).to_numpy()
    # same stable sort and same per-row / per-block order as the oracle => identical bits
    assert np.array_equal(y, ref)


@pytest.mark.parametrize("bs,nb", [(9, 41), (12, 41), (16, 41), (18, 41), (20, 41), (23, 41), (27, 41), (32, 41), (4, 1500), (7, 1100)])
def test_block_csr_more_shapes_vs_oracle(bs, nb):
    """Block sizes with 2 .. 16 loads of 64 entries per block, block rows with 0, 1 and up to 12 blocks, duplicates; and, from
    1024 block rows on, the wave kernels are dealt the rows longest first (spmv.hip bcsr_build_device): every row is still
    summed by one wave in its stored order — the bits of the oracle (src/HierarchicalSparse.h:101-161)."""
    rng = np.random.default_rng(1000 + bs)
    rows, cols = [], []
    for r in range(nb):
        k = 0 if r % 7 == 3 else 1 if r % 7 == 5 else int(rng.integers(2, 13))
        cs = rng.choice(nb, size=k, replace=True)
        rows += [r] * k
        cols += list(cs)
    perm = rng.permutation(len(rows))
    rows, cols = np.array(rows, np.int32)[perm], np.array(cols, np.int32)[perm]
    blocks = rng.uniform(-1, 1, (rows.size, bs, bs)) + 1j * rng.uniform(-1, 1, (rows.size, bs, bs))
    x = problems.rhs_grid(nb * bs, 7)
    H = HierarchicalSparse(nb, nb, rows, cols, blocks)
    ref = orc.bcsr_from_triplets(nb, nb, bs, rows, cols, blocks)(x)
    xf = Field((nb * bs,), x)
    assert np.array_equal(H(xf).to_numpy(), ref)


@pytest.mark.parametrize("nrow,ncol,kw", [
    (1, 1, dict(min_len=1, max_len=1)),
    (257, 300, dict(min_len=0, max_len=9)),             # empty rows, ragged
    (5000, 5000, dict(min_len=1, max_len=7)),
    (3000, 2500, dict(min_len=0, max_len=6, long_rows=5, long_len=900)),   # CSR tail
    (700, 700, dict(min_len=30, max_len=45)),           # multi-lane rows
    (64, 4096, dict(min_len=1000, max_len=1500)),       # everything long
])
def test_spmv_vs_oracle_irregular(nrow, ncol, kw):
    rng = np.random.default_rng(nrow * 7 + ncol)
    rowptr, col, val = problems.random_csr(nrow, ncol, rng, **kw)
    x = problems.rhs_grid(ncol, 3)
    ref = orc.csr(nrow, ncol, rowptr, col, val)(x)
    A = Sparse(nrow, ncol, rowptr, col, val)
    y = A(Field((ncol,), x)).to_numpy()
    scale = max(np.abs(ref).max(), 1.0)
    assert np.abs(y - ref).max() <= 1e-13 * scale * max(1, kw.get("max_len", 1) // 8)
    if nrow == ncol:
        k = 0.3 - 0.2j
        refd = orc.dirac(orc.csr(nrow, ncol, rowptr, col, val), k)(x)
        yd = DiracOp(A, k)(Field((ncol,), x)).to_numpy()
        assert np.abs(yd - refd).max() <= 1e-13 * max(np.abs(refd).max(), 1.0) * max(1, kw.get("max_len", 1) // 8)


def test_spmv_bit_exact_when_one_thread_per_row():
    """L = 1 layout keeps the reference's per-row summation order (src/Operator.h:338-341)."""
    n = 24
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    val = val * (1.0 + 0.25j)
    x = problems.rhs_grid(N, 9)
    ref = orc.csr(N, ncol, rowptr, col, val)(x)
    y = Sparse(N, ncol, rowptr, col, val)(Field((n, n, n), x)).to_numpy()
    assert np.array_equal(y, ref)


@pytest.mark.parametrize("case", ["real-constant", "complex-constant", "random-values", "random-columns"])
def test_row_pattern_storage_is_bit_exact(case):
    """Sparse matrices of >= 2^15 rows whose rows repeat a few (column - row, value) patterns are stored
    as a 2-byte pattern id per row (format 1), or — values differing from row to row — with the columns
    in the dictionary and the values in a slab (format 2).  Either way the SpMV multiplies and adds in
    CSR order: y has the bits of the reference's row loop (src/Operator.h:338-341), with and without
    the DiracOp epilogue, and the same bits as the plain ELL slab."""
    n = 40  # 64000 rows
    rng = np.random.default_rng(11)
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    want = 1
    if case == "complex-constant":
        val = val * (0.75 - 0.5j)
    elif case == "random-values":
        val = rng.standard_normal(val.size) + 1j * rng.standard_normal(val.size)
        want = 2
    elif case == "random-columns":
        # sorted random columns per row: nothing repeats, the dictionary overflows and the slab stays
        col = np.sort(rng.integers(0, ncol, size=(N, 5)), axis=1).ravel()
        rowptr = np.arange(N + 1, dtype=np.int64) * 5
        val = rng.standard_normal(col.size) + 1j * rng.standard_normal(col.size)
        want = 0
    x = problems.rhs_grid(N, 9)
    O = orc.csr(N, ncol, rowptr, col, val)
    ref = O(x)
    xf = Field((N,), x)
    k = 0.3 - 0.2j
    # constant-coefficient stencils additionally get the stencil view (format 3: 7 slots, presence words per wave,
    # csrc/spmv.hip sten_try), which the apply kernels then read instead of the dictionary: both must give the bits
    for stencil in (1, 0):
        prev = mg.set_option("stencil_storage", stencil)
        try:
            A = Sparse(N, ncol, rowptr, col, val)
            fmt, npat = A.storage_format()
            if stencil and want == 1:
                assert (fmt, npat) == (3, 7), (fmt, npat)
            else:
                assert fmt == want, (fmt, npat)
                if want == 1:
                    assert npat == 27  # 3 boundary classes per axis
            y = A(xf).to_numpy()
            assert np.array_equal(y, ref)
            assert np.array_equal(DiracOp(A, k)(xf).to_numpy(), orc.dirac(O, k)(x))
        finally:
            mg.set_option("stencil_storage", prev)
    prev = mg.set_option("pattern_storage", 0)
    try:
        B = Sparse(N, ncol, rowptr, col, val)
    finally:
        mg.set_option("pattern_storage", prev)
    assert B.storage_format() == (0, 0)
    assert np.array_equal(B(xf).to_numpy(), y)
    if want:
        assert A.stored_bytes()["matrix_bytes"] < B.stored_bytes()["matrix_bytes"]


def test_row_pattern_storage_same_solve():
    """GCR on the pattern-dictionary operator and on the plain slab: identical histories and x."""
    n = 40
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    b = Field((n, n, n)).fill_rhs(3)
    out = []
    for on, stencil in ((1, 1), (1, 0), (0, 0)):   # stencil view | dictionary | plain slab
        prev = mg.set_option("pattern_storage", on)
        prev_s = mg.set_option("stencil_storage", stencil)
        try:
            A = Sparse(N, ncol, rowptr, col, val)
            g = GCR(A, GCR_Param(0, 5, 60, 1e-12, False))
            x = Field((n, n, n)).set_zero()
            g.solve(b, x)
            out.append((A.storage_format()[0], g.last_history.copy(), x.to_numpy()))
        finally:
            mg.set_option("pattern_storage", prev)
            mg.set_option("stencil_storage", prev_s)
    assert [o[0] for o in out] == [3, 1, 0]
    for o in out[1:]:
        assert np.array_equal(out[0][1], o[1]) and np.array_equal(out[0][2], o[2])


@pytest.mark.parametrize("kind,restart,tol", [("poisson", 5, 1e-10), ("poisson", 1, 1e-6), ("poisson", 8, 1e-10),
                                              ("poisson", 9, 1e-10), ("poisson", 10, 1e-10), ("poisson", 16, 1e-10),
                                              ("dirac", 5, 1e-12), ("dirac", 3, 1e-12), ("dirac", 12, 1e-12),
                                              ("flex", 4, 1e-10), ("flex", 11, 1e-10)])
def test_lean_restart_cycles_match_classic(kind, restart, tol, sample, sample_oracle):
    """Restart-mode GCR keeps, inside a cycle, the residuals the directions were started from instead of
    the directions (gcr.hip header).  r, Ap and every scalar follow the same recurrences: the residual
    history has the same bits as with the classic kernels; x is the same linear combination summed in a
    different order: it differs by rounding, and its TRUE residual is as good."""
    if kind == "dirac":
        dims = DIMS
        n = sample.get_dim()
        A = DiracOp(sample, 0.18)
        b = Field(dims, problems.rhs_grid(n, 5))
        M = None
    else:
        n1 = 24
        n, ncol, rowptr, col, val = problems.poisson3d_csr(n1)
        dims = (n1, n1, n1)
        A = Sparse(n, ncol, rowptr, col, val * (1.0 + 0.125j))
        b = Field(dims, problems.rhs_grid(n, 5))
        # the preconditioner runs in truncated mode, which has no lean variant: identical in both passes
        M = GCR(A, GCR_Param(3, 0, 3, 1e-30, False)) if kind == "flex" else None
    out = []
    for on in (1, 0):
        prev = mg.set_option("lean_cycles", on)
        try:
            g = GCR(A, GCR_Param(0, restart, 400, tol, False, None, M, flexible=M is not None))
            x = Field(dims).set_zero()
            g.solve(b, x)
        finally:
            mg.set_option("lean_cycles", prev)
        true_res = (b - A(x)).norm() / b.norm()
        out.append((g.last_history.copy(), x.to_numpy(), g.last_iterations, true_res))
    (h1, x1, it1, t1), (h0, x0, it0, t0) = out
    assert it1 == it0 and np.array_equal(h1, h0)
    assert np.abs(x1 - x0).max() <= 1e-12 * np.abs(x0).max()
    assert t1 <= max(2.0 * t0, 1.05 * tol)


@pytest.mark.parametrize("fmt", ["pattern", "slab", "dirac-pattern", "slab-generic-width"])
@pytest.mark.parametrize("mode", [dict(restart=5), dict(restart=10), dict(truncation=3), dict()])
def test_fused_apply_and_dots_same_bits(fmt, mode):
    """SpMV + beta dot products as one kernel (spmv.hip: spmv_multidot_kernel) against the two separate
    kernels: same y, same partial sums, hence the same residual history and the same x, bit for bit —
    for every storage format, with the DiracOp epilogue, in restart / truncated / full mode."""
    n1 = 40 if "pattern" in fmt else 20
    n, ncol, rowptr, col, val = problems.poisson3d_csr(n1)
    val = val * (1.0 - 0.25j)
    if fmt == "slab-generic-width":   # rows of up to 5 entries: the run-time-width code path
        keep = np.ones(val.size, bool)
        keep[rowptr[:-1]] = np.diff(rowptr) < 6
        keep[rowptr[1:] - 1] = np.diff(rowptr) < 7
        newptr = np.concatenate([[0], np.cumsum(np.add.reduceat(keep, rowptr[:-1]))]).astype(np.int64)
        rowptr, col, val = newptr, col[keep], val[keep]
        val[rowptr[:-1] + (np.diff(rowptr) // 2)] += 7.0  # keep it diagonally dominant enough to converge
    prev = mg.set_option("pattern_storage", 1 if "pattern" in fmt else 0)
    try:
        A = Sparse(n, ncol, rowptr, col, val)
    finally:
        mg.set_option("pattern_storage", prev)
    assert (A.storage_format()[0] != 0) == ("pattern" in fmt)
    op = DiracOp(A, 0.05 + 0.02j) if fmt.startswith("dirac") else A
    b = Field((n,), problems.rhs_grid(n, 8))
    out = []
    for on in (1, 0):
        prev = mg.set_option("fused_apply", on)
        try:
            g = GCR(op, GCR_Param(mode.get("truncation", 0), mode.get("restart", 0), 40, 1e-11, False))
            x = Field((n,)).set_zero()
            g.solve(b, x)
        finally:
            mg.set_option("fused_apply", prev)
        out.append((g.last_history.copy(), x.to_numpy()))
    assert out[0][0].size > 5
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # smoother-shaped solves (fewer iterations than restart slots): step 0 — Ap_0 = A r_0 with <r_0,Ap_0>, <Ap_0,Ap_0>,
    # |r_0|^2 and |b|^2 — is one kernel too (gcr_fused.hip init_apply_kernel), from x0 = 0 and from a given x0
    if "restart" in mode:
        x0 = problems.rhs_grid(n, 4) * 0.1
        for use_x0 in (False, True):
            out = []
            for on in (1, 0):
                prev = mg.set_option("fused_apply", on)
                try:
                    g = GCR(op, GCR_Param(0, 10, 3, 1e-30, False, use_x0=use_x0))
                    x = Field((n,), x0) if use_x0 else Field((n,)).set_zero()
                    g.solve(b, x)
                finally:
                    mg.set_option("fused_apply", prev)
                out.append((g.last_history.copy(), x.to_numpy()))
            assert out[0][0].size == 4
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_graph_replay_same_results():
    """Restart cycles captured in a hipGraph and replayed (opt-in) enqueue the same kernels with the same
    arguments: history and x are bit-identical to the eager launches, for lean cycles of 5 and 10 slots."""
    n1 = 20
    n, ncol, rowptr, col, val = problems.poisson3d_csr(n1)
    A = Sparse(n, ncol, rowptr, col, val * (1.0 + 0.25j))
    b = Field((n,), problems.rhs_grid(n, 2))
    for restart in (5, 10):
        out = []
        for on in (1, 0):
            prev = mg.set_option("graph_replay", on)
            try:
                g = GCR(A, GCR_Param(0, restart, 73, 1e-30, False, check_every=7))
                x = Field((n,)).set_zero()
                g.solve(b, x)
            finally:
                mg.set_option("graph_replay", prev)
            out.append((g.last_history.copy(), x.to_numpy()))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("dim", [1, 4, 20, 64, 100])
def test_dense_operator_vs_oracle(dim):
    """Dense::operator() (src/Operator.h:159-173): every row accumulated in column order — the bits of the
    oracle's restatement (a block-CSR operator of one block)."""
    rng = np.random.default_rng(dim)
    m = rng.standard_normal((dim, dim)) + 1j * rng.standard_normal((dim, dim))
    x = problems.rhs_grid(dim, 3)
    ref = orc.bcsr_from_triplets(1, 1, dim, [0], [0], m.reshape(1, dim, dim))(x)
    y = Dense(m, dim)(Field((dim,), x)).to_numpy()
    assert np.array_equal(y, ref)
    # and the plain definition, to rounding
    assert np.abs(y - m @ x).max() <= 1e-13 * max(np.abs(m @ x).max(), 1.0) * dim


def test_gcr_vs_oracle_random_nonhermitian():
    """Seeded diagonally dominant complex matrix, every mode, against the CPU oracle."""
    rng = np.random.default_rng(42)
    n = 4000
    rowptr, col, val = problems.random_csr(n, n, rng, min_len=3, max_len=10)
    # make it diagonally dominant: add a diagonal entry larger than the row sum
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    rowsum = np.bincount(rows, weights=np.abs(val), minlength=n)
    newptr = rowptr + np.arange(n + 1)
    ncol_arr, nval = np.empty(newptr[-1], np.int64), np.empty(newptr[-1], np.complex128)
    for r in range(n):
        s, e = rowptr[r], rowptr[r + 1]
        ncol_arr[newptr[r]:newptr[r] + (e - s)] = col[s:e]
        nval[newptr[r]:newptr[r] + (e - s)] = val[s:e]
        ncol_arr[newptr[r + 1] - 1] = r
        nval[newptr[r + 1] - 1] = 2.0 * rowsum[r] + 1.0
    b = problems.rhs_grid(n, 1)
    Ao = orc.csr(n, n, newptr, ncol_arr, nval)
    A = Sparse(n, n, newptr, ncol_arr, nval)
    for kw_o, kw_g in [(dict(restart=4, max_iter=200, tol=1e-11), dict(re=4, max_it=200, tau=1e-11)),
                       (dict(truncation=3, max_iter=200, tol=1e-11), dict(trunc=3, max_it=200, tau=1e-11)),
                       (dict(truncation=11, max_iter=200, tol=1e-11), dict(trunc=11, max_it=200, tau=1e-11)),
                       (dict(max_iter=30, tol=1e-11), dict(max_it=30, tau=1e-11))]:
        xo, ho, ito, _ = orc.gcr_solve(Ao, orc.gcr_param(**kw_o), b)
        _, sens, rng = orc.gcr_reorder_sensitivity(Ao, orc.gcr_param(**kw_o), b)
        x = Field((n,)).set_zero()
        gcr = GCR(A, GCR_Param(verb=False, **kw_g))
        fb = Field((n,), b)
        gcr.solve(fb, x)
        assert its_close(gcr.last_iterations, ito, rng), (kw_g, gcr.last_iterations, ito, rng)
        hist_close(gcr.last_history, ho, str(kw_g), sens)
        x_close(x, A, fb, gcr, xo if gcr.last_iterations == ito else None, sens)


def test_gcr_as_operator_and_flexible_precond():
    """GCR used as an Operator (src/GCR.h:62-68) with x0 = 0, nested as a flexible right
    preconditioner — against the same construction in the oracle."""
    n = 12
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    b = problems.rhs_grid(N, 2)
    Ao = orc.csr(N, ncol, rowptr, col, val)
    inner_o = orc.gcr_op(Ao, orc.gcr_param(restart=4, max_iter=4, tol=1e-30), x0_mode=1)
    xo, ho, ito, _ = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=60, tol=1e-10, right=inner_o, flexible=True), b)
    A = Sparse(N, ncol, rowptr, col, val)
    inner = GCR(A, GCR_Param(0, 4, 4, 1e-30, False))
    y = inner(Field((n, n, n), b)).to_numpy()
    assert np.abs(y - inner_o(b)).max() <= 1e-12 * np.abs(y).max()
    outer = GCR(A, GCR_Param(0, 5, 60, 1e-10, False, None, inner, flexible=True))
    x = Field((n, n, n)).set_zero()
    outer.solve(Field((n, n, n), b), x)
    assert abs(outer.last_iterations - ito) <= 1 and outer.last_converged
    hist_close(outer.last_history, ho, "flexible")
    assert outer.last_iterations < 30  # the preconditioner helps
    r = Field((n, n, n), b) - A(x)
    assert r.norm() / np.linalg.norm(b) <= 2e-10


def test_use_x0_extension():
    n = 10
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    b, x0 = problems.rhs_grid(N, 2), problems.rhs_grid(N, 8)
    Ao = orc.csr(N, ncol, rowptr, col, val)
    xo, ho, ito, _ = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=100, tol=1e-10, use_x0=True), b, x0)
    _, sens, rng = orc.gcr_reorder_sensitivity(Ao, orc.gcr_param(restart=5, max_iter=100, tol=1e-10, use_x0=True), b, x0)
    A = Sparse(N, ncol, rowptr, col, val)
    x = Field((n, n, n), x0)
    gcr = GCR(A, GCR_Param(0, 5, 100, 1e-10, False, use_x0=True))
    gcr.solve(Field((n, n, n), b), x)
    assert its_close(gcr.last_iterations, ito, rng)
    hist_close(gcr.last_history, ho, "use_x0", sens)
    r = Field((n, n, n), b) - A(x)
    assert r.norm() / np.linalg.norm(b) <= 1.5e-10


def test_linearity_and_row_sums_full_size():
    """Size-independent checks at BASELINE config 3's matrix size (256^3, 117 M nnz)."""
    n = 256
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    A = Sparse(N, ncol, rowptr, col, val)
    del rowptr, col, val
    ones = Field((n, n, n)).set_constant(1.0)
    y = A(ones).to_numpy().reshape(n, n, n)
    # row sum = 6 - (#in-range neighbours): 0 in the interior, 1 per missing neighbour
    idx = np.arange(n)
    edge = ((idx == 0) | (idx == n - 1)).astype(np.float64)
    expect = edge[:, None, None] + edge[None, :, None] + edge[None, None, :]
    assert np.array_equal(y.real, expect) and not y.imag.any()
    del y
    a, b = Field((n, n, n)).fill_rhs(1), Field((n, n, n)).fill_rhs(2)
    alpha = 0.5 - 0.25j  # exactly representable: A(a + alpha b) == A a + alpha A b up to rounding
    lhs = A(a.add_scaled(alpha, b))
    rhs = A(a).add_scaled(alpha, A(b))
    d = lhs - rhs
    assert d.norm() <= 1e-14 * lhs.norm()


def test_error_paths(sample):
    with pytest.raises(MgcrError):
        sample(Field((10,)))  # Sparse matrix dimension does not match Field dimension!
    with pytest.raises(MgcrError):
        Field((4,)).assign(Field((5,)))  # Dimension mismatch.
    with pytest.raises(MgcrError):
        GCR(sample, GCR_Param(3, 3, 10, 1e-8, False))  # Do not support concurrent restarting and truncation.
    with pytest.raises(MgcrError):
        Sparse(2, 2, [0, 1, 2], [0, 5], [1.0, 1.0])  # column out of range
    with pytest.raises(MgcrError):
        mg.set_option("no_such_switch", 1)
    with pytest.raises(MgcrError):
        GCR(sample, GCR_Param(0, 5, 10, 1e-8, False)).storage_format()  # not a Sparse
    assert mg.set_option("lean_cycles", 1) in (0, 1) and mg.set_option("fused_apply", 1) in (0, 1)


# ------------------------------------------------------------------ edge cases
def test_degenerate_matrices():
    # no entries at all: y = 0
    A = Sparse(5, 7, np.zeros(6, np.int64), np.zeros(0, np.int64), np.zeros(0, np.complex128))
    y = A(Field((7,), problems.rhs_grid(7, 1))).to_numpy()
    assert y.shape == (5,) and not y.any()
    # 1 x 1
    A = Sparse(1, 1, [0, 1], [0], [2.5 - 1j])
    assert A(Field((1,), [1 + 1j])).to_numpy()[0] == (2.5 - 1j) * (1 + 1j)
    # only empty rows except the last one
    rowptr = np.zeros(1001, np.int64)
    rowptr[-1] = 3
    A = Sparse(1000, 1000, rowptr, [0, 500, 999], [1.0, 2.0, 3.0])
    x = problems.rhs_grid(1000, 2)
    y = A(Field((1000,), x)).to_numpy()
    assert not y[:-1].any() and y[-1] == 1.0 * x[0] + 2.0 * x[500] + 3.0 * x[999]
    # block-CSR with an empty block row and 1x1 blocks
    H = HierarchicalSparse(4, 4, [0, 2, 3, 3], [1, 2, 0, 3], np.array([2.0, 3.0, 4.0, 5.0]).reshape(4, 1, 1))
    x4 = np.array([1, 2, 3, 4], np.complex128)
    assert np.array_equal(H(Field((4,), x4)).to_numpy(), np.array([4, 0, 9, 24], np.complex128))


@pytest.mark.parametrize("kw_o,kw_g", [
    (dict(restart=1, max_iter=40, tol=1e-10), dict(re=1, max_it=40, tau=1e-10)),          # steepest-descent-like
    (dict(truncation=1, max_iter=40, tol=1e-10), dict(trunc=1, max_it=40, tau=1e-10)),
    (dict(restart=50, max_iter=7, tol=1e-30), dict(re=50, max_it=7, tau=1e-30)),           # restart never reached
    (dict(restart=3, max_iter=1, tol=1e-30), dict(re=3, max_it=1, tau=1e-30)),
    (dict(truncation=20, max_iter=30, tol=1e-30), dict(trunc=20, max_it=30, tau=1e-30)),   # > 8 directions: chunked kernels
])
def test_gcr_corner_parameters_vs_oracle(kw_o, kw_g):
    n = 9
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    val = val * (1.0 + 0.1j)
    b = problems.rhs_grid(N, 4)
    Ao = orc.csr(N, ncol, rowptr, col, val)
    xo, ho, ito, co = orc.gcr_solve(Ao, orc.gcr_param(**kw_o), b)
    _, sens, rng = orc.gcr_reorder_sensitivity(Ao, orc.gcr_param(**kw_o), b)
    A = Sparse(N, ncol, rowptr, col, val)
    x = Field((N,)).set_zero()
    gcr = GCR(A, GCR_Param(verb=False, **kw_g))
    fb = Field((N,), b)
    gcr.solve(fb, x)
    assert its_close(gcr.last_iterations, ito, rng) and gcr.last_converged == co
    hist_close(gcr.last_history, ho, str(kw_g), sens)
    x_close(x, A, fb, gcr, xo if gcr.last_iterations == ito else None, sens)


def test_zero_rhs_behaves_like_the_reference():
    """b = 0: |r|^2/|b|^2 is NaN, the reference's `while (NaN > tol^2 && ...)` is false: one iteration,
    reported as converged (global_count != max_iter), NaN in the history (no breakdown guard)."""
    n = 6
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    A = Sparse(N, ncol, rowptr, col, val)
    x = Field((N,)).set_zero()
    gcr = GCR(A, GCR_Param(0, 5, 50, 1e-10, False))
    gcr.solve(Field((N,)).set_zero(), x)
    xo, ho, ito, co = orc.gcr_solve(orc.csr(N, ncol, rowptr, col, val), orc.gcr_param(restart=5, max_iter=50, tol=1e-10), np.zeros(N, np.complex128))
    assert gcr.last_iterations == ito == 1 and gcr.last_converged == co
    assert np.isnan(gcr.last_history[1]) and np.isnan(ho[1])


def test_solver_object_is_reusable_and_reads_params_at_solve_time():
    n = 10
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    A = Sparse(N, ncol, rowptr, col, val)
    b = Field((N,), problems.rhs_grid(N, 1))
    prm = GCR_Param(0, 5, 30, 1e-30, False)
    gcr = GCR(A, prm)
    x1 = Field((N,)).set_zero()
    gcr.solve(b, x1)
    h1 = gcr.last_history.copy()
    prm.restart, prm.max_iter = 3, 12      # GCR_Param is held by pointer in the reference and read per solve
    x2 = Field((N,)).set_zero()
    gcr.solve(b, x2)
    assert gcr.last_iterations == 12
    fresh = GCR(A, GCR_Param(0, 3, 12, 1e-30, False))
    x3 = Field((N,)).set_zero()
    fresh.solve(b, x3)
    assert np.array_equal(gcr.last_history, fresh.last_history) and np.array_equal(x2.to_numpy(), x3.to_numpy())
    prm.restart, prm.max_iter = 5, 30
    x4 = Field((N,)).set_zero()
    gcr.solve(b, x4)
    assert np.array_equal(gcr.last_history, h1) and np.array_equal(x4.to_numpy(), x1.to_numpy())  # run-to-run reproducible


def _stencil_csr(dims, offsets_nd, vals):
    """Constant-coefficient stencil on a row-major grid with truncated boundaries: entry (p, p + o) exists iff p + o stays
    inside the grid in every dimension; columns ascending within a row."""
    dims = tuple(int(d) for d in dims)
    N = int(np.prod(dims))
    idx = np.indices(dims).reshape(len(dims), -1)
    strides = np.array([int(np.prod(dims[d + 1:])) for d in range(len(dims))], np.int64)
    lin = [(int(np.dot(o, strides)), o, v) for o, v in zip(offsets_nd, vals)]
    lin.sort(key=lambda t: t[0])
    masks, cols, vs = [], [], []
    rows = np.arange(N, dtype=np.int64)
    for lo, o, v in lin:
        ok = np.ones(N, bool)
        for d in range(len(dims)):
            ok &= (idx[d] + o[d] >= 0) & (idx[d] + o[d] < dims[d])
        masks.append(ok)
        cols.append(rows + lo)
        vs.append(v)
    mask = np.stack(masks, axis=1)
    rowptr = np.zeros(N + 1, np.int64)
    np.cumsum(mask.sum(axis=1), out=rowptr[1:])
    col = np.stack(cols, axis=1)[mask]
    val = np.broadcast_to(np.array(vs, np.complex128), mask.shape)[mask]
    return N, rowptr, col, val


@pytest.mark.parametrize("seed", range(12))
def test_stencil_view_random_stencils_bit_exact(seed):
    """The stencil view (format 3) on random constant-coefficient stencils: 1 to 3 dimensions, 2 to 9 slots, complex or
    real values, grids whose row count is no multiple of the wave, tile or workgroup sizes, near offsets that do / do not
    reach the LDS window's threshold.  y must equal the oracle's CSR row loop bit for bit — stand-alone apply (window and
    plain kernel), DiracOp epilogue, and inside a solve (fused apply, one-workgroup path excluded by the size) — and the
    dictionary kernels (stencil view off) must give the same bits."""
    rng = np.random.default_rng(900 + seed)
    nd = int(rng.integers(1, 4))
    if nd == 1:
        dims = (int(rng.integers(40000, 70000)),)
    elif nd == 2:
        dims = (int(rng.integers(190, 300)), int(rng.integers(190, 300)))
    else:
        dims = (int(rng.integers(33, 46)), int(rng.integers(33, 46)), int(rng.integers(33, 46)))
    ns = int(rng.integers(2, 10))
    offs = {tuple([0] * nd)}
    while len(offs) < ns:
        offs.add(tuple(int(v) for v in rng.integers(-2 if nd > 1 else -40, 3 if nd > 1 else 41, nd)))
    offs = sorted(offs)
    real = seed % 3 == 0
    vals = [complex(rng.uniform(-1, 1), 0. if real else rng.uniform(-1, 1)) for _ in offs]
    vals[offs.index(tuple([0] * nd))] = 4.0 * ns          # dominant diagonal: the solve below converges
    N, rowptr, col, val = _stencil_csr(dims, offs, vals)
    assert N >= 2 ** 15
    x = problems.rhs_grid(N, seed)
    O = orc.csr(N, N, rowptr, col, val)
    ref = O(x)
    k = 0.11 - 0.07j
    refd = orc.dirac(O, k)(x)
    xf = Field((N,), x)
    b = Field((N,), problems.rhs_grid(N, seed + 50))
    outs = []
    for stencil in (1, 0):
        prev = mg.set_option("stencil_storage", stencil)
        try:
            A = Sparse(N, N, rowptr, col, val)
            fmt, npat = A.storage_format()
            assert fmt == (3 if stencil else 1), (fmt, npat, dims, offs)
            if stencil:
                assert npat == ns
            assert np.array_equal(A(xf).to_numpy(), ref), (dims, offs)
            assert np.array_equal(DiracOp(A, k)(xf).to_numpy(), refd), (dims, offs)
            g = GCR(A, GCR_Param(0, 4, 12, 1e-30, False))
            xs = Field((N,)).set_zero()
            g.solve(b, xs)
            outs.append((g.last_history.copy(), xs.to_numpy()))
        finally:
            mg.set_option("stencil_storage", prev)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_rare_tail_stencil_kernels_same_bits(monkeypatch):
    """The kernels a distributed row block uses (stencil view in its rare-tail layout: 7 common slots + 2 rarely present
    ones looked at after the common sum — MODE 4 of the fused apply, sten_spmv<9, true>) run here on a single-GPU operator
    (MGCR_TEST_FORCE_RARE: the two extra slots have no presence bits): apply, DiracOp epilogue and a whole solve must
    have the bits of the 7-slot kernels.  (With real rare slots they are exercised by tests/test_gpu_dist.py, 48^3.)"""
    n = 40
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    val = val * (0.9 + 0.3j)
    x = Field((N,), problems.rhs_grid(N, 4))
    b = Field((N,), problems.rhs_grid(N, 5))
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("MGCR_TEST_FORCE_RARE", force)
        A = Sparse(N, ncol, rowptr, col, val)
        assert A.storage_format() == (3, 7)
        g = GCR(A, GCR_Param(0, 5, 23, 1e-30, False))
        xs = Field((N,)).set_zero()
        g.solve(b, xs)
        out.append((A(x).to_numpy(), DiracOp(A, 0.2 - 0.1j)(x).to_numpy(), g.last_history.copy(), xs.to_numpy()))
    for a, c in zip(out[0], out[1]):
        assert np.array_equal(a, c)


@pytest.mark.parametrize("n,planes", [(300, 4), (512, 2), (200, 5), (320, 3), (192, 4), (64, 40)])   # halo 64 k: the window is filled by LDS-DMA
def test_wide_plane_window_kernels_bit_exact(n, planes):
    """Grids whose lines are longer than 256 points: the +-n neighbours are served by the 1024-row LDS window (halo up to 512)
    of the stand-alone SpMV and of the fused apply + dots / fused step 0 (rows reach >= 2^15 rows away), n = 200 by the
    512-row window.  y equals the oracle's CSR row loop bit for bit, and a solve through the fused kernels has the bits of
    the solve through the separate SpMV + dot-product kernels and of the dictionary storage."""
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, 0, planes, ni=planes)
    val = val * (1.0 - 0.5j)
    x = problems.rhs_grid(N, 2)
    O = orc.csr(N, ncol, rowptr, col, val)
    xf = Field((N,), x)
    b = Field((N,), problems.rhs_grid(N, 3))
    A = Sparse(N, ncol, rowptr, col, val)
    assert A.storage_format() == (3, 7)
    assert np.array_equal(A(xf).to_numpy(), O(x))
    assert np.array_equal(DiracOp(A, 0.1 + 0.05j)(xf).to_numpy(), orc.dirac(O, 0.1 + 0.05j)(x))
    out = []
    for fused, stencil in ((1, 1), (0, 1), (1, 0)):
        p1, p2 = mg.set_option("fused_apply", fused), mg.set_option("stencil_storage", stencil)
        try:
            B = A if stencil else Sparse(N, ncol, rowptr, col, val)
            for prm in (GCR_Param(0, 4, 11, 1e-30, False), GCR_Param(0, 10, 2, 1e-30, False)):   # restart cycles; a smoother-shaped solve (fused step 0)
                g = GCR(B, prm)
                xs = Field((N,)).set_zero()
                g.solve(b, xs)
                out.append((fused, stencil, g.last_history.copy(), xs.to_numpy()))
        finally:
            mg.set_option("fused_apply", p1)
            mg.set_option("stencil_storage", p2)
    for k in (0, 1):
        for o in out[2 + k::2]:
            assert np.array_equal(out[k][2], o[2]) and np.array_equal(out[k][3], o[3]), (n, o[0], o[1])
