"""Bit-for-bit parity of the HIP path with a CPU model of itself — the proof that SUMMATION ORDER is the only thing
in which the device path differs from the reference.

The argument has two halves, and both are asserted here with np.array_equal:

  (1) oracle, summation order 0 (index order)  ==  the real reference        (golden vectors, tests/golden/*.npz —
      outputs of the reference's own classes compiled here, oracle/ref_harness.cpp);
  (2) oracle, summation order 3 (device order) ==  the GPU                   (every entry of the residual history,
      the iteration count, the convergence flag, AND the solution x — in restart mode with the oracle's model of the lean
      cycles' coefficient tables, which associate the same linear combination differently).

Order 3 (oracle/mgcr_oracle.c, "order 3") changes NOTHING in the oracle but the association of the sums: the terms
conj(a_i) b_i are the same doubles, added per thread in ascending row order, then by the wave64 tree of csrc/reduce.h,
the 16 waves of a workgroup in order, and the per-workgroup partials by the same tree — and, for matrices whose rows are
dealt to several lanes or keep a CSR tail, the row sums are associated the way csrc/spmv.hip's kernels do.  Element-wise
arithmetic, the conjugation order, the complex division, the restart / truncation bookkeeping, the stop test are the
order-0 code.  So wherever tests/test_gpu_parity.py sees a deviation from the reference (up to 0.85 relative on
p16_restart3, tests/golden/observed_r02.json), this file shows it is re-association and nothing else: 0 ulp against
order 3.

Covered: every golden history of the reference (4x4 sample: restart 5 / restart 2 / truncation 8 / full / max_iter 0 /
complex k / left and right literal preconditioner hooks; Poisson 8^3 .. 128^3), on the multi-kernel path, the
one-workgroup path (gcr_small.hip), the one-launch resident solver (gcr_resident.hip) and the one-launch steps
(gcr_stepbuild.hip); 192^3 (banded row map + LDS-window kernels); a seeded sweep over modes / shifts / x0.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("mgpreconditionedgcr_amd")
from mgpreconditionedgcr_amd import DiracOp, Field, GCR, GCR_Param, Sparse, problems, read_data  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = (4, 4, 4, 4, 4, 3)
RECORD = {}


@pytest.fixture(scope="module", autouse=True)
def _init():
    mg.init()
    yield
    if RECORD:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "observed_bitwise.json"), "w") as f:
            json.dump(RECORD, f, indent=1, sort_keys=True)


@pytest.fixture(params=["multi-kernel", "one-workgroup"])
def solver_path(request):
    mg.lib().mgcr_set_small_solve_rows(0 if request.param == "multi-kernel" else 16384)
    yield request.param
    mg.lib().mgcr_set_small_solve_rows(1024)


def device_model(A, N, took_small):
    """The oracle's model parameters for the solve that just ran on operator A (N rows)."""
    if took_small:
        return orc.device_order(blocks=1)     # one workgroup; rows summed sequentially (gcr_small.hip:small_row)
    lay = A.ell_layout()
    band, per = orc.row_map(N, lay["reach"])
    lanes = lay["lanes"] > 1 or lay["tail_rows"] > 0
    return orc.device_order(blocks=0, band=band, per=per, init_banded=band > 0,
                            ell_width=lay["ell_width"] if lanes else -1, ell_lanes=lay["lanes"], tail_cap=lay["tail_chunk_cap"],
                            lean=True,      # (restart mode: x from the lean cycles' coefficient tables, like the device)
                            xr_banded=band > 0 and A.xr_fuse_kind() in (1, 2),   # residual update inside the apply kernel: |r|^2 over ITS row map
                            plane=orc.row_map_plane(N, lay["reach"]))


def solve_both(A, Ao, N, gp, po, b, x0=None, dims=None):
    """GPU solve, then the oracle in the device's order.  Returns (gcr, x_gpu, (x, hist, it, conv) of the oracle)."""
    dims = dims or (N,)
    fb = Field(dims, b)
    x = Field(dims, x0) if x0 is not None else Field(dims).set_zero()
    gcr = GCR(A, gp)
    s0 = mg.stat("small_solves")
    gcr.solve(fb, x)
    took_small = mg.stat("small_solves") > s0
    with device_model(A, N, took_small):
        ref = orc.gcr_solve(Ao, po, b, x0)
    return gcr, x, ref, took_small


def assert_bitwise(tag, path, gcr, ref, golden=None, x=None):
    xo, ho, ito, co = ref
    h = gcr.last_history
    RECORD.setdefault(tag, {})[path] = {
        "iterations_gpu": int(gcr.last_iterations), "iterations_oracle_device_order": int(ito),
        "entries_compared": int(min(h.size, ho.size)),
        "max_abs_dev_vs_device_order": float(np.abs(h[:min(h.size, ho.size)] - ho[:min(h.size, ho.size)]).max()),
        "iterations_reference": None if golden is None else int(golden.size - 1)}
    assert gcr.last_iterations == ito, "%s: %d iterations, oracle in device order %d" % (tag, gcr.last_iterations, ito)
    assert np.array_equal(h, ho), "%s: first differing step %d" % (tag, int(np.argmax(h != ho)))
    assert gcr.last_converged == co
    if x is not None:   # the SOLUTION too: the oracle forms x the way the path that ran does (lean cycles' tables / iteration order)
        xg = x.to_numpy().ravel()
        RECORD[tag][path]["x_max_abs_dev_vs_device_order"] = float(np.abs(xg - xo).max())
        assert np.array_equal(xg, xo), "%s: x differs in %d of %d entries (max %.3e)" % (tag, int((xg != xo).sum()), xo.size, np.abs(xg - xo).max())


@pytest.fixture(scope="module")
def sample(sample_matrix_path):
    D = read_data(os.path.basename(sample_matrix_path), directory=os.path.dirname(sample_matrix_path))
    nrow, ncol, rowptr, col, val = orc.read_text_csr(sample_matrix_path)
    return D, orc.csr(nrow, ncol, rowptr, col, val)


def _okw(kw):
    m = dict(re="restart", trunc="truncation", max_it="max_iter", tau="tol")
    return {m[k]: v for k, v in kw.items()}


SAMPLE_CASES = [
    ("g3_restart5", 0.15, dict(re=5, max_it=4000, tau=1e-13)),
    ("g4_restart2", 0.15, dict(re=2, max_it=4000, tau=1e-13)),
    ("g5_trunc8", 0.15, dict(trunc=8, max_it=300, tau=1e-3)),
    ("g6_full", 0.15, dict(max_it=60, tau=1e-13)),
    ("g10_maxiter0", 0.15, dict(re=10, max_it=0, tau=1e-8)),
    ("g3b_complexk", 0.12 + 0.05j, dict(re=5, max_it=40, tau=1e-13)),
]


@pytest.mark.parametrize("tag,k,kw", SAMPLE_CASES)
def test_sample_histories_bit_for_bit(sample, sample_gold, tag, k, kw, solver_path):
    """4x4 sample (39 entries per row: 8 lanes per row on the multi-kernel path, sequential rows in the one-workgroup solver)."""
    D, Do = sample
    g = sample_gold
    b = g["gcr_rhs"]
    po = orc.gcr_param(**_okw(kw))
    Ao = orc.dirac(Do, k)
    # half (1): index order == the reference
    gold = g[tag + "_hist"]
    _, h0, it0, _ = orc.gcr_solve(Ao, po, b)
    m = min(h0.size, gold.size)
    assert np.array_equal(h0[1:m], gold[1:m]) and (tag == "g6_full" or it0 == gold.size - 1)
    # half (2): device order == the GPU
    gcr, x, ref, small = solve_both(DiracOp(D, k), Ao, 3072, GCR_Param(verb=False, **kw), po, b, dims=DIMS)
    assert small or solver_path == "multi-kernel" or tag == "g6_full" or kw.get("re", 0) > 8   # (the one-workgroup solver keeps <= 8 directions)
    assert_bitwise(tag, solver_path if small else "multi-kernel", gcr, ref, gold, x)


def test_sample_literal_preconditioner_hooks_bit_for_bit(sample, sample_gold):
    """r = M(r) / Ar = Ml(Ar) (src/GCR.h:197-204,236-247), M = 1 + 0.15 D: goldens g11_*."""
    D, Do = sample
    g = sample_gold
    b = g["gcr_rhs"]
    for tag, left, n_it in (("g11_right_neumann", False, 20), ("g11_left_neumann", True, 60)):
        Mo = orc.dirac(Do, -0.15)
        po = orc.gcr_param(restart=5, max_iter=n_it, tol=1e-13, left=Mo if left else None, right=None if left else Mo)
        Ao = orc.dirac(Do, 0.15)
        gold = g[tag + "_hist"]
        _, h0, _, _ = orc.gcr_solve(Ao, po, b)
        assert np.array_equal(h0[1:], gold[1:])
        M = DiracOp(D, -0.15)
        gp = GCR_Param(0, 5, n_it, 1e-13, False, M if left else None, None if left else M)
        gcr, x, ref, small = solve_both(DiracOp(D, 0.15), Ao, 3072, gp, po, b, dims=DIMS)
        assert not small
        assert_bitwise(tag, "multi-kernel", gcr, ref, gold, x)


POISSON_CASES = [(32, dict(re=5, max_it=10, tau=1e-13), "p32"),
                 (8, dict(trunc=4, max_it=300, tau=1e-10), "p8_trunc4"),
                 (8, dict(max_it=25, tau=1e-10), "p8_full"),
                 (16, dict(re=3, max_it=300, tau=1e-12), "p16_restart3")]


@pytest.mark.parametrize("n,kw,tag", POISSON_CASES)
def test_poisson_histories_bit_for_bit(poisson_gold, n, kw, tag, solver_path):
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    b = problems.rhs_grid(N, 0)
    po = orc.gcr_param(**_okw(kw))
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gold = poisson_gold[tag + "_hist"]
    _, h0, it0, _ = orc.gcr_solve(Ao, po, b)
    assert np.array_equal(h0[1:], gold[1:]) and it0 == gold.size - 1
    for resident in ((1, 0) if n == 32 else (1,)):   # 32^3, restart 5: the one-launch resident solver, and the kernels it replaces
        prev = mg.set_option("resident_solver", resident)
        try:
            r0 = mg.stat("resident_solves")
            gcr, x, ref, small = solve_both(Sparse(N, ncol, rowptr, col, val), Ao, N, GCR_Param(verb=False, **kw), po, b, dims=(n, n, n))
            took_resident = mg.stat("resident_solves") > r0
        finally:
            mg.set_option("resident_solver", prev)
        path = "one-workgroup" if small else "resident" if took_resident else "multi-kernel"
        assert_bitwise(tag, path, gcr, ref, gold, x)


def test_poisson128_headline_bit_for_bit(poisson_gold):
    """BASELINE configs[1] — the bench's own solve (Poisson 128^3, GCR restart 5): the reference's first 10 steps (golden
    p128), the oracle in index order == that golden, and the GPU == the oracle in device order over 20 steps, through the
    one-launch steps (gcr_stepbuild.hip) and through the three kernels they replace."""
    n = 128
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    b = problems.rhs_grid(N, 0)
    Ao = orc.csr(N, ncol, rowptr, col, val)
    _, h0, _, _ = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=10, tol=1e-13), b)
    assert np.array_equal(h0[1:], poisson_gold["p128_hist"][1:])
    A = Sparse(N, ncol, rowptr, col, val)
    po = orc.gcr_param(restart=5, max_iter=20, tol=1e-13)
    for step_build in (1, 0):
        prev = mg.set_option("step_build", step_build)
        try:
            l0 = mg.stat("step_build_launches")
            gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, 5, 20, 1e-13, False), po, b, dims=(n, n, n))
            launches = mg.stat("step_build_launches") - l0
        finally:
            mg.set_option("step_build", prev)
        assert (launches > 0) == bool(step_build)
        assert_bitwise("p128_20steps", "one-launch steps" if step_build else "three kernels", gcr, ref, poisson_gold["p128_hist"], x)
        assert np.array_equal(gcr.last_history[1:11], ref[1][1:11])


def test_poisson192_banded_row_map_bit_for_bit():
    """192^3 (7 M rows): rows reach 36 864 rows away, so the kernels that embed the apply deal their rows in eight bands
    (gcr_dev.h:make_row_map) and stage x in LDS windows (gcr_fused.hip *_tile_kernel) — the layout of BASELINE
    configs[2] (256^3), at a size the oracle finishes in seconds.  6 steps, restart 5 (one cycle closes)."""
    n = 192
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
    b = problems.rhs_grid(N, 0)
    A = Sparse(N, ncol, rowptr, col, val)
    lay = A.ell_layout()
    assert orc.row_map(N, lay["reach"])[0] > 0
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, 5, 6, 1e-13, False), orc.gcr_param(restart=5, max_iter=6, tol=1e-13), b, dims=(n, n, n))
    assert_bitwise("p192_6steps", "multi-kernel (banded)", gcr, ref, None, x)


@pytest.mark.parametrize("restart,steps,nz", [(5, 7, 16), (10, 12, 16), (3, 3, 16), (5, 7, 21), (12, 14, 16), (16, 18, 16)])
def test_poisson_256x256_slab_carried_window_bit_for_bit(restart, steps, nz):
    """16 planes of a 256 x 256 grid (1 M rows): the far neighbours of a row are exactly one step of the banded row map away, so the
    windowed kernels carry them in registers from trip to trip and the residual update runs inside the apply kernel
    (gcr_fused.hip CARRY, gcr_fused_xr_tile.h) — the kernels of BASELINE configs[2] (256^3) at a size the oracle finishes in seconds.
    |r|^2 of every step but the last is then summed over the banded map: the oracle's model follows (xr_banded).  The same solve with
    the separate update kernel (test_carried_window_and_fused_update_variants_agree) differs in those sums only.  21 planes: bands of
    2.625 planes — a band may start in the middle of a plane."""
    n = 256
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
    b = problems.rhs_grid(N, 0)
    A = Sparse(N, ncol, rowptr, col, val)
    lay = A.ell_layout()
    band, per = orc.row_map(N, lay["reach"])
    assert band > 0 and per * 1024 == n * n and A.xr_fuse_kind() == 2
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, restart, steps, 1e-13, False), orc.gcr_param(restart=restart, max_iter=steps, tol=1e-13), b,
                                    dims=(nz, n, n))
    assert not small
    assert_bitwise("p256x256x%d_restart%d_%dsteps" % (nz, restart, steps), "multi-kernel (banded, carried window)", gcr, ref, None, x)


_VARIANT_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, ".")
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import *
n, nz = 256, 16
N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
A = Sparse(N, ncol, rowptr, col, val)
b = Field((nz, n, n), problems.rhs_grid(N, 0))
x = Field((nz, n, n)).set_zero()
g = GCR(A, GCR_Param(0, 5, 12, 1e-30, False))
g.solve(b, x)
np.save(sys.argv[1], np.concatenate([np.asarray(g.last_history), x.to_numpy().ravel().view(np.float64)]))
print("kind", A.xr_fuse_kind())
"""


def test_carried_window_and_fused_update_variants_agree(tmp_path):
    """The same 12 steps of GCR(5) on the 256 x 256 x 16 slab in three child processes: far neighbours gathered (MGCR_TILE_CARRY=0),
    carried in registers with the update kernel separate (MGCR_XR_FUSE_TILE=0), and the default (update inside the windowed apply).
    The first two agree in every bit; the third differs from them in |r|^2's summation order only: x identical, history to 1e-15."""
    import subprocess
    import sys
    outs = {}
    for tag, env_add in (("gathered", dict(MGCR_TILE_CARRY="0")), ("carried", dict(MGCR_XR_FUSE_TILE="0")), ("fused", {})):
        f = str(tmp_path / (tag + ".npy"))
        p = subprocess.run([sys.executable, "-c", _VARIANT_CHILD, f], env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=240, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[tag] = (np.load(f), p.stdout)
    assert "kind 0" in outs["gathered"][1] and "kind 0" in outs["carried"][1] and "kind 2" in outs["fused"][1]
    h = 13
    assert np.array_equal(outs["gathered"][0], outs["carried"][0])
    assert np.array_equal(outs["carried"][0][h:], outs["fused"][0][h:])
    a, b = outs["carried"][0][:h], outs["fused"][0][:h]
    assert not np.array_equal(a, b) and np.max(np.abs(a - b) / a) < 1e-15


@pytest.mark.parametrize("nz,ny,nx", [(32, 128, 256), (20, 256, 128), (6, 256, 512)])
def test_plane_walk_row_map_on_boxes_bit_for_bit(nz, ny, nx):
    """Planes that are not square: 128 x 256 (32 workgroups per band, 16 bands: two whole bands per XCD would also do — strips are what
    runs), 256 x 128 (the +- n neighbours 128 rows away), 256 x 512 (128 workgroups per band, 4 bands; window halo 512)."""
    N, ncol, rowptr, col, val = problems.poisson3d_box_csr(nz, ny, nx)
    b = problems.rhs_grid(N, 0)
    A = Sparse(N, ncol, rowptr, col, val)
    lay = A.ell_layout()
    band, per = orc.row_map(N, lay["reach"])
    assert lay["reach"] == ny * nx and per == ny * nx // 1024 and band > 0 and A.xr_fuse_kind() == 2
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, 5, 7, 1e-13, False), orc.gcr_param(restart=5, max_iter=7, tol=1e-13), b, dims=(nz, ny, nx))
    assert not small
    assert_bitwise("p%dx%dx%d_restart5_7steps" % (nz, ny, nx), "multi-kernel (plane walk, carried window)", gcr, ref, None, x)


@pytest.mark.parametrize("nz,ny,nx", [
    (15, 192, 192),    # fewer planes than twice the bands: bands of 1.3 planes
    (14, 200, 200),    # ragged, 12 bands for 14 planes: 7 bands of 2 planes are used
    (18, 100, 320),    # ragged, plane = 500 waves
    (14, 64, 640),     # planes of 40 x 1024 rows whose +- nx neighbours are too far for the LDS window: the plane-walk map under the un-windowed kernels
    (9, 250, 250),     # planes of 62 500 rows: not a multiple of 64 — the 8-band map stays
    (3, 512, 512),     # 256 workgroups per band, 2 bands, 3 planes
])
def test_row_map_edge_shapes_bit_for_bit(nz, ny, nx):
    """Shapes at the edges of gcr_dev.h make_row_map's cases; 6 steps of GCR(5), history and x against the oracle's model of the map
    (oracle.row_map / row_map_plane mirror it)."""
    N, ncol, rowptr, col, val = problems.poisson3d_box_csr(nz, ny, nx)
    assert N >= 512 * 1024
    b = problems.rhs_grid(N, 0)
    A = Sparse(N, ncol, rowptr, col, val)
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, 5, 6, 1e-13, False), orc.gcr_param(restart=5, max_iter=6, tol=1e-13), b, dims=(nz, ny, nx))
    assert not small
    RECORD.setdefault("row_map_edge_%dx%dx%d" % (nz, ny, nx), {})["map"] = dict(zip(("band", "per"), orc.row_map(N, A.ell_layout()["reach"])),
                                                                                  plane=orc.row_map_plane(N, A.ell_layout()["reach"]), xr_fuse_kind=A.xr_fuse_kind())
    assert_bitwise("row_map_edge_%dx%dx%d" % (nz, ny, nx), "multi-kernel", gcr, ref, None, x)


@pytest.mark.parametrize("n,nz", [(200, 24), (264, 15), (328, 10), (400, 7)])
def test_ragged_plane_walk_row_map_bit_for_bit(n, nz):
    """Planes of a multiple of 64 sites that is not a multiple of 1024 (n a multiple of 8): ceil(n^2 / 1024) workgroups tile one plane —
    the last tile is short, its threads beyond the plane's end fill their window entry and nothing else — and step by the plane; whole
    planes per band.  Carried window and fused residual update as on the other sizes; the oracle's model follows (row_map_plane)."""
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
    b = problems.rhs_grid(N, 0)
    A = Sparse(N, ncol, rowptr, col, val)
    lay = A.ell_layout()
    band, per = orc.row_map(N, lay["reach"])
    assert orc.row_map_plane(N, lay["reach"]) == n * n and per == (n * n + 1023) // 1024 and band % (n * n) == 0 and A.xr_fuse_kind() == 2
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, 5, 7, 1e-13, False), orc.gcr_param(restart=5, max_iter=7, tol=1e-13), b, dims=(nz, n, n))
    assert not small
    assert_bitwise("p%dx%dx%d_restart5_7steps" % (n, n, nz), "multi-kernel (ragged plane walk, carried window)", gcr, ref, None, x)


@pytest.mark.parametrize("n,nz", [(192, 24), (320, 10), (384, 6), (512, 4)])
def test_plane_walk_row_map_bit_for_bit(n, nz):
    """Grids whose planes hold a multiple of 1024 sites other than 256 x 256: the band's workgroups tile one plane and step from plane to
    plane (gcr_dev.h make_row_map: 36, 100, 144, 256 workgroups per band; 12, 4, 3, 2 bands; the logical workgroups behind the last band
    have no rows), so the windowed kernels carry the far neighbours and the residual update runs inside the apply here as well.  7 steps
    of GCR(5) against the oracle's model of that map."""
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
    b = problems.rhs_grid(N, 0)
    A = Sparse(N, ncol, rowptr, col, val)
    lay = A.ell_layout()
    band, per = orc.row_map(N, lay["reach"])
    assert per == n * n // 1024 and band > 0 and A.xr_fuse_kind() == 2
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, 5, 7, 1e-13, False), orc.gcr_param(restart=5, max_iter=7, tol=1e-13), b, dims=(nz, n, n))
    assert not small
    assert_bitwise("p%dx%dx%d_restart5_7steps" % (n, n, nz), "multi-kernel (plane walk, carried window)", gcr, ref, None, x)


@pytest.mark.parametrize("nz,ny,nx", [(16, 256, 256), (24, 200, 200)])
def test_shifted_operator_on_a_slab_bit_for_bit(nz, ny, nx):
    """DiracOp = 1 - k D with a complex k on the slabs of the carried-window kernels (D = the 7-point operator: real stencil
    coefficients, the shift is the kernels' epilogue y = r' - k (D r')): fused update + apply, stand-alone apply, 7 steps of GCR(5)."""
    k = 0.05 + 0.02j
    N, ncol, rowptr, col, val = problems.poisson3d_box_csr(nz, ny, nx)
    b = problems.rhs_grid(N, 0)
    D = Sparse(N, ncol, rowptr, col, val)
    A = DiracOp(D, k)
    assert A.xr_fuse_kind() == 2
    Ao = orc.dirac(orc.csr(N, ncol, rowptr, col, val), k)
    with device_model(A, N, False):
        assert np.array_equal(A(Field((nz, ny, nx), b)).to_numpy().ravel(), Ao(b))
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(0, 5, 7, 1e-13, False), orc.gcr_param(restart=5, max_iter=7, tol=1e-13), b, dims=(nz, ny, nx))
    assert not small
    assert_bitwise("shifted_p%dx%dx%d_restart5_7steps" % (nz, ny, nx), "multi-kernel (carried window)", gcr, ref, None, x)


@pytest.mark.parametrize("kw,okw", [(dict(trunc=6, max_it=10), dict(truncation=6, max_iter=10)), (dict(max_it=9), dict(max_iter=9))])
def test_truncated_and_full_gcr_on_a_slab_bit_for_bit(kw, okw):
    """The classic (not lean) kernels — truncated and full GCR — on the 256 x 256 x 16 slab: carried-window apply + dots, separate update."""
    n, nz = 256, 16
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
    b = problems.rhs_grid(N, 0)
    A = Sparse(N, ncol, rowptr, col, val)
    Ao = orc.csr(N, ncol, rowptr, col, val)
    gcr, x, ref, small = solve_both(A, Ao, N, GCR_Param(tau=1e-13, verb=False, **kw), orc.gcr_param(tol=1e-13, **okw), b, dims=(nz, n, n))
    assert not small
    assert_bitwise("p256x256x16_%s" % "_".join("%s%d" % kv for kv in sorted(okw.items())), "multi-kernel (banded, carried window)", gcr, ref, None, x)


@pytest.mark.parametrize("hook", ["right", "left", "flexible"])
def test_preconditioner_hooks_on_a_slab_bit_for_bit(hook):
    """The literal hooks r = M(r) / Ar = Ml(Ar) (src/GCR.h:197-204,236-247) and the flexible right preconditioner on the 256 x 256 x 16 slab,
    M = 1 + 0.04 D (an operator apply through the carried-window kernel with its shift epilogue): the solver then keeps the update kernel and
    the plain-order start; 9 steps of GCR(4)."""
    n, nz = 256, 16
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
    b = problems.rhs_grid(N, 2)
    D = Sparse(N, ncol, rowptr, col, val)
    Do = orc.csr(N, ncol, rowptr, col, val)
    M, Mo = DiracOp(D, -0.04), orc.dirac(Do, -0.04)
    flex = hook == "flexible"
    gp = GCR_Param(0, 4, 9, 1e-13, False, M if hook == "left" else None, None if hook == "left" else M, flexible=flex)
    po = orc.gcr_param(restart=4, max_iter=9, tol=1e-13, left=Mo if hook == "left" else None, right=None if hook == "left" else Mo, flexible=flex)
    gcr, x, ref, small = solve_both(D, Do, N, gp, po, b, dims=(nz, n, n))
    assert not small
    assert_bitwise("p256x256x16_hook_%s" % hook, "multi-kernel (banded, carried window)", gcr, ref, None, x)


def test_carried_window_apply_same_bits():
    """The stand-alone apply (A x and the shifted x - k A x) of the 256 x 256 x 16 slab with and without the carried-window kernel
    (gcr_fused.hip sten_apply_carry_kernel; MGCR_APPLY_CARRY=0 takes spmv.hip's sten_spmv_tile): identical results."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "apply_carry_check.py")], capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "identical: True" in p.stdout, p.stdout


def _fuzz_system(rng):
    kind = rng.choice(["poisson-small", "poisson-slab", "poisson-pattern", "random", "random-wide"])
    if kind in ("random", "random-wide"):
        N = int(rng.integers(200, 3000))
        wide = kind == "random-wide"
        rowptr, col, val = problems.random_csr(N, N, rng, min_len=10 if wide else 1, max_len=30 if wide else 8,
                                               **(dict(long_rows=4, long_len=300) if wide else {}))
        rows = np.repeat(np.arange(N), np.diff(rowptr))
        rowsum = np.bincount(rows, weights=np.abs(val), minlength=N)
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


@pytest.mark.parametrize("seed", range(64))
def test_random_parameters_bit_for_bit(seed):
    """The sweep of tests/test_gpu_fuzz.py (modes, shifts, x0, solver paths, storages; plus matrices with multi-lane rows
    and CSR tails), here with equality instead of the sensitivity envelope."""
    rng = np.random.default_rng(5000 + seed)
    kind, N, rowptr, col, val = _fuzz_system(rng)
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
    if shift is not None:
        Ao, A = orc.dirac(Ao, shift), DiracOp(A, shift)
    po = orc.gcr_param(use_x0=use_x0, **kw)
    gp = GCR_Param(kw.get("truncation", 0), kw.get("restart", 0), kw["max_iter"], kw["tol"], False, use_x0=use_x0,
                   check_every=int(rng.choice([0, 1, 3, 50])))
    small_limit = int(rng.choice([0, 1024, 16384]))
    mg.lib().mgcr_set_small_solve_rows(small_limit)
    try:
        gcr, x, ref, small = solve_both(A, Ao, N, gp, po, b, x0)
    finally:
        mg.lib().mgcr_set_small_solve_rows(1024)
    what = "%s N=%d %s shift=%s x0=%s small=%s" % (kind, N, kw, shift, use_x0, small)
    h, ho = gcr.last_history, ref[1]
    m = min(h.size, ho.size)
    assert np.array_equal(h[:m], ho[:m]), "%s: first differing step %d (%.17g against %.17g)" % (what, int(np.argmax(h[:m] != ho[:m])), h[int(np.argmax(h[:m] != ho[:m]))], ho[int(np.argmax(h[:m] != ho[:m]))])
    assert gcr.last_iterations == ref[2], what
    assert gcr.last_converged == ref[3], what
    xg = x.to_numpy()
    assert np.array_equal(xg, ref[0]), "%s: x differs in %d entries (max %.3e)" % (what, int((xg != ref[0]).sum()), np.abs(xg - ref[0]).max())


@pytest.mark.parametrize("seed", range(int(os.environ.get("MGCR_SLAB_FUZZ_SEEDS", "12"))))   # (more seeds: export MGCR_SLAB_FUZZ_SEEDS=60)
def test_random_parameters_on_slabs_bit_for_bit(seed):
    """The same sweep of solver parameters (modes, restart / truncation lengths up to 17, shifts, x0, tolerances that end a solve in the
    middle of a cycle) on slabs that take the plane-walk row maps, the carried window and the fused residual update."""
    rng = np.random.default_rng(9000 + seed)
    nz, ny, nx = [(16, 256, 256), (24, 200, 200), (24, 192, 192), (10, 320, 320), (20, 256, 128)][seed % 5]
    N, ncol, rowptr, col, val = problems.poisson3d_box_csr(nz, ny, nx)
    mode = rng.choice(["restart", "truncation", "full"], p=[0.6, 0.25, 0.15])
    kw = dict(max_iter=int(rng.choice([1, 2, 3, 7, 12])), tol=float(rng.choice([1e-30, 3e-2, 1e-1])))
    if mode == "restart":
        kw["restart"] = int(rng.integers(1, 18))
    elif mode == "truncation":
        kw["truncation"] = int(rng.integers(1, 13))
    shift = complex(rng.uniform(0.02, 0.08), rng.uniform(-0.03, 0.03)) if rng.random() < 0.35 else None
    use_x0 = bool(rng.random() < 0.3)
    b = problems.rhs_grid(N, int(rng.integers(0, 50)))
    x0 = problems.rhs_grid(N, 77) * 0.1 if use_x0 else None
    Ao = orc.csr(N, ncol, rowptr, col, val)
    A = Sparse(N, ncol, rowptr, col, val)
    if shift is not None:
        Ao, A = orc.dirac(Ao, shift), DiracOp(A, shift)
    po = orc.gcr_param(use_x0=use_x0, **kw)
    gp = GCR_Param(kw.get("truncation", 0), kw.get("restart", 0), kw["max_iter"], kw["tol"], False, use_x0=use_x0, check_every=int(rng.choice([0, 1, 3, 50])))
    gcr, x, ref, small = solve_both(A, Ao, N, gp, po, b, x0, dims=(nz, ny, nx))
    what = "%dx%dx%d %s shift=%s x0=%s" % (nz, ny, nx, kw, shift, use_x0)
    h, ho = gcr.last_history, ref[1]
    m = min(h.size, ho.size)
    assert not small
    assert np.array_equal(h[:m], ho[:m]), "%s: first differing step %d" % (what, int(np.argmax(h[:m] != ho[:m])))
    assert gcr.last_iterations == ref[2] and gcr.last_converged == ref[3], what
    xg = x.to_numpy().ravel()
    assert np.array_equal(xg, ref[0]), "%s: x differs in %d entries (max %.3e)" % (what, int((xg != ref[0]).sum()), np.abs(xg - ref[0]).max())


def test_dot_and_norm_bit_for_bit(sample_gold):
    """Field::dot / squarednorm (src/Fields.h:216-235) through mgcr_dot / mgcr_norm2: device order == the GPU."""
    for n, seed in ((3072, 1), (1000, 2), (70000, 3), (1 << 21, 4)):
        a, b = problems.rhs_grid(n, seed), problems.rhs_grid(n, seed + 10)
        fa, fb = Field((n,), a), Field((n,), b)
        with orc.device_order():
            d, s = orc.dot(a, b), orc.sqnorm(a)
        assert fa.dot(fb) == d and fa.squarednorm() == s


def test_dagger_behind_a_borrowed_handle():
    """Sparse::dagger works in place on the object a DiracOp / GCR points at (src/Operator.h:117,296-328): operators that
    borrowed the handle before the dagger apply the daggered matrix afterwards (mgcr_csr_replace keeps the handle alive)."""
    rng = np.random.default_rng(7)
    N = 900
    rowptr, col, val = problems.random_csr(N, N, rng, min_len=1, max_len=9)
    S = Sparse(N, N, rowptr, col, val)
    k = 0.2 - 0.1j
    Dk = DiracOp(S, k)
    gcr = GCR(Dk, GCR_Param(0, 5, 3, 1e-30, False))
    x = problems.rhs_grid(N, 4)
    fx = Field((N,), x)
    before = Dk(fx).to_numpy()
    with device_model(S, N, False):     # (rows longer than the ELL width keep a tail: the row sums' association, see module docstring)
        assert np.array_equal(before, orc.dirac(orc.csr(N, N, rowptr, col, val), k)(x))
    S.dagger()
    So = orc.csr(N, N, S.ROW, S.COL, S.VAL)
    with device_model(S, N, False):
        assert np.array_equal(S(fx).to_numpy(), So(x))
        after = Dk(fx).to_numpy()                      # the DiracOp made BEFORE the dagger
        assert np.array_equal(after, orc.dirac(So, k)(x)) and not np.array_equal(after, before)
    sol = Field((N,)).set_zero()
    s0 = mg.stat("small_solves")
    gcr.solve(fx, sol)                                 # ... and the GCR made before it
    with device_model(S, N, mg.stat("small_solves") > s0):
        ref = orc.gcr_solve(orc.dirac(So, k), orc.gcr_param(restart=5, max_iter=3, tol=1e-30), x)
    assert np.array_equal(gcr.last_history, ref[1])
    S.dagger()                                         # twice = the original matrix
    assert np.array_equal(Dk(fx).to_numpy(), before)


@pytest.mark.parametrize("nrow,ncol,kw", [
    (257, 300, dict(min_len=0, max_len=9)),                                  # empty rows, ragged
    (3000, 2500, dict(min_len=0, max_len=6, long_rows=5, long_len=900)),     # tail rows summed by the chunk kernel (one thread, CSR order)
    (6000, 6000, dict(min_len=1, max_len=7, long_rows=7, long_len=3000)),    # ... and rows longer than a chunk (one wave, lanes + tree)
    (40000, 40000, dict(min_len=3, max_len=40)),                             # many tail rows: chunks of up to 256 rows / 1024 entries
    (700, 700, dict(min_len=30, max_len=45)),                                # multi-lane rows
    (64, 4096, dict(min_len=1000, max_len=1500)),                            # everything long
])
def test_irregular_spmv_bit_for_bit(nrow, ncol, kw):
    """Sparse::operator() (src/Operator.h:330-346) on ELL slab + CSR tail layouts: the device's row sums are the oracle's with
    the rows associated as the layout says (mgcr_op_ell_layout) — and exactly the reference's wherever a row is summed by one
    thread in CSR order."""
    rng = np.random.default_rng(nrow * 7 + ncol)
    rowptr, col, val = problems.random_csr(nrow, ncol, rng, **kw)
    x = problems.rhs_grid(ncol, 3)
    A = Sparse(nrow, ncol, rowptr, col, val)
    Ao = orc.csr(nrow, ncol, rowptr, col, val)
    lay = A.ell_layout()
    y = A(Field((ncol,), x)).to_numpy()
    with device_model(A, nrow, False):
        assert np.array_equal(y, Ao(x)), lay
    if nrow == ncol:
        k = 0.3 - 0.2j
        yd = DiracOp(A, k)(Field((ncol,), x)).to_numpy()
        with device_model(A, nrow, False):
            assert np.array_equal(yd, orc.dirac(Ao, k)(x)), lay
    if lay["lanes"] == 1:
        # rows whose tail is summed by one thread: y_row = fl(ELL sum in CSR order) + fl(tail sum in CSR order)
        tails = np.diff(rowptr) - lay["ell_width"]
        seq = tails <= 0
        assert np.array_equal(y[seq], Ao(x)[seq])      # no tail: the reference's bits


@pytest.mark.parametrize("window,want_h", [(300, 1024), (3000, 4096), (30000, 0)])
def test_banded_irregular_spmv_window_bit_for_bit(window, want_h):
    """Banded irregular matrices (>= 90 % of the slab's columns within 1024 / 4096 rows of their row) are multiplied by the kernel
    that stages x in an LDS window (spmv.hip ell_spmv_window): same products in the same order as the slab kernel — the oracle's
    bits with the layout's association, DiracOp epilogue included, and the bits of the kernel without the window."""
    rng = np.random.default_rng(window)
    N = 70000
    rowptr, col, val = problems.skewed_csr(N, rng, window=window, long_rows=3, long_len=2500)
    x = problems.rhs_grid(N, 3)
    A = Sparse(N, N, rowptr, col, val)
    lay = A.ell_layout()
    assert lay["x_window"] == want_h and lay["tail_rows"] > 0
    Ao = orc.csr(N, N, rowptr, col, val)
    y = A(Field((N,), x)).to_numpy()
    k = 0.3 - 0.2j
    yd = DiracOp(A, k)(Field((N,), x)).to_numpy()
    with device_model(A, N, False):
        assert np.array_equal(y, Ao(x))
        assert np.array_equal(yd, orc.dirac(Ao, k)(x))
    if want_h:
        os.environ["MGCR_ELL_WINDOW"] = "0"      # (read once per process: only a child process sees it)
        import subprocess
        import sys
        code = ("import sys, numpy as np; sys.path.insert(0, %r); import mgpreconditionedgcr_amd as mg; from mgpreconditionedgcr_amd import problems, Sparse, Field;"
                "rng = np.random.default_rng(%d); rp, c, v = problems.skewed_csr(%d, rng, window=%d, long_rows=3, long_len=2500);"
                "A = Sparse(%d, %d, rp, c, v); assert A.ell_layout()['x_window'] == 0;"
                "np.save(sys.argv[1], A(Field((%d,), problems.rhs_grid(%d, 3))).to_numpy())" % (ROOT, window, N, window, N, N, N, N))
        out = os.path.join(ROOT, "gpurun_out", "y_nowindow_%d.npy" % window)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        try:
            p = subprocess.run([sys.executable, "-c", code, out], capture_output=True, text=True, timeout=300)
        finally:
            del os.environ["MGCR_ELL_WINDOW"]
        assert p.returncode == 0, p.stderr[-2000:]
        assert np.array_equal(y, np.load(out))
