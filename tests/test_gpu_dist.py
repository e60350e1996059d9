"""Multi-rank HIP path on ONE GPU: 2 and 3 processes share cuda:0 and talk through the
host-staged (gloo) transport — same partition plan, halo-aware SpMV kernels, folded + all-reduced
scalars and device-side convergence logic as the RCCL build, only the wire differs.  Checked
against the single-process HIP solve of the same system."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

mg = pytest.importorskip("mgpreconditionedgcr_amd")
from mgpreconditionedgcr_amd import Field, GCR, GCR_Param, Sparse, problems  # noqa: E402
from tests.test_dist_cpu import run_workers  # noqa: E402
from tests.dist_worker import problem  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)


def test_distributed_gcr_with_transport_collectives(tmp_path, monkeypatch):
    """Same comparison with the peer-write kernels switched off: scalars and halos travel through the transport's own
    all-reduce / exchange (here the host callbacks; on a multi-GPU node RCCL) — the fallback every rank takes when the
    peer-write self-test fails anywhere."""
    monkeypatch.setenv("MGCR_PEER_ALLREDUCE", "0")
    test_distributed_gcr_matches_single_process(tmp_path, 2)


@pytest.mark.parametrize("n,nz,world", [(256, 16, 2), (200, 28, 2), (256, 24, 3)])
def test_distributed_slab_carried_window(tmp_path, monkeypatch, n, nz, world):
    """Two ranks, 8 planes of a 256 x 256 grid (14 of a 200 x 200 grid: the ragged plane walk) each: the row blocks' windowed kernels carry the far neighbours from trip to trip (the halo
    columns are rare slots of their own).  Same history, same x as with the far neighbours gathered (MGCR_TILE_CARRY=0), on every rank;
    and the single-GPU solve within re-association."""
    mg.init()          # (3 ranks: the middle one has a halo plane on either side — the leading slot AND the rare one behind the common slots)
    (tmp_path / "carry").mkdir()
    (tmp_path / "gather").mkdir()
    res = run_workers("slab:%d:%d" % (n, nz), world, tmp_path / "carry", timeout=240)
    monkeypatch.setenv("MGCR_TILE_CARRY", "0")
    ref = run_workers("slab:%d:%d" % (n, nz), world, tmp_path / "gather", timeout=240)
    monkeypatch.delenv("MGCR_TILE_CARRY")
    for r in range(world):
        assert res[r]["slab"]["format"] == 3 and res[r]["slab"]["layout"]["reach"] == n * n
        assert np.array_equal(res[r]["slab"]["hist"], ref[r]["slab"]["hist"]) and np.array_equal(res[r]["slab"]["hist"], res[0]["slab"]["hist"])
        assert np.array_equal(res[r]["slab"]["x"], ref[r]["slab"]["x"]) and np.array_equal(res[r]["slab"]["y"], ref[r]["slab"]["y"])
        for tag in ("restart12", "trunc4_shift_x0"):
            assert np.array_equal(res[r]["slab"]["more"][tag]["hist"], ref[r]["slab"]["more"][tag]["hist"]), (tag, r)
            assert np.array_equal(res[r]["slab"]["more"][tag]["x"], ref[r]["slab"]["more"][tag]["x"]), (tag, r)
            assert np.isfinite(res[r]["slab"]["more"][tag]["hist"]).all() and res[r]["slab"]["more"][tag]["hist"][-1] < res[r]["slab"]["more"][tag]["hist"][0]
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
    A = Sparse(N, ncol, rowptr, col, val)
    b = Field((N,), problems.rhs_grid(N, 1))
    y = A(b).to_numpy()
    assert np.array_equal(np.concatenate([res[r]["slab"]["y"] for r in range(world)]), y)
    xs = Field((N,)).set_zero()
    g = GCR(A, GCR_Param(0, 5, 12, 1e-30, False))
    g.solve(b, xs)
    h = res[0]["slab"]["hist"]
    assert h.size == g.last_history.size and np.max(np.abs(h - g.last_history) / g.last_history) < 1e-12
    # ... and bit for bit against the oracle's rank model WITH each rank's banded row map (equal blocks: one map for all): the distributed
    # solve keeps the update kernel (plain |r|^2), its first step and its dot products run in the apply kernels' order
    nloc = N // world
    band, per = orc.row_map(nloc, n * n)
    with orc.device_order(rank_offsets=np.arange(world + 1, dtype=np.int64) * nloc, band=band, per=per, plane=orc.row_map_plane(nloc, n * n),
                          init_banded=True, lean=True):
        Ao = orc.csr(N, ncol, rowptr, col, val)
        xo, ho, ito, _ = orc.gcr_solve(Ao, orc.gcr_param(restart=5, max_iter=12, tol=1e-30), problems.rhs_grid(N, 1))
    for r in range(world):
        assert np.array_equal(res[r]["slab"]["hist"], ho), (r, int(np.argmax(res[r]["slab"]["hist"] != ho)))
    assert np.array_equal(xd_all := np.concatenate([res[r]["slab"]["x"] for r in range(world)]), xo)
    xd = np.concatenate([res[r]["slab"]["x"] for r in range(world)])
    assert np.abs(xd - xs.to_numpy()).max() <= 1e-11 * np.abs(xs.to_numpy()).max()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_gcr_matches_single_process(tmp_path, world):
    mg.init()
    res = run_workers("gcr", world, tmp_path, timeout=240)
    for kind in ("poisson", "random", "poisson48"):
        N, rowptr, col, val, gran = problem(kind)
        A = Sparse(N, N, rowptr, col, val)
        # stencil view on EVERY rank: the upper halo column a rarely present slot behind the common ones, the lower one — first in its
        # rows' storage order — a slot summed before them (spmv.hip sten_try: leading slot)
        assert all(res[r][kind]["format"] == (3 if kind == "poisson48" else 0) for r in range(world)), [res[r][kind]["format"] for r in range(world)]
        # the per-iteration scalars went through the peer-write mailboxes (self-test passed on every rank), unless
        # the run asked for the transport's own all-reduce
        want = "host" if os.environ.get("MGCR_PEER_ALLREDUCE") == "0" else "peer-write"
        if want == "peer-write" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            # without dmabuf IPC the mailboxes cannot be shared on this pool: the self-test then sends every rank to the
            # transport's collectives, which is the behaviour to check in that environment
            want = res[0][kind]["allreduce"]
            assert want in ("peer-write", "host")
        assert all(res[r][kind]["allreduce"] == want for r in range(world)), [res[r][kind]["allreduce"] for r in range(world)]
        want_h = "host" if os.environ.get("MGCR_PEER_HALO") == "0" or want == "host" else "peer-write"
        assert all(res[r][kind]["halo"] == want_h for r in range(world)), [res[r][kind]["halo"] for r in range(world)]
        # the stand-alone apply of a peer-write rank runs SPLIT by default — boundary rows stored and published, the rows that need
        # no halo multiplied, then the wait for the neighbours and the boundary rows — with the bits of the unsplit apply
        for r in range(world):
            assert np.array_equal(res[r][kind]["y"], res[r][kind]["y_unsplit"]) and res[r][kind]["n_unsplit"] == 0
            if want_h == "peer-write" and kind == "poisson48":     # (row blocks of several planes: rows that touch no halo column exist on every rank)
                assert res[r][kind]["n_split"] == 1, (kind, r, res[r][kind]["n_split"])
            assert res[r][kind]["n_split"] in (0, 1)
        if want_h == "peer-write" and kind == "poisson":
            assert sum(res[r][kind]["n_split"] for r in range(world)) >= 1
        # peer-write scalars: the fold + cross-rank sum of a stencil row block's reductions runs in the last workgroup of the kernel that
        # produces the partials (2 per iteration: apply + dots, build) — same history and x as with the separate fold launches
        for r in range(world):
            assert np.array_equal(res[r][kind]["hist"], res[r][kind]["hist_notail"]) and np.array_equal(res[r][kind]["x"], res[r][kind]["x_notail"])
            if want == "peer-write":
                assert res[r][kind]["n_tail"] >= (40 if kind == "poisson48" else 20), (kind, r, res[r][kind]["n_tail"])
            else:
                assert res[r][kind]["n_tail"] == 0
        if kind == "poisson":
            print("all-reduce of 11 doubles, %d ranks on one GPU, %s: %.1f us" % (world, want, res[0][kind]["allreduce_us"]))
        x = problems.rhs_grid(N, 5)
        y = A(Field((N,), x)).to_numpy()
        got = np.concatenate([res[r][kind]["y"] for r in range(world)])
        assert np.abs(got - y).max() <= 1e-13 * np.abs(y).max()
        b = Field((N,), problems.rhs_grid(N, 1))
        Ao = orc.csr(N, N, rowptr, col, val)
        # BIT FOR BIT against the oracle summing in the distributed device's order (tests/test_gpu_bitwise.py, here with the rank
        # model: every rank sums its rows with its own grid, the rank totals are added in rank order — which the peer-write
        # all-reduce guarantees; with two ranks any order gives the same bits): apply, restarted and truncated GCR histories.
        # (Poisson kinds: every rank's block has the same ELL width and no tail; a row's entries keep their CSR order, halo columns
        # included, also where the block is stored as a stencil view.)
        if kind != "random" and (want == "peer-write" or world == 2):
            from tests.dist_worker import split_rows
            offs = np.array(split_rows(N // gran, world), np.int64) * gran
            with orc.device_order(rank_offsets=offs, lean=True):
                assert np.array_equal(got, Ao(x)), kind
                for tag, okw in (("", dict(restart=4, max_iter=25, tol=1e-30)), ("_trunc", dict(truncation=11, max_iter=25, tol=1e-30))):
                    xo, ho, ito, _ = orc.gcr_solve(Ao, orc.gcr_param(**okw), problems.rhs_grid(N, 1))
                    for r in range(world):
                        assert np.array_equal(res[r][kind]["hist" + tag], ho), (kind, tag, r)
                    assert np.array_equal(np.concatenate([res[r][kind]["x" + tag] for r in range(world)]), xo), (kind, tag)   # the solution too
        for tag, prm, okw in (("", GCR_Param(0, 4, 25, 1e-30, False), dict(restart=4, max_iter=25, tol=1e-30)),
                              ("_trunc", GCR_Param(11, 0, 25, 1e-30, False), dict(truncation=11, max_iter=25, tol=1e-30))):
            # how far the reference algorithm itself moves under re-association of its dot products
            _, sens, _ = orc.gcr_reorder_sensitivity(Ao, orc.gcr_param(**okw), problems.rhs_grid(N, 1))
            _, xsens = orc.gcr_x_sensitivity(Ao, orc.gcr_param(**okw), problems.rhs_grid(N, 1))
            xs = Field((N,)).set_zero()
            gcr = GCR(A, prm)
            gcr.solve(b, xs)
            for r in range(world):
                h = res[r][kind]["hist" + tag]
                assert h.size == gcr.last_history.size
                # same algorithm, dot products summed in a different order (per-rank folds + all-reduce)
                tol = np.maximum(1e-8 * gcr.last_history, 8 * sens) + 1e-17
                assert (np.abs(h - gcr.last_history)[1:] <= tol[1:]).all(), (kind, tag, r)
                assert np.array_equal(h, res[0][kind]["hist" + tag])  # every rank sees identical scalars
            xd = np.concatenate([res[r][kind]["x" + tag] for r in range(world)])
            # x after the same number of steps: within what the reference algorithm's own x moves when its
            # dot products are summed in another order (unconverged solves amplify that: poisson48)
            assert np.abs(xd - xs.to_numpy()).max() <= max(1e-7 * np.abs(xs.to_numpy()).max(), 20 * xsens)


def test_rccl_single_rank_collectives(tmp_path):
    """RCCL binding on the one GPU we have: a 1-rank communicator with the fold + ncclAllReduce path
    forced on (MGCR_TEST_FORCE_COLLECTIVES) must reproduce the plain single-GPU solve bit for bit."""
    import subprocess
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
os.environ["MGCR_TEST_FORCE_COLLECTIVES"] = "1"
import torch, torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1)
import mgpreconditionedgcr_amd as mg
from mgpreconditionedgcr_amd import Comm, DistSparse, Field, GCR, GCR_Param, Sparse, problems
mg.init(0)
comm = Comm.rccl(dist)
n = 20
N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
A = DistSparse(comm, N, 0, rowptr, col, val)
B = Sparse(N, ncol, rowptr, col, val)
b = Field((N,)).fill_rhs(3)
out = []
for op in (A, B):
    x = Field((N,)).set_zero()
    g = GCR(op, GCR_Param(0, 5, 40, 1e-30, False))
    g.solve(b, x)
    out.append((g.last_history, x.to_numpy()))
assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
print("RCCL_OK")
''' % ROOT
    from tests.test_dist_cpu import free_port
    p = subprocess.run([sys.executable, "-c", code, str(free_port())], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_peer_write_wait_times_out_cleanly(tmp_path, monkeypatch):
    """A rank that never joins an all-reduce: the waiting kernel gives up after its time limit (here 300 ms) and the
    call returns MGCR_ERR_COMM — no wave spins on the GPU without bound."""
    monkeypatch.setenv("MGCR_PEER_TIMEOUT_MS", "300")
    res = run_workers("pw-timeout", 2, tmp_path, timeout=240)
    if res[0]["kind"] != "peer-write":
        pytest.skip("peer-write path not available here (%s)" % res[0]["kind"])
    assert res[0]["rc"] != 0 and "did not arrive" in res[0]["error"], res[0]
    assert res[0]["seconds"] < 30.0
    assert res[1]["rc"] == 0


def test_peer_write_halo_timeout_is_reported_by_an_apply(tmp_path, monkeypatch):
    """A neighbour that never starts its halo exchange, seen from an operator apply with no solver around it: the apply
    is enqueued, the download that hands y back returns MGCR_ERR_COMM, and the rows next to the missing halo are NaN —
    a stale or half-written receive slot can not yield plausible numbers."""
    monkeypatch.setenv("MGCR_PEER_TIMEOUT_MS", "300")
    res = run_workers("pw-timeout-apply", 2, tmp_path, timeout=240)
    if res[0]["kind"] != "peer-write":
        pytest.skip("peer-write halo path not available here (%s)" % res[0]["kind"])
    assert res[0]["rc_apply"] == 0                      # enqueueing succeeds ...
    assert res[0]["rc"] != 0 and "did not arrive" in res[0]["error"], res[0]     # ... the hand-back does not
    assert res[0]["seconds"] < 30.0
    y = res[0]["y"]
    n = 6
    assert np.isnan(y[-n * n:]).all()                   # the last plane reads the halo that never came
    assert np.isfinite(y[:-n * n]).all()
    assert res[1]["rc"] == 0


def test_peer_write_sequence_number_wrap(tmp_path, monkeypatch):
    """The peer-write exchanges alternate between two slots by sequence parity; across the 32-bit wrap the sequence goes
    0xFFFFFFFF -> 2 (never 0, parity keeps alternating).  Starting three steps below the wrap, a distributed solve must
    give what it gives anywhere else."""
    monkeypatch.setenv("MGCR_TEST_PW_SEQ0", "0xFFFFFFFD")
    test_distributed_gcr_matches_single_process(tmp_path, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_near_null_vectors(tmp_path, world):
    """MG_Param without given null vectors on a distributed operator: the inverse iteration (src/MG.h:90-122) runs with
    global norms and dot products and finds the vectors the single-process run finds (to solver tolerance)."""
    from mgpreconditionedgcr_amd import MG, MG_Param, Mesh
    mg.init()
    res = run_workers("nullvec", world, tmp_path, timeout=400)
    n, planes = 8, 8
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, 0, world * planes, ni=world * planes)
    A = Sparse(N, ncol, rowptr, col, val)
    prm = MG_Param(Mesh((world * planes, n, n)), 2, 2, GCR_Param(0, 10, 400, 1e-12, False), GCR(GCR_Param(0, 10, 50, 1e-2, False)),
                   GCR(GCR_Param(0, 10, 2, 1e-30, False)), 1, None, None)
    ref = MG(None, prm).near_null_vectors(A)
    got = np.concatenate([res[r]["vecs"] for r in range(world)], axis=1)
    assert got.shape == ref.shape == (2, N)
    gram = got.conj() @ got.T
    assert np.abs(gram - np.eye(2)).max() < 1e-12          # orthonormal GLOBALLY
    assert np.abs(got - ref).max() < 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize("world,mode", [(2, "mg"), (3, "mg"), (2, "mg-large")])
def test_distributed_mg_gcr_matches_single_process(tmp_path, world, mode):
    """BASELINE config 4 in miniature: slab-partitioned Poisson, 3-level aggregation MG built
    collectively (Galerkin across the slab boundaries, distributed coarse operators), flexible
    outer GCR — against the single-process MG-GCR of the same global problem."""
    from mgpreconditionedgcr_amd import MG, MG_Param, Mesh
    mg.init()
    res = run_workers(mode, world, tmp_path, timeout=240)
    n, planes = (8, 8) if mode == "mg" else (48, 16)
    ni = world * planes
    N, ncol, rowptr, col, val = problems.poisson3d_csr(n, 0, ni, ni=ni)
    A = Sparse(N, ncol, rowptr, col, val)
    dims = (ni, n, n)
    prm = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                   2, None, None, null_vectors=np.ones((1, N), np.complex128))
    M = MG(A, prm)
    b = problems.rhs_grid(N, 3)
    y = M(Field(dims, b)).to_numpy()
    yd = np.concatenate([res[r]["mg"]["y"] for r in range(world)])
    for l in range(3):
        assert sum(res[r]["mg"]["levels"][l]["dim"] for r in range(world)) == M.level_info(l)["dim"]
    assert np.abs(yd - y).max() <= 1e-9 * np.abs(y).max()        # one V-cycle
    outer = GCR(A, GCR_Param(0, 5, 60, 1e-9, False, None, M, flexible=True))
    x = Field(dims).set_zero()
    outer.solve(Field(dims, b), x)
    for r in range(world):
        assert res[r]["mg"]["conv"] and abs(res[r]["mg"]["its"] - outer.last_iterations) <= 1
        m_ = min(res[r]["mg"]["hist"].size, outer.last_history.size)
        assert np.allclose(res[r]["mg"]["hist"][1:m_], outer.last_history[1:m_], rtol=1e-5, atol=1e-15)
    xd = np.concatenate([res[r]["mg"]["x"] for r in range(world)])
    rr = Field(dims, b) - A(Field(dims, xd))
    assert rr.norm() / np.linalg.norm(b) <= 2e-9


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_mg_on_unstructured_blocks(tmp_path, world):
    """BASELINE config 5 in miniature (row-block partition + per-neighbour gather lists, potentially
    every peer): the unstructured block operator dealt to 2 / 3 ranks by block rows, MG with aggregates of two block
    rows built collectively, flexible outer GCR — against the single-process solve of the same operator."""
    from mgpreconditionedgcr_amd import MG, MG_Param, Mesh
    from tests.dist_worker import unstructured_blocks
    mg.init()
    res = run_workers("mg-unstructured", world, tmp_path, timeout=240)
    nb, bs, rowptr, col, val = unstructured_blocks()
    N = nb * bs
    from mgpreconditionedgcr_amd import DiracOp
    A0 = Sparse(N, N, rowptr, col, val)
    A = DiracOp(A0, 0.05 - 0.02j)
    dims = (nb, bs)
    vecs = np.random.default_rng(9).standard_normal((2, N)) + 1j * np.random.default_rng(10).standard_normal((2, N))
    prm = MG_Param(Mesh(dims), 2, 2, None, GCR(GCR_Param(0, 10, 30, 1e-3, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                   1, None, None, spacetime=[True, False], null_vectors=vecs)
    M = MG(A, prm)
    b = problems.rhs_grid(N, 3)
    y = M(Field(dims, b)).to_numpy()
    yd = np.concatenate([res[r]["mg"]["y"] for r in range(world)])
    assert sum(res[r]["mg"]["levels"][1]["dim"] for r in range(world)) == M.level_info(1)["dim"]
    assert np.abs(yd - y).max() <= 1e-9 * np.abs(y).max()
    outer = GCR(A, GCR_Param(0, 5, 60, 1e-10, False, None, M, flexible=True))
    x = Field(dims).set_zero()
    outer.solve(Field(dims, b), x)
    for r in range(world):
        assert res[r]["mg"]["conv"] and abs(res[r]["mg"]["its"] - outer.last_iterations) <= 1
    xd = np.concatenate([res[r]["mg"]["x"] for r in range(world)])
    rr = Field(dims, b) - A(Field(dims, xd))
    assert rr.norm() / np.linalg.norm(b) <= 2e-10


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_hierarchical_sparse(tmp_path, world):
    """BASELINE configs[4] with the operator as it is (reference apply: src/HierarchicalSparse.h:101-161): a distributed
    HierarchicalSparse — block rows dealt to the ranks, block columns anywhere, halo at block granularity read in place by
    the block kernel.  Apply bit-equal to the single-process operator (same block order in every row, duplicates kept);
    GCR on it; MG with aggregates of two block rows whose Galerkin coarse operator is a distributed block-CSR again."""
    from mgpreconditionedgcr_amd import HierarchicalSparse, MG, MG_Param, Mesh
    from tests.dist_worker import unstructured_block_triplets
    mg.init()
    res = run_workers("bcsr", world, tmp_path, timeout=240)
    nb, bs, rows, cols, blocks = unstructured_block_triplets()
    N = nb * bs
    H = HierarchicalSparse(nb, nb, rows.astype(np.int32), cols.astype(np.int32), blocks)
    dims = (nb, bs)
    xv = problems.rhs_grid(N, 3)
    y = H(Field(dims, xv)).to_numpy()
    yd = np.concatenate([res[r]["bcsr"]["y"] for r in range(world)])
    assert np.array_equal(yd, y)                                   # bit for bit
    want_h = "host" if os.environ.get("MGCR_PEER_HALO") == "0" or os.environ.get("MGCR_PEER_ALLREDUCE") == "0" else "peer-write"
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0":
        assert all(res[r]["bcsr"]["halo"] == want_h for r in range(world))
    g = GCR(H, GCR_Param(0, 4, 30, 1e-30, False))
    xs = Field(dims).set_zero()
    g.solve(Field(dims, xv), xs)
    hd = res[0]["bcsr"]["hist"]
    assert hd.size == g.last_history.size
    assert np.abs(hd - g.last_history).max() <= 1e-9 * np.abs(g.last_history).max()
    xd = np.concatenate([res[r]["bcsr"]["x"] for r in range(world)])
    assert np.abs(xd - xs.to_numpy()).max() <= 1e-9 * np.abs(xs.to_numpy()).max()
    vecs = np.random.default_rng(9).standard_normal((2, N)) + 1j * np.random.default_rng(10).standard_normal((2, N))
    prm = MG_Param(Mesh(dims), 2, 2, None, GCR(GCR_Param(0, 10, 30, 1e-3, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                   1, None, None, spacetime=[True, False], null_vectors=vecs)
    M = MG(H, prm)
    assert sum(res[r]["bcsr"]["levels"][1]["dim"] for r in range(world)) == M.level_info(1)["dim"]
    assert all(res[r]["bcsr"]["coarse_is_block"] for r in range(world))
    ymg = M(Field(dims, xv)).to_numpy()
    ymgd = np.concatenate([res[r]["bcsr"]["ymg"] for r in range(world)])
    assert np.abs(ymgd - ymg).max() <= 1e-9 * np.abs(ymg).max()
    outer = GCR(H, GCR_Param(0, 5, 60, 1e-10, False, None, M, flexible=True))
    xo = Field(dims).set_zero()
    outer.solve(Field(dims, xv), xo)
    for r in range(world):
        assert res[r]["bcsr"]["conv"] and abs(res[r]["bcsr"]["its"] - outer.last_iterations) <= 1
    xod = np.concatenate([res[r]["bcsr"]["xo"] for r in range(world)])
    rr = Field(dims, xv) - H(Field(dims, xod))
    assert rr.norm() / np.linalg.norm(xv) <= 2e-10
