"""Worker process of the multi-rank tests (launched by test_dist_cpu.py / test_gpu_dist.py).

    python tests/dist_worker.py <mode> <rank> <world> <port> <outdir>

modes
  plan   (CPU, gloo)  partition plan + numpy SpMV through the plan's halo lists vs the global SpMV
  gcr    (GPU, gloo-staged transport; all ranks may share GPU 0) distributed SpMV and GCR on the
         HIP path vs what the caller computes in one process
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def split_rows(n, world):
    base = n // world
    offs = [r * base for r in range(world)] + [n]
    return offs


def problem(kind):
    from mgpreconditionedgcr_amd import problems
    if kind in ("poisson", "poisson48"):
        # 48^3: every rank's row block has >= 2^15 rows, i.e. is stored as a row-pattern dictionary
        # (halo columns included: nloc + slot - row is constant along a boundary plane)
        n = 6 if kind == "poisson" else 48
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        return N, rowptr, col, val, (n * n)  # rows per plane: slabs must hold whole planes
    rng = np.random.default_rng(7)
    N = 1000
    rowptr, col, val = problems.random_csr(N, N, rng, min_len=1, max_len=9)
    # diagonally dominant so that GCR converges
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    rowsum = np.bincount(rows, weights=np.abs(val), minlength=N)
    newptr = rowptr + np.arange(N + 1)
    ncol_arr, nval = np.empty(newptr[-1], np.int64), np.empty(newptr[-1], np.complex128)
    for r in range(N):
        s, e = rowptr[r], rowptr[r + 1]
        ncol_arr[newptr[r]:newptr[r] + (e - s)] = col[s:e]
        nval[newptr[r]:newptr[r] + (e - s)] = val[s:e]
        ncol_arr[newptr[r + 1] - 1] = r
        nval[newptr[r + 1] - 1] = 2.0 * rowsum[r] + 1.0
    return N, newptr, ncol_arr, nval, 1


def unstructured_blocks(nb=24, bs=3, seed=5):
    """Config 5 in miniature: random block operator (2-5 blocks per block row, dominant diagonal blocks) written out
    as scalar CSR; every block row couples to block rows anywhere, i.e. to every rank of a row-block partition."""
    rng = np.random.default_rng(seed)
    rp, ci, va = [0], [], []
    for br in range(nb):
        others = rng.choice([c for c in range(nb) if c != br], size=int(rng.integers(1, 5)), replace=False)
        bcols = np.sort(np.concatenate([[br], others]))
        blk = {int(c): (rng.standard_normal((bs, bs)) + 1j * rng.standard_normal((bs, bs))) * 0.1 for c in bcols}
        blk[br] = blk[br] + np.eye(bs) * 3.0
        for r in range(bs):
            for c in bcols:
                ci += [int(c) * bs + k for k in range(bs)]
                va += list(blk[int(c)][r])
            rp.append(len(ci))
    return nb, bs, np.array(rp, np.int64), np.array(ci, np.int64), np.array(va, np.complex128)


def unstructured_block_triplets(nb=24, bs=4, seed=15):
    """The same kind of operator as (block_row, block_col, block) triplets — what HierarchicalSparse is built from — with a
    duplicated (row, col) pair here and there (kept and summed at apply time, src/HierarchicalSparse.h:20-21,89-94)."""
    rng = np.random.default_rng(seed)
    rows, cols, blocks = [], [], []
    for br in range(nb):
        others = rng.choice([c for c in range(nb) if c != br], size=int(rng.integers(1, 6)), replace=False)
        for c in [br] + [int(o) for o in others]:
            blk = (rng.standard_normal((bs, bs)) + 1j * rng.standard_normal((bs, bs))) * 0.1
            if c == br:
                blk = blk + np.eye(bs) * 3.0
            rows.append(br); cols.append(c); blocks.append(blk)
        if br % 5 == 2:   # a duplicate of this row's first off-diagonal block position
            rows.append(br); cols.append(int(others[0]))
            blocks.append((rng.standard_normal((bs, bs)) + 1j * rng.standard_normal((bs, bs))) * 0.05)
    order = rng.permutation(len(rows))   # triplets arrive unsorted
    return nb, bs, np.array(rows, np.int64)[order], np.array(cols, np.int64)[order], np.array(blocks)[order]


def local_block(rowptr, col, val, r0, r1):
    lp = rowptr[r0:r1 + 1] - rowptr[r0]
    return lp, col[rowptr[r0]:rowptr[r1]], val[rowptr[r0]:rowptr[r1]]


def main():
    mode, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from mgpreconditionedgcr_amd import Comm, Plan, problems
    comm = Comm.host(dist)
    results = {}
    if mode == "mg-unstructured":
        # distributed MG-GCR on the unstructured block operator: block rows dealt to the ranks, aggregates of 2
        # consecutive block rows, 2 given near-null vectors; the halo lists reach every other rank
        import mgpreconditionedgcr_amd as mg
        from mgpreconditionedgcr_amd import DistSparse, Field, GCR, GCR_Param, MG, MG_Param, Mesh
        mg.init(0)
        nb, bs, rowptr, col, val = unstructured_blocks()
        N = nb * bs
        per = nb // world
        r0, r1 = rank * per * bs, (rank + 1) * per * bs
        lp, lc, lv = local_block(rowptr, col, val, r0, r1)
        from mgpreconditionedgcr_amd import DiracOp
        A0 = DistSparse(comm, N, r0, lp, lc, lv)
        A = DiracOp(A0, 0.05 - 0.02j)           # the shifted form 1 - k D on a distributed D
        dims = (per, bs)
        vecs = np.random.default_rng(9).standard_normal((2, N)) + 1j * np.random.default_rng(10).standard_normal((2, N))
        prm = MG_Param(Mesh(dims), 2, 2, None, GCR(GCR_Param(0, 10, 30, 1e-3, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                       1, None, None, spacetime=[True, False], null_vectors=vecs[:, r0:r1])
        M = MG(A, prm)
        b = problems.rhs_grid(N, 3)[r0:r1]
        y = M(Field(dims, b)).to_numpy()
        outer = GCR(A, GCR_Param(0, 5, 60, 1e-10, False, None, M, flexible=True, check_every=3))
        x = Field(dims).set_zero()
        outer.solve(Field(dims, b), x)
        results["mg"] = dict(y=y, x=x.to_numpy(), hist=outer.last_history, its=outer.last_iterations, conv=outer.last_converged,
                             levels=[M.level_info(l) for l in range(2)])
        np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode.startswith("slab"):
        # 16 planes of a 256 x 256 grid over 2 ranks (8 planes = 512 x 1024 rows each: full grids): the row blocks take the plane-walk
        # row map and, unless MGCR_TILE_CARRY=0, the windowed kernels that carry the far neighbours in registers (gcr_fused.hip CARRY,
        # rare-slot layout: the halo columns); GCR(5), 12 steps, and the V-cycle-free flexible variant is left to the MG tests
        import mgpreconditionedgcr_amd as mg
        from mgpreconditionedgcr_amd import DistSparse, Field, GCR, GCR_Param
        mg.init(0)
        n, nz = (int(v) for v in mode.split(":")[1:3]) if ":" in mode else (256, 16)   # "slab:200:28": planes of 40 000 rows (ragged plane walk)
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n, ni=nz)
        per = nz // world
        r0, r1 = rank * per * n * n, (rank + 1) * per * n * n
        lp, lc, lv = local_block(rowptr, col, val, r0, r1)
        A = DistSparse(comm, N, r0, lp, lc, lv)
        bfull = problems.rhs_grid(N, 1)
        b = Field((r1 - r0,), bfull[r0:r1])
        y = A(b).to_numpy()
        xs = Field((r1 - r0,)).set_zero()
        gcr = GCR(A, GCR_Param(0, 5, 12, 1e-30, False, check_every=4))
        gcr.solve(b, xs)
        # ... and other shapes of solve through the same kernels: a cycle of 12 directions, truncated GCR on the shifted operator from x0 != 0
        from mgpreconditionedgcr_amd import DiracOp
        more = {}
        for tag, op, prm, x0 in (("restart12", A, GCR_Param(0, 12, 14, 1e-30, False, check_every=5), None),
                                 ("trunc4_shift_x0", DiracOp(A, 0.05 + 0.02j), GCR_Param(4, 0, 9, 1e-30, False, use_x0=True), problems.rhs_grid(N, 77)[r0:r1] * 0.1)):
            xv = Field((r1 - r0,), x0) if x0 is not None else Field((r1 - r0,)).set_zero()
            gg = GCR(op, prm)
            gg.solve(b, xv)
            more[tag] = dict(hist=gg.last_history.copy(), x=xv.to_numpy())
        results["slab"] = dict(y=y, hist=gcr.last_history, x=xs.to_numpy(), its=gcr.last_iterations, format=A.storage_format()[0],
                               layout=A.ell_layout(), allreduce=comm.allreduce_kind, halo=A.halo_kind, more=more)
        np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "bcsr":
        # BASELINE configs[4]'s operator as it is: a distributed HierarchicalSparse (block rows dealt to the ranks, block
        # columns anywhere), its apply, GCR on it, and MG with aggregates of two block rows whose Galerkin coarse operator
        # is a distributed HierarchicalSparse again
        import mgpreconditionedgcr_amd as mg
        from mgpreconditionedgcr_amd import DistHierarchicalSparse, Field, GCR, GCR_Param, MG, MG_Param, Mesh
        mg.init(0)
        nb, bs, rows, cols, blocks = unstructured_block_triplets()
        per = nb // world
        b0, b1 = rank * per, (rank + 1) * per if rank + 1 < world else nb
        sel = (rows >= b0) & (rows < b1)
        H = DistHierarchicalSparse(comm, nb, b0, b1 - b0, rows[sel] - b0, cols[sel], blocks[sel])
        N = nb * bs
        r0, r1 = b0 * bs, b1 * bs
        dims = (b1 - b0, bs)
        xv = problems.rhs_grid(N, 3)[r0:r1]
        y = H(Field(dims, xv)).to_numpy()
        g = GCR(H, GCR_Param(0, 4, 30, 1e-30, False))
        xs = Field(dims).set_zero()
        g.solve(Field(dims, xv), xs)
        vecs = np.random.default_rng(9).standard_normal((2, N)) + 1j * np.random.default_rng(10).standard_normal((2, N))
        prm = MG_Param(Mesh(dims), 2, 2, None, GCR(GCR_Param(0, 10, 30, 1e-3, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                       1, None, None, spacetime=[True, False], null_vectors=vecs[:, r0:r1])
        M = MG(H, prm)
        ymg = M(Field(dims, xv)).to_numpy()
        Ac = M.level_operator(1)
        nc = M.level_info(1)["dim"]
        wc = problems.rhs_grid(2 * (nb // 2), 6)[rank * nc: rank * nc + nc] if world > 1 else problems.rhs_grid(nc, 6)
        outer = GCR(H, GCR_Param(0, 5, 60, 1e-10, False, None, M, flexible=True, check_every=3))
        xo = Field(dims).set_zero()
        outer.solve(Field(dims, xv), xo)
        results["bcsr"] = dict(y=y, hist=g.last_history, x=xs.to_numpy(), halo=H.halo_kind, ymg=ymg, xo=xo.to_numpy(),
                               hist_mg=outer.last_history, its=outer.last_iterations, conv=outer.last_converged,
                               levels=[M.level_info(l) for l in range(2)], coarse_is_block=Ac.get_nrow() == nc)
        np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "pw-timeout":
        # a rank that never arrives: the peer-write wait must give up after its time limit and surface MGCR_ERR_COMM
        # (never spin on the GPU forever); MGCR_PEER_TIMEOUT_MS is shortened by the test
        import ctypes
        import mgpreconditionedgcr_amd as mg
        from mgpreconditionedgcr_amd import DistSparse
        mg.init(0)
        N, rowptr, col, val, gran = problem("poisson")
        offs = split_rows(N // gran, world)
        r0, r1 = offs[rank] * gran, offs[rank + 1] * gran
        lp, lc, lv = local_block(rowptr, col, val, r0, r1)
        A = DistSparse(comm, N, r0, lp, lc, lv)        # collective: self-tests pass here
        results["kind"] = comm.allreduce_kind
        us = ctypes.c_double()
        rc = 0
        if rank == 0 and results["kind"] == "peer-write":   # rank 1 stays away from this all-reduce
            import time
            t0 = time.time()
            rc = mg.lib().mgcr_comm_bench_allreduce(comm.h, 4, 1, ctypes.byref(us))
            results["seconds"] = time.time() - t0
            results["error"] = mg.lib().mgcr_last_error().decode()
        results["rc"] = rc
        np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        os._exit(0)   # the communicator is unusable after the timeout: skip its destructors
    if mode == "pw-timeout-apply":
        # the same for an operator apply alone (no solve around it): rank 1 never starts the halo exchange rank 0 waits
        # for.  The download that hands y back must fail with MGCR_ERR_COMM, and the rows that depend on the halo that
        # never arrived must be NaN, not stale values
        import ctypes
        import time
        import mgpreconditionedgcr_amd as mg
        from mgpreconditionedgcr_amd import DistSparse, Field
        mg.init(0)
        N, rowptr, col, val, gran = problem("poisson")
        offs = split_rows(N // gran, world)
        r0, r1 = offs[rank] * gran, offs[rank + 1] * gran
        lp, lc, lv = local_block(rowptr, col, val, r0, r1)
        A = DistSparse(comm, N, r0, lp, lc, lv)
        results["kind"] = A.halo_kind
        results["rc"] = 0
        if rank == 0 and results["kind"] == "peer-write":
            xf = Field((r1 - r0,), problems.rhs_grid(N, 5)[r0:r1])
            yf = Field((r1 - r0,))
            t0 = time.time()
            rc_apply = mg.lib().mgcr_op_apply(A.h, xf.h, yf.h)
            out = np.empty(r1 - r0, np.complex128)
            rc = mg.lib().mgcr_vec_download(yf.h, out.ctypes.data)
            results.update(rc_apply=rc_apply, rc=rc, seconds=time.time() - t0, error=mg.lib().mgcr_last_error().decode(), y=out)
        np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        os._exit(0)
    if mode == "nullvec":
        # MG::Arnoldi (src/MG.h:90-122) on a distributed operator: the inverse iteration's norms and the Gram-Schmidt
        # dot products must be GLOBAL (Comm.dot), and the start vector the same global vector on every world size
        import mgpreconditionedgcr_amd as mg
        from mgpreconditionedgcr_amd import DistSparse, GCR, GCR_Param, MG, MG_Param, Mesh
        mg.init(0)
        n, planes = 8, 8
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n, rank * planes, (rank + 1) * planes, ni=world * planes)
        A = DistSparse(comm, ncol, rank * N, rowptr, col, val)
        prm = MG_Param(Mesh((planes, n, n)), 2, 2, GCR_Param(0, 10, 400, 1e-12, False), GCR(GCR_Param(0, 10, 50, 1e-2, False)),
                       GCR(GCR_Param(0, 10, 2, 1e-30, False)), 1, None, None)
        results["vecs"] = MG(None, prm).near_null_vectors(A)
        np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode in ("mg", "mg-large"):
        # distributed 3-level aggregation MG as flexible right preconditioner (BASELINE config 4 shape, small)
        import mgpreconditionedgcr_amd as mg
        from mgpreconditionedgcr_amd import DistSparse, Field, GCR, GCR_Param, MG, MG_Param, Mesh
        mg.init(0)
        # every rank owns `planes` planes of a (planes*world) x n x n grid; "mg-large": local blocks of >= 2^15 rows,
        # i.e. row-pattern storage, the fused apply and the aliased smoother start on a distributed operator
        n, planes = (8, 8) if mode == "mg" else (48, 16)
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n, rank * planes, (rank + 1) * planes, ni=world * planes)
        A = DistSparse(comm, ncol, rank * N, rowptr, col, val)
        dims = (planes, n, n)                 # LOCAL mesh of this rank's row block
        prm = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                       2, None, None, null_vectors=np.ones((1, N), np.complex128))
        M = MG(A, prm)
        b = problems.rhs_grid(ncol, 3)[rank * N:(rank + 1) * N]
        y = M(Field(dims, b)).to_numpy()
        outer = GCR(A, GCR_Param(0, 5, 60, 1e-9, False, None, M, flexible=True, check_every=3))
        x = Field(dims).set_zero()
        outer.solve(Field(dims, b), x)
        results["mg"] = dict(y=y, x=x.to_numpy(), hist=outer.last_history, its=outer.last_iterations,
                             conv=outer.last_converged, levels=[M.level_info(l) for l in range(3)], allreduce=comm.allreduce_kind)
        np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    for kind in ("poisson", "random") + (("poisson48",) if mode == "gcr" else ()):
        N, rowptr, col, val, gran = problem(kind)
        offs = split_rows(N // gran, world)
        r0, r1 = offs[rank] * gran, offs[rank + 1] * gran
        lp, lc, lv = local_block(rowptr, col, val, r0, r1)
        x = problems.rhs_grid(N, 5)
        if mode == "plan":
            plan = Plan(comm, N, r0, lp, lc)
            # halo exchange by the plan's lists, with plain gloo point-to-point
            xl = x[r0:r1]
            halo = np.empty(plan.n_halo, np.complex128)
            reqs, bufs = [], []
            off = 0
            for p in range(plan.npeers):
                t = torch.empty(2 * int(plan.recv_counts[p]), dtype=torch.float64)
                reqs.append(dist.irecv(t, src=int(plan.peers[p])))
                bufs.append((off, t))
                off += int(plan.recv_counts[p])
            for p in range(plan.npeers):
                s = np.ascontiguousarray(xl[plan.send_rows[p]]).view(np.float64)
                reqs.append(dist.isend(torch.from_numpy(s.copy()), dst=int(plan.peers[p])))
            for r in reqs:
                r.wait()
            for o, t in bufs:
                halo[o:o + t.numel() // 2] = t.numpy().view(np.complex128)
            assert np.array_equal(halo, x[plan.halo_globals])  # every halo slot got the right global entry
            xe = np.concatenate([xl, halo])
            y = np.zeros(r1 - r0, np.complex128)
            rows = np.repeat(np.arange(r1 - r0), np.diff(lp))
            np.add.at(y, rows, lv * xe[plan.col_local])
            # rows in the "interior" range touch no halo column
            ib, ie = plan.interior
            for r in range(ib, ie):
                assert (plan.col_local[lp[r]:lp[r + 1]] < (r1 - r0)).all()
            results[kind] = dict(y=y, r0=r0, n_halo=plan.n_halo, peers=plan.peers.copy(), interior=plan.interior)
        else:
            from mgpreconditionedgcr_amd import DistSparse, Field, GCR, GCR_Param
            import mgpreconditionedgcr_amd as mg
            mg.init(0)
            A = DistSparse(comm, N, r0, lp, lc, lv)
            xf = Field((r1 - r0,), x[r0:r1])
            s0 = mg.stat("halo_split_exchanges")
            y = A(xf).to_numpy()
            n_split = mg.stat("halo_split_exchanges") - s0
            prev = mg.set_option("halo_split", 0)     # the same apply with the exchange completed before any row
            y_unsplit = A(xf).to_numpy()
            n_unsplit = mg.stat("halo_split_exchanges") - s0 - n_split
            mg.set_option("halo_split", prev)
            b = Field((r1 - r0,), problems.rhs_grid(N, 1)[r0:r1])
            xs = Field((r1 - r0,)).set_zero()
            t0 = mg.stat("pw_tail_folds")
            gcr = GCR(A, GCR_Param(0, 4, 25, 1e-30, False, check_every=5))
            gcr.solve(b, xs)
            n_tail = mg.stat("pw_tail_folds") - t0
            # the same solve with every fold + exchange as a launch of its own
            prev_t = mg.set_option("pw_tail", 0)
            g0 = GCR(A, GCR_Param(0, 4, 25, 1e-30, False, check_every=5))
            x0s = Field((r1 - r0,)).set_zero()
            g0.solve(b, x0s)
            mg.set_option("pw_tail", prev_t)
            hist_notail, x_notail = g0.last_history.copy(), x0s.to_numpy()
            g2 = GCR(A, GCR_Param(11, 0, 25, 1e-30, False))
            xt = Field((r1 - r0,)).set_zero()
            g2.solve(b, xt)
            results[kind] = dict(y=y, r0=r0, hist=gcr.last_history, x=xs.to_numpy(), its=gcr.last_iterations,
                                 hist_trunc=g2.last_history, x_trunc=xt.to_numpy(), format=A.storage_format()[0],
                                 allreduce=comm.allreduce_kind, halo=A.halo_kind, n_split=n_split, n_unsplit=n_unsplit, y_unsplit=y_unsplit,
                                 n_tail=n_tail, hist_notail=hist_notail, x_notail=x_notail)
            if kind == "poisson":
                import ctypes
                us = ctypes.c_double()
                mg.lib().mgcr_comm_bench_allreduce(comm.h, 11, 200, ctypes.byref(us))
                results[kind]["allreduce_us"] = us.value
    np.save(os.path.join(outdir, "rank%d.npy" % rank), results, allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
