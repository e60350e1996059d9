#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (not the driver's bench contract —
that is bench.py).  One JSON line per workload on stdout.

  python bench_extra.py --workload sample     config 1: data/sample_matrix format, DiracOp k=0.15, GCR restart 5
  python bench_extra.py --workload poisson256 config 2 at 256^3: unpreconditioned GCR restart 5, 100 iterations
  python bench_extra.py --workload mg256      config 3: Poisson 256^3, 3-level aggregation MG (2^3 aggregates,
                                              piecewise-constant P, Galerkin), 2 GCR sweeps, flexible GCR restart 5
  python bench_extra.py --workload poisson128tol  config 2 time-to-tolerance: GCR restart 5 to 1e-13 on Poisson 128^3
                                              (bounded at 200 000 iterations), and the same solve MG-preconditioned
  python bench_extra.py --workload bcsr       config 5 (single GPU): unstructured HierarchicalSparse, bs = 20,
                                              skewed blocks/row, >= 2 GB of blocks: apply GB/s + GCR on it
"""
import argparse
import ctypes
import gzip
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def timed_solve(mg, gcr, rhs, x, warm=False):
    """Wall time of one solve.  warm: a first, untimed solve of the same system allocates the solver's work vectors
    (a solver object keeps them, like bench.py's warm-up does), then x is reset and the solve is timed."""
    if warm:
        x0 = x.copy()
        gcr.solve(rhs, x)
        x.assign(x0)
    mg.lib().mgcr_synchronize()
    t0 = time.perf_counter()
    gcr.solve(rhs, x)
    mg.lib().mgcr_synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=["sample", "poisson256", "mg256", "poisson128tol", "bcsr"])
    ap.add_argument("--grid", dest="n", type=int, default=256)
    ap.add_argument("--levels", type=int, default=2, help="number of coarse grids (2 = 3-level)")
    args = ap.parse_args()
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import (DiracOp, Field, GCR, GCR_Param, HierarchicalSparse, MG, MG_Param, Mesh, Sparse,
                                         problems, read_data)
    mg.init(0)
    out = {"workload": args.workload}
    if args.workload == "sample":
        d = tempfile.mkdtemp()
        with gzip.open(os.path.join(ROOT, "tests", "golden", "4x4parsed.txt.gz"), "rb") as fi, open(os.path.join(d, "4x4parsed.txt"), "wb") as fo:
            shutil.copyfileobj(fi, fo)
        D = read_data("4x4parsed.txt", directory=d)
        dirac = DiracOp(D, 0.15)
        g = np.load(os.path.join(ROOT, "tests", "golden", "sample_4x4.npz"))
        dims = (4, 4, 4, 4, 4, 3)
        rhs = Field(dims, g["gcr_rhs"])
        x = Field(dims).set_zero()
        gcr = GCR(dirac, GCR_Param(0, 5, 4000, 1e-13, False, check_every=20))
        timed_solve(mg, gcr, rhs, x)
        x.set_zero()
        dt = timed_solve(mg, gcr, rhs, x)
        ref = g["g3_restart5_hist"]
        out.update(iterations=gcr.last_iterations, seconds=dt, it_per_s=gcr.last_iterations / dt,
                   final_rel_residual=float(gcr.last_history[-1]), reference_iterations=int(ref.size - 1),
                   reference_final=float(ref[-1]),
                   max_rel_dev_first_60_steps=float(np.max(np.abs(gcr.last_history[1:60] - ref[1:60]) / ref[1:60])),
                   note="N = 3072: latency-bound (4 launches/iteration), the reference CPU path does ~1.7k it/s on this input")
    elif args.workload == "poisson128tol":
        n = 128
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        A = Sparse(N, ncol, rowptr, col, val)
        del rowptr, col, val
        dims = (n, n, n)
        rhs = Field(dims).fill_rhs(0)
        x = Field(dims).set_zero()
        tol = 1e-13
        cap = 200000
        gcr = GCR(A, GCR_Param(0, 5, cap, tol, False, check_every=50))
        dt = timed_solve(mg, gcr, rhs, x, warm=True)
        r = rhs - A(x)
        out.update(n=n, rows=N, tol=tol, iterations=gcr.last_iterations, converged=gcr.last_converged, seconds_to_tol=dt,
                   it_per_s=gcr.last_iterations / dt, final_rel_residual=float(gcr.last_history[-1]),
                   true_rel_residual=r.norm() / rhs.norm())
        # the same system and tolerance with the 3-level MG of config 3 as flexible right preconditioner
        prm = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                       2, None, None, null_vectors=np.ones((1, N), np.complex128))
        t0 = time.perf_counter()
        M = MG(A, prm)
        mg.lib().mgcr_synchronize()
        out["mg_setup_seconds"] = time.perf_counter() - t0
        x.set_zero()
        outer = GCR(A, GCR_Param(0, 5, 500, tol, False, None, M, flexible=True, check_every=2))
        dt = timed_solve(mg, outer, rhs, x, warm=True)
        r = rhs - A(x)
        out.update(mg_outer_iterations=outer.last_iterations, mg_converged=outer.last_converged, mg_seconds_to_tol=dt,
                   mg_true_rel_residual=r.norm() / rhs.norm())
    elif args.workload in ("poisson256", "mg256"):
        n = args.n
        t0 = time.perf_counter()
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        nnz = int(rowptr[-1])
        A = Sparse(N, ncol, rowptr, col, val)
        del rowptr, col, val
        out["setup_matrix_seconds"] = time.perf_counter() - t0
        dims = (n, n, n)
        rhs = Field(dims).fill_rhs(0)
        x = Field(dims).set_zero()
        V = 16 * N
        stored = A.stored_bytes()
        if args.workload == "poisson256":
            iters = 100
            gcr = GCR(A, GCR_Param(0, 5, 20, 0.0, False, check_every=20))
            timed_solve(mg, gcr, rhs, x)
            x.set_zero()
            prm = GCR_Param(0, 5, iters, 0.0, False, check_every=iters)
            gcr = GCR(A, prm)
            timed_solve(mg, gcr, rhs, x)          # allocates the work vectors
            x.set_zero()
            dt = timed_solve(mg, gcr, rhs, x)     # the timed solve: no events in the loop
            x.set_zero()
            prm.profile_spmv = True
            timed_solve(mg, gcr, rhs, x)          # a third one with hipEvents between the phases
            ph, na, fu = (ctypes.c_double * 3)(), ctypes.c_int32(), ctypes.c_int32()
            mg.lib().mgcr_gcr_last_profile(ph, ctypes.byref(na), ctypes.byref(fu))
            ms = ctypes.c_double(ph[1] / max(na.value, 1))   # operator apply (+ beta dots when fused)
            out["phase_us_per_iteration"] = [1e3 * v / max(na.value, 1) for v in ph]
            out["apply_fused_with_dots"] = bool(fu.value)
            b_stored = stored["matrix_bytes"] + 2 * V + (3 * V if fu.value else 0)
            b_survey = nnz * 20 + (N + 1) * 4 + 2 * V
            out.update(n=n, rows=N, nnz=nnz, iterations=iters, ms_per_iteration=dt * 1e3 / iters, it_per_s=iters / dt,
                       spmv_ms_in_situ=ms.value, spmv_GBps_stored_layout=b_stored / ms.value / 1e6,
                       spmv_frac_hbm_peak=b_stored / ms.value / 1e6 / HBM_PEAK_GBS,
                       spmv_GBps_survey_formula=b_survey / ms.value / 1e6,
                       iteration_bytes_survey=b_survey + 22 * V,
                       iteration_GBps_survey=(b_survey + 22 * V) / dt * iters / 1e9,
                       iteration_frac_hbm_peak_survey=(b_survey + 22 * V) / dt * iters / 1e9 / HBM_PEAK_GBS)
        else:
            t0 = time.perf_counter()
            prm = MG_Param(Mesh(dims), 2, 1, None, GCR(GCR_Param(0, 10, 50, 1e-2, False)), GCR(GCR_Param(0, 10, 2, 1e-30, False)),
                           args.levels, None, None, null_vectors=np.ones((1, N), np.complex128))
            M = MG(A, prm)
            out["setup_mg_seconds"] = time.perf_counter() - t0
            out["levels"] = [M.level_info(l) for l in range(args.levels + 1)]
            # one V-cycle
            y = Field(dims)
            M(rhs, out=y)
            mg.lib().mgcr_synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                M(rhs, out=y)
            mg.lib().mgcr_synchronize()
            out["vcycle_ms"] = (time.perf_counter() - t0) * 1e3 / reps
            # SURVEY.md §8(d) "algorithmic bytes — V-cycle": per level, nu_pre + nu_post smoother iterations
            # B_iter(lim) = B_spmv + (13 + 3 lim) V with lim = 1, 2, the residual B_spmv + 2 V, restrict
            # V_l + V_(l+1) and prolong+add V_(l+1) + 2 V_l (piecewise-constant P); the coarsest solve is not
            # counted (its iteration count is data dependent).  nnz_l = 7 n_l^3 - 6 n_l^2.
            model, nl = [], n
            for l in range(args.levels):
                Nl, nnzl = nl ** 3, 7 * nl ** 3 - 6 * nl ** 2
                Vl, Vc = 16 * Nl, 16 * (nl // 2) ** 3
                bsp = nnzl * 20 + (Nl + 1) * 4 + 2 * Vl
                smooth = 2 * sum(bsp + (13 + 3 * lim) * Vl for lim in (1, 2))
                model.append(dict(level=l, rows=Nl, smoother_bytes=smooth, residual_bytes=bsp + 2 * Vl, restrict_bytes=Vl + Vc,
                                  prolong_bytes=Vc + 2 * Vl))
                nl //= 2
            tot = sum(m["smoother_bytes"] + m["residual_bytes"] + m["restrict_bytes"] + m["prolong_bytes"] for m in model)
            out["vcycle_bytes_model_survey"] = model
            out["vcycle_GBps_survey"] = tot / (out["vcycle_ms"] * 1e-3) / 1e9
            out["vcycle_frac_hbm_peak_survey"] = out["vcycle_GBps_survey"] / HBM_PEAK_GBS
            # What THIS implementation moves per V-cycle on the levels above the coarsest (row-pattern storage: 2 B per
            # row + table per matrix pass; a smoother = 2 sweeps of lean GCR(10) whose last sweep stops after the x/r
            # update): pre-smoother from x0 = 0: [A r0 + its dots: M + 2V] + [xr 3V] + [A r + dots: M + 3V] + [build 4V]
            # + [xr 3V] + [x = sum: 3V] = 2M + 18V; no residual pass (the smoother's recurrence residual is restricted); restrict V + Vc; prolong+add Vc + 2V; post-smoother
            # from x: [b - A x: M + 3V] + [A r0 + dots, |b|^2: M + 3V] + 3V + [M + 3V] + 4V + 3V + [x += : 4V] = 3M + 23V.
            # The survey model above counts the reference layout (20 B per non-zero) and full last sweeps instead.
            moved, nl = 0, n
            for l in range(args.levels):
                Nl = nl ** 3
                Vl, Vc, Mb = 16 * Nl, 16 * (nl // 2) ** 3, 2 * Nl
                moved += 5 * Mb + 44 * Vl + 2 * Vc
                nl //= 2
            out["vcycle_bytes_moved_model_excl_coarsest"] = moved
            out["vcycle_GBps_moved_lower_bound"] = moved / (out["vcycle_ms"] * 1e-3) / 1e9   # the time includes the coarsest solve
            tol = 1e-8
            outer = GCR(A, GCR_Param(0, 5, 200, tol, False, None, M, flexible=True, check_every=2))
            dt = timed_solve(mg, outer, rhs, x, warm=True)
            r = rhs - A(x)
            out.update(n=n, rows=N, nnz=nnz, tol=tol, outer_iterations=outer.last_iterations, converged=outer.last_converged,
                       seconds_to_tol=dt, outer_it_per_s=outer.last_iterations / dt, history=[float(h) for h in outer.last_history],
                       true_rel_residual=r.norm() / rhs.norm())
            # unpreconditioned, same tolerance, for comparison (bounded)
            x.set_zero()
            plain = GCR(A, GCR_Param(0, 5, 3000, tol, False, check_every=50))
            dtp = timed_solve(mg, plain, rhs, x, warm=True)
            out.update(plain_iterations=plain.last_iterations, plain_converged=plain.last_converged, plain_seconds=dtp,
                       plain_final=float(plain.last_history[-1]))
    else:
        rng = np.random.default_rng(5)
        bs = 20
        nb = 36000  # block rows; ~9.6 blocks/row on average -> ~2.2 GB of blocks
        per_row = np.where(rng.random(nb) < 0.8, rng.integers(5, 10, nb), rng.integers(10, 65, nb))
        rows = np.repeat(np.arange(nb, dtype=np.int32), per_row)
        cols = rng.integers(0, nb, rows.size).astype(np.int32)
        # first block of every row is the diagonal block
        first = np.concatenate([[0], np.cumsum(per_row)[:-1]])
        cols[first] = np.arange(nb, dtype=np.int32)
        nblk = rows.size
        blocks = np.empty((nblk, bs, bs), np.complex128)
        chunk = 20000
        for s in range(0, nblk, chunk):
            e = min(nblk, s + chunk)
            blocks[s:e] = (rng.uniform(-1, 1, (e - s, bs, bs)) + 1j * rng.uniform(-1, 1, (e - s, bs, bs))) * (0.5 / bs)
        # diagonally dominant: diag block = I * (1 + sum of off-diagonal block norms)  
        offsum = np.bincount(rows, weights=np.abs(blocks).sum(axis=(1, 2)) / bs, minlength=nb)
        blocks[first] = np.eye(bs)[None] * (1.0 + offsum)[:, None, None]
        H = HierarchicalSparse(nb, nb, rows, cols, blocks)
        del blocks
        n = nb * bs
        xf = Field((n,)).fill_rhs(1)
        yf = Field((n,))
        ms = H.bench_apply(xf, yf, reps=20)
        b_alg = nblk * (bs * bs * 16 + 4) + (nb + 1) * 4 + 2 * n * 16
        out.update(block_rows=nb, bs=bs, blocks=int(nblk), blocks_per_row_min=int(per_row.min()), blocks_per_row_max=int(per_row.max()),
                   matrix_GB=nblk * bs * bs * 16 / 1e9, apply_ms=ms, algorithmic_bytes=b_alg, GBps=b_alg / ms / 1e6,
                   frac_hbm_peak=b_alg / ms / 1e6 / HBM_PEAK_GBS)
        rhs = Field((n,)).fill_rhs(2)
        x = Field((n,)).set_zero()
        gcr = GCR(H, GCR_Param(0, 5, 200, 1e-10, False, check_every=5))
        dt = timed_solve(mg, gcr, rhs, x, warm=True)
        r = rhs - H(x)
        out.update(gcr_iterations=gcr.last_iterations, gcr_converged=gcr.last_converged, gcr_seconds=dt,
                   gcr_it_per_s=gcr.last_iterations / dt, true_rel_residual=r.norm() / rhs.norm())
        # MG-GCR on the same operator (config 5 says "MG-GCR"): algebraic aggregates of 2 consecutive block rows, 2
        # near-null vectors by inverse iteration (10 GCR steps each), coarse operator = block-CSR again
        dims2 = (nb, bs)
        t0 = time.perf_counter()
        prm = MG_Param(Mesh(dims2), 2, 2, GCR_Param(0, 10, 10, 1e-8, False), GCR(GCR_Param(0, 10, 50, 1e-2, False)),
                       GCR(GCR_Param(0, 10, 2, 1e-30, False)), 1, None, None, spacetime=[True, False])
        M = MG(H, prm)
        mg.lib().mgcr_synchronize()
        out["mg_setup_seconds_incl_near_null_vectors"] = time.perf_counter() - t0
        out["mg_levels"] = [M.level_info(l) for l in range(2)]
        rhs2, x2 = Field(dims2), Field(dims2).set_zero()
        rhs2.assign(rhs)
        outer = GCR(H, GCR_Param(0, 5, 200, 1e-10, False, None, M, flexible=True, check_every=2))
        dt = timed_solve(mg, outer, rhs2, x2, warm=True)
        r2 = rhs2 - H(x2)
        out.update(mg_gcr_iterations=outer.last_iterations, mg_gcr_converged=outer.last_converged, mg_gcr_seconds=dt,
                   mg_gcr_true_rel_residual=r2.norm() / rhs2.norm())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
