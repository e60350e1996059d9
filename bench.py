#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: restarted GCR iterations/sec and SpMV HBM GB/s
on the 3-D 7-point Poisson system of BASELINE.json configs[1] (128^3, unpreconditioned GCR
restart 5, complex fp64, x0 = 0, deterministic RHS), one process per GPU.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one GCR iteration (1 SpMV + the fused orthogonalisation kernels).  The timed region
is one mgcr_gcr_solve call limited to exactly K iterations (tol = 0 so it cannot stop early),
inputs resident in HBM, bracketed by barrier + device synchronisation; the max over ranks is
reported.  N > 1: the grid grows along i to (128 N) x 128 x 128 and is slab-partitioned, 128
planes (= the N=1 problem) per GPU — weak scaling; value = N * iterations/s (shard-iterations/s).

One JSON line on stdout (rank 0), with `roofline` (the phase of the iteration that takes longest,
hipEvent-timed on the library's own stream inside a solve) and `cpu_baseline` (the real reference, oracle/_ref/ref_harness, on this box's host
cores; falls back to the oracle port when that binary is absent) plus `cpu_baseline_optimised` (an OpenMP,
fused-pass CPU port on all host cores).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def spmv_algorithmic_bytes(nnz, nrow, ncol):
    """complex-fp64 values + int32 columns + int32 row pointers + x read once
    + y written once."""
    return nnz * 20 + (nrow + 1) * 4 + ncol * 16 + nrow * 16


def cpu_baseline(n, iters=10):
    """Time the reference CPU path on this box's host cores (bounded sample: `iters` GCR
    iterations of the same system, 1 thread — the reference's Sparse/Field path is single
    threaded, src/Operator.h:330-346, src/Fields.h:192-308)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    ncores = os.cpu_count() or 1
    if os.path.exists(exe):
        out = subprocess.run([exe, "/tmp", "bench", str(n), str(iters)], capture_output=True, text=True, timeout=900)
        for line in out.stdout.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                return {"value": d["it_per_s"], "unit": "it/s", "cores": 1, "kind": "reference",
                        "sample": "%d GCR iterations (restart 5) of the same Poisson %d^3 system by the real reference "
                                  "(oracle/_ref/ref_harness, g++ -O3), 1 thread of %d host cores" % (iters, n, ncores),
                        "spmv_seconds": d["spmv_seconds"], "host_cores": ncores}
    # fallback: the oracle port (same operation order, fewer temporaries => faster than the reference)
    import numpy as np  # noqa: F401
    from oracle import oracle as orc
    N, rowptr, col, val = orc.poisson3d(n)
    A = orc.csr(N, N, rowptr, col, val)
    b = orc.fill_rhs(N, 0)
    t0 = time.perf_counter()
    orc.gcr_solve(A, orc.gcr_param(restart=5, max_iter=iters, tol=0.0), b)
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "it/s", "cores": 1, "kind": "port",
            "sample": "%d GCR iterations (restart 5) of the same Poisson %d^3 system by oracle/mgcr_oracle.c, 1 thread"
                      % (iters, n), "host_cores": ncores}


def cpu_baseline_optimised(n, restart, seconds=8.0):
    """SURVEY.md §8(d) "optimised CPU" row: the OpenMP port with fused passes (oracle/mgcr_cpu_opt.c) on all
    of this box's host cores, bounded to about `seconds` of work."""
    from oracle import oracle as orc
    ncores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = ncores
    # The visible core count is not the usable one: a 1-GPU box exposes 256 cores and its cgroup grants 16
    # (cpu.max "1600000 100000"); more threads than that are throttled, not run.
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                usable = max(1, min(usable, int(q / per + 0.5)))
        except Exception:
            pass
    dt, _ = orc.opt_gcr_poisson(n, restart, 5, usable)          # page-in + a first rate estimate
    iters = int(max(10, min(500, seconds / max(dt / 5, 1e-6))))
    dt, hist = orc.opt_gcr_poisson(n, restart, iters, usable)
    return {"value": iters / dt, "unit": "it/s", "cores": usable, "kind": "port",
            "sample": "%d GCR iterations (restart %d) of the same Poisson %d^3 system by oracle/mgcr_cpu_opt.c: OpenMP over "
                      "%d threads (= the CPU share this process is granted), CSR with int32 columns and real values, update / "
                      "SpMV+dots / build fused as on the GPU" % (iters, restart, n, usable),
            "final_rel_residual": float(hist[-1]), "host_cores": ncores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", dest="n", type=int, default=128, help="Poisson grid edge per GPU shard (128 = BASELINE configs[1])")
    ap.add_argument("--restart", type=int, default=5)
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, world, args.gpus))

    import numpy as np
    import torch
    import mgpreconditionedgcr_amd as mg
    from mgpreconditionedgcr_amd import Field, GCR, GCR_Param, Sparse, problems

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    # bring-up switch: MGCR_BENCH_TRANSPORT=host runs the N > 1 code path with every rank on GPU 0 and
    # the host-staged (gloo) transport, so that it can be rehearsed on a one-GPU box; numbers
    # obtained that way are not benchmark results
    host_transport = os.environ.get("MGCR_BENCH_TRANSPORT", "rccl") == "host"
    if host_transport:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    mg.init(local_rank)
    dist = None
    n = args.n
    if world > 1:
        # control plane (barriers, id broadcast, max-reduce of the timing) over gloo; the data path
        # (halo exchange + dot-product all-reduces) is RCCL inside libmgcr_hip.so
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from mgpreconditionedgcr_amd import Comm, DistSparse
        rccl_note = None
        if host_transport:
            comm = Comm.host(dist)
        else:
            # RCCL communicator; should its creation fail on any rank, every rank falls back to the host-staged
            # transport so that the run still yields a (flagged) line instead of nothing
            try:
                comm, err = Comm.rccl(dist), None
            except Exception as e:  # noqa: BLE001
                comm, err = None, repr(e)
            bad = torch.tensor([0 if err is None else 1], dtype=torch.int32)
            dist.all_reduce(bad)
            if int(bad[0]):
                rccl_note = "RCCL communicator creation failed on %d rank(s) (%s): host-staged transport" % (int(bad[0]), err)
                comm = Comm.host(dist)
        # weak scaling: the grid grows along i, every GPU owns n planes (= the N=1 problem)
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n, rank * n, (rank + 1) * n, ni=world * n)
        nnz = int(rowptr[-1])
        A = DistSparse(comm, ncol, rank * N, rowptr, col, val)
        ncol = N  # per-shard accounting below
    else:
        N, ncol, rowptr, col, val = problems.poisson3d_csr(n)
        nnz = int(rowptr[-1])
        A = Sparse(N, ncol, rowptr, col, val)
    del rowptr, col, val
    dims = (n, n, n)
    rhs = Field(dims).fill_rhs(0, global_offset=rank * N)
    x = Field(dims)

    def barrier():
        if dist is not None:
            dist.barrier()

    # one solver object for all runs: its work vectors are allocated by the warm-up solve and re-used
    # (GCR_Param is re-read at every solve, like the reference's GCR does with its GCR_Param*)
    gparam = GCR_Param(0, args.restart, 1, 0.0, False)
    gcr = GCR(A, gparam)

    def run(iters, profile=False):
        x.set_zero()
        gparam.max_iter, gparam.check_every, gparam.profile_spmv = iters, max(iters, 1), profile
        torch.cuda.synchronize()
        mg.lib().mgcr_synchronize()
        barrier()
        t0 = time.perf_counter()
        gcr.solve(rhs, x)
        mg.lib().mgcr_synchronize()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:  # max over ranks
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        assert gcr.last_iterations == iters, (gcr.last_iterations, iters)
        return dt, gcr

    run(max(args.warmup, 1))  # at least one untimed solve: it allocates the solver's work vectors
    dt, gcr = run(args.steps)
    hist = gcr.last_history
    ms_per_step = dt * 1e3 / args.steps
    it_per_s = args.steps / dt

    # Per-phase timing IN SITU: a third solve of the same length with hipEvents (library stream) between
    # the phases of every iteration — back-to-back replays of one kernel would be served from the
    # 256 MiB Infinity Cache once its operands fit, which at this size they do.
    import ctypes
    run(args.steps, profile=True)
    ph_ms = (ctypes.c_double * 3)()
    n_it, fused = ctypes.c_int32(), ctypes.c_int32()
    mg.lib().mgcr_gcr_last_profile(ph_ms, ctypes.byref(n_it), ctypes.byref(fused))
    ph_us = [1e3 * v / max(n_it.value, 1) for v in ph_ms]           # average microseconds per iteration
    y = Field(dims)
    spmv_ms_replay = A.bench_apply(rhs, y, reps=args.spmv_reps)
    # The stand-alone SpMV with COLD caches (the metric's "SpMV GB/s vs HBM roofline"): between two applies a
    # 256 MiB copy sweeps L2 and the Infinity Cache, the apply alone sits between the library's stream events.
    spmv_ms_cold = None
    if world == 1:
        nf = 16 * 1024 * 1024                      # 2 x 256 MiB of complex fp64
        fa, fb = Field((nf,)).set_zero(), Field((nf,))
        t_c = ctypes.c_double()
        tot = 0.0
        for _ in range(args.spmv_reps):
            fb.assign(fa)
            mg.lib().mgcr_timer_start()
            A(rhs, out=y)
            mg.lib().mgcr_timer_stop(ctypes.byref(t_c))
            tot += t_c.value
        spmv_ms_cold = tot / args.spmv_reps
        del fa, fb
    stored = A.stored_bytes()
    fmt, npat = A.storage_format()
    V = 16 * N
    R = args.restart
    # Bytes each phase has to move per launch with the layout actually stored ("if the
    # implementation stores something else ... it must report with its stored sizes"), averaged over a
    # restart cycle (lim = number of stored directions the step orthogonalises against):
    #   xr      r, Ap read + r written                                              3 V
    #   apply   matrix (pattern ids + tables, or slab) + r read + Ar written + lim Aps_j read (+ Ar re-read
    #           by the separate multidot kernel when the fused kernel is not used)
    #   build   in-cycle (3 + lim) V, lim = 1..R-1;  cycle-closing step (2R + 6) V
    # exact for the K iterations that were timed: iteration k of a cycle orthogonalises against lim = k stored
    # directions, k = 1..R, and k = R closes the cycle
    lims = [((k - 1) % R) + 1 for k in range(1, max(n_it.value, 1) + 1)]
    b_apply = [stored["matrix_bytes"] + 16 * ncol + 16 * N + l * V + (0 if fused.value else V) + (V if l > 8 else 0) for l in lims]
    b_build = [(2 * R + 6) * V if l == R else (3 + l) * V for l in lims]
    b_phase = [3.0 * V, sum(b_apply) / len(lims), sum(b_build) / len(lims)]
    names = ["xr_update_kernel (alpha, residual ring, |r|^2)",
             "step_apply_kernel (SpMV + beta dot products, one kernel)" if fused.value else "SpMV + multidot_kernel",
             "build_lean_kernel<1..%d> / build_close_kernel<%d> (direction build + x update)" % (R - 1, R)]
    keys = ["xr", "apply_dots", "build"]
    dom = max(range(3), key=lambda k: ph_us[k])
    achieved = b_phase[dom] / (ph_us[dom] * 1e-6) / 1e9
    traffic = None
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(prof):
        try:
            d = json.load(open(prof))
            if d.get("n") == n and world == 1:
                traffic = d["phase_hbm_bytes_per_launch"].get(keys[dom])
        except Exception:
            traffic = None
    b_spmv_survey = spmv_algorithmic_bytes(nnz, N, ncol)
    mean_lim = sum(lims) / float(len(lims))
    iter_bytes_survey = b_spmv_survey + (13 + 3 * mean_lim) * V   # SURVEY.md §8(d) accounting
    iter_bytes_ours = sum(b_phase)                                # what this implementation moves

    out = {
        "metric": "gcr_iterations_per_sec", "value": it_per_s * world, "unit": "it/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "value_definition": "iterations/s of ONE solve on one GPU" if world == 1 else
                            "iterations/s of the ONE distributed solve (%.1f) x n_gpus: every iteration sweeps n_gpus shards of "
                            "%d^3 rows, the weak-scaling aggregate (shard-iterations per second)" % (it_per_s, n),
        "config": {"workload": "3D 7-point Poisson %d^3 per GPU, unpreconditioned GCR restart %d, complex fp64, x0=0, "
                               "RHS splitmix64 seed 0" % (n, args.restart),
                   "rows": N, "nnz": nnz, "complex": True,
                   "matrix_storage": {0: "ELL slab",
                                      1: "row-pattern dictionary, %s patterns (2 B per row + table)" % npat,
                                      2: "row-pattern dictionary for the columns (%s patterns) + value slab" % npat}[fmt],
                   "stored_matrix_bytes": stored["matrix_bytes"], "ell_width": stored["ell_width"], "tail_nnz": stored["tail_nnz"],
                   "partition": "1 GPU" if world == 1 else "slab x%d (grid %dx%dx%d), %s" % (
                       world, world * n, n, n, "host-staged transport (bring-up, not a result)" if host_transport
                       else (rccl_note or "RCCL communicator; halo: %s, all-reduce: %s" % (A.halo_kind, comm.allreduce_kind)))},
        "phases": {keys[k]: {"kernel": names[k], "us_per_iteration": ph_us[k], "bytes_per_launch": b_phase[k],
                             "GBps": b_phase[k] / (ph_us[k] * 1e-6) / 1e9 if ph_us[k] > 0 else None} for k in range(3)},
        "phases_timed": "in situ: hipEvents between the phases of each of the %d iterations of a GCR solve" % n_it.value,
        "spmv": {"kernel": "stand-alone operator apply (inside the solver loop it runs fused with the beta dot products: phases.apply_dots)",
                 "ms_cold_caches": spmv_ms_cold, "ms_back_to_back_replay": spmv_ms_replay, "includes_halo_exchange": world > 1,
                 "bytes_moved_stored_layout": stored["matrix_bytes"] + 16 * ncol + 16 * N,
                 "GBps": None if not spmv_ms_cold else (stored["matrix_bytes"] + 16 * ncol + 16 * N) / spmv_ms_cold / 1e6,
                 "frac_hbm_peak": None if not spmv_ms_cold else (stored["matrix_bytes"] + 16 * ncol + 16 * N) / spmv_ms_cold / 1e6 / HBM_PEAK_GBS,
                 "algorithmic_bytes_survey_formula": b_spmv_survey,
                 "GBps_survey_formula": None if not spmv_ms_cold else b_spmv_survey / spmv_ms_cold / 1e6},
        "iteration": {"bytes_moved_model": iter_bytes_ours, "GBps": iter_bytes_ours / (ms_per_step * 1e-3) / 1e9,
                      "frac_hbm_peak": iter_bytes_ours / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "algorithmic_bytes_survey": iter_bytes_survey,
                      "GBps_survey": iter_bytes_survey / (ms_per_step * 1e-3) / 1e9},
        "roofline": {"kernel": names[dom], "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": b_phase[dom],
                     "us_per_launch": ph_us[dom],
                     "note": "dominant phase of the iteration by time; achieved = bytes of the stored layout the phase's "
                             "kernel has to move per launch (average over a restart cycle) / its hipEvent-timed duration"
                             + ("" if world == 1 else "; N > 1: the phase times include the halo exchange (apply_dots) and both "
                                                      "all-reduces of the iteration (build)")},
        "final_rel_residual": float(hist[-1]),
    }
    if world > 1:
        # what the iteration's two all-reduces (4 doubles; 1 + 2 lim doubles) and its halo exchange cost on their own
        us = ctypes.c_double()
        comm_us = {}
        for cnt in (4, 1 + 2 * args.restart):
            mg.lib().mgcr_comm_bench_allreduce(comm.h, cnt, 50, ctypes.byref(us))
            comm_us["allreduce_%d_doubles_us" % cnt] = us.value
        out["comm"] = comm_us
        out["comm"]["allreduce_kind"] = comm.allreduce_kind
        out["comm"]["halo_kind"] = A.halo_kind
        out["comm"]["spmv_with_halo_exchange_ms_replay"] = spmv_ms_replay
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(n)
        except Exception as e:  # the baseline leg must never take the GPU numbers down with it
            out["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}
        try:
            out["cpu_baseline_optimised"] = cpu_baseline_optimised(n, args.restart)
        except Exception as e:
            out["cpu_baseline_optimised"] = {"value": None, "unit": "it/s", "cores": None, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    # orderly teardown: solver state and operator first, then the communicator, then the process group
    del gcr, x, rhs, y
    del A
    if dist is not None:
        dist.barrier()
        lib = mg.lib()
        lib.mgcr_synchronize()
        lib.mgcr_comm_destroy(comm.h)
        comm.h = None
        dist.destroy_process_group()
    mg.finalize()


if __name__ == "__main__":
    main()
