"""N > 1 path on CPU (gloo, world_size 2 and 3): the partition plan of libmgcr_hip.so (pure host
code, no GPU) — halo lists, send lists, local column numbering, interior range — driven through
real inter-process exchanges, checked against the global SpMV."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_workers(mode, world, outdir, timeout=300):
    """Starts `world` worker processes (tests/dist_worker.py) and collects what they saved.  The ranks meet in collectives: when one of
    them fails, the others would wait for it until their own time limits — so the first non-zero exit ends the run at once, with THAT
    rank's output in the assertion (a failing rank must show its cause, not a time-out of the ranks that waited for it)."""
    import tempfile
    import time
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    logs = [tempfile.TemporaryFile(mode="w+") for _ in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(r), str(world),
                               str(port), str(outdir)], env=env, stdout=logs[r], stderr=subprocess.STDOUT)
             for r in range(world)]

    def output(r):
        logs[r].seek(0)
        return logs[r].read()[-3000:]
    t0 = time.time()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() - t0 > timeout:
            failed = bad[0] if bad else -1
            for q in procs:          # the exact processes started above
                if q.poll() is None:
                    q.kill()
            for q in procs:
                q.wait()
            break
        time.sleep(0.05)
    if failed is None:
        bad = [r for r, p in enumerate(procs) if p.returncode != 0]
        failed = bad[0] if bad else None
    try:
        assert failed is None, ("rank %d failed:\n%s" % (failed, output(failed))) if failed >= 0 else (
            "no rank finished within %d s; rank 0 so far:\n%s" % (timeout, output(0)))
    finally:
        for f in logs:
            f.close()
    return [np.load(os.path.join(str(outdir), "rank%d.npy" % r), allow_pickle=True).item() for r in range(world)]


@pytest.mark.parametrize("world", [2, 3, 6])
def test_partition_plan_gloo(tmp_path, world):
    from tests.dist_worker import problem
    from mgpreconditionedgcr_amd import problems
    res = run_workers("plan", world, tmp_path)
    for kind in ("poisson", "random"):
        N, rowptr, col, val, gran = problem(kind)
        x = problems.rhs_grid(N, 5)
        y = np.zeros(N, np.complex128)
        np.add.at(y, np.repeat(np.arange(N), np.diff(rowptr)), val * x[col])
        got = np.concatenate([res[r][kind]["y"] for r in range(world)])
        assert np.allclose(got, y, rtol=1e-14, atol=1e-14)
        if kind == "poisson":
            for r in range(world):
                nb = [q for q in (r - 1, r + 1) if 0 <= q < world]
                assert list(res[r][kind]["peers"]) == nb          # slab partition: nearest neighbours only
                assert res[r][kind]["n_halo"] == 36 * len(nb)     # one 6x6 plane per neighbour
                ib, ie = res[r][kind]["interior"]
                planes = 6 // world
                assert (ie - ib) == 36 * max(planes - len(nb), 0)   # all but the first/last plane (6 ranks: one plane each, every row of a middle rank touches a halo column)
