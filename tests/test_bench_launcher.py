"""bench.py's process structure (no GPU): `python bench.py --gpus N` with WORLD_SIZE unset must start its N ranks
itself, each rank supervising a worker child it can restart with a more conservative transport; a failed rank must
surface as a non-zero exit.  The launcher's command / environment construction is a pure function and is checked here;
the supervisor ladder is run end to end over gloo with a stand-in worker."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launcher_plan_builds_one_rank_per_gpu():
    plan = bench.launcher_plan(4, ["--gpus", "4", "--steps", "7"], 29511, base_env={"PATH": "/usr/bin", "MGCR_BENCH_ROLE": "worker"})
    assert len(plan) == 4
    for r, (cmd, env) in enumerate(plan):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "7"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == "4" and env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"     # dmabuf IPC: hipIpc mailboxes and RCCL need it on this pool
        assert "MGCR_BENCH_ROLE" not in env                 # the ranks are supervisors, not workers
        assert env["PATH"] == "/usr/bin"


def test_worker_command_carries_the_transport_mode():
    cmd, env = bench.worker_command(["--gpus", "2"], 1234, {"MGCR_PEER_ALLREDUCE": "0"}, base_env={"RANK": "1", "WORLD_SIZE": "2"})
    assert env["MGCR_BENCH_ROLE"] == "worker" and env["MASTER_PORT"] == "1234" and env["MGCR_PEER_ALLREDUCE"] == "0"
    assert env["RANK"] == "1" and cmd[-2:] == ["--gpus", "2"]
    # the ladder ends with the host-staged transport and starts with the library's own choice
    assert bench.LADDER[0][1] == {} and bench.LADDER[-1][1] == {"MGCR_BENCH_TRANSPORT": "host"}
    assert len(bench.ATTEMPT_TIMEOUT_S) == len(bench.LADDER)


def test_last_json_line_and_stats():
    assert bench.last_json_line("noise\n{\"a\": 1}\nwarning: x\n") == {"a": 1}
    assert bench.last_json_line("nothing here") is None
    s = bench.stats([3.0, 1.0, 2.0, 10.0])
    assert s == {"median": 2.5, "min": 1.0, "max": 10.0, "repetitions": 4}


def test_traffic_is_only_reported_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    sha = bench.source_sha16()
    assert len(sha) == 16 and sha == bench.source_sha16()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "mgpreconditionedgcr_amd" / "csrc")
    (tmp_path / "mgpreconditionedgcr_amd" / "csrc" / "a.hip").write_text("kernel v1")
    now = bench.source_sha16()
    json.dump({"n": 128, "src_sha16": now, "phase_hbm_bytes_per_launch": {"build": 5.0}}, open(tmp_path / "profiles" / "pmc_traffic.json", "w"))
    assert bench.pmc_traffic("build", 128)[0] == 5.0
    assert bench.pmc_traffic("build", 256)[0] is None
    (tmp_path / "mgpreconditionedgcr_amd" / "csrc" / "a.hip").write_text("kernel v2")   # a kernel changed: stale
    val, note = bench.pmc_traffic("build", 128)
    assert val is None and "other kernel sources" in note


def _run_ladder(tmp_path, fail_modes, hang_modes=()):
    """Two supervisors over gloo with a stand-in worker that fails in the given transport modes."""
    fake = tmp_path / "fake_bench.py"
    fake.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        import bench
        if os.environ.get("MGCR_BENCH_ROLE") == "worker":
            mode = "host" if os.environ.get("MGCR_BENCH_TRANSPORT") == "host" else "rccl" if os.environ.get("MGCR_PEER_ALLREDUCE") == "0" else "default"
            if mode in %r and os.environ["RANK"] == "1":     # only ONE rank fails: every rank must move on together
                sys.exit(3)
            if mode in %r and os.environ["RANK"] == "0":     # ... even when the other one would wait for it for minutes
                import time
                time.sleep(120)
            if os.environ["RANK"] == "0":
                print(json.dumps({"metric": "gcr_iterations_per_sec", "value": 1.0, "mode": mode}))
            sys.exit(0)
        bench.__file__ = os.path.abspath(__file__)
        sys.exit(bench.run_supervisor(None, sys.argv[1:]))
    """ % (ROOT, list(fail_modes), list(hang_modes))))
    port = bench.free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("MGCR_BENCH_ROLE", None)
        procs.append(subprocess.Popen([sys.executable, str(fake), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    return [p.returncode for p in procs], outs


def test_supervisors_fall_back_together(tmp_path):
    rcs, outs = _run_ladder(tmp_path, ["default"])
    assert rcs == [0, 0], outs
    d = bench.last_json_line(outs[0][0])
    assert d["mode"] == "rccl"
    att = d["launch"]["attempts"]
    assert [a["ok"] for a in att] == [False, True] and att[0]["ranks_failed"] == 1 and d["launch"]["ranks"] == 2
    assert bench.last_json_line(outs[1][0]) is None        # exactly one line, from rank 0


def test_supervisors_exit_non_zero_when_nothing_works(tmp_path):
    rcs, outs = _run_ladder(tmp_path, ["default", "rccl", "host"])
    assert rcs == [1, 1], outs
    assert bench.last_json_line(outs[0][0]) is None


def test_a_rank_that_dies_at_start_up_ends_the_attempt_for_everybody(tmp_path):
    """Rank 1's worker exits at once, rank 0's would wait two minutes (a rendezvous nobody else joins): the attempt ends within
    seconds on both and the ladder moves on."""
    import time
    t0 = time.time()
    rcs, outs = _run_ladder(tmp_path, ["default"], hang_modes=["default"])
    assert time.time() - t0 < 60, time.time() - t0
    assert rcs == [0, 0], outs
    d = bench.last_json_line(outs[0][0])
    assert d["mode"] == "rccl" and [a["ok"] for a in d["launch"]["attempts"]] == [False, True]


def test_launcher_ends_the_workers_of_ranks_it_kills(tmp_path, monkeypatch):
    """At its deadline the launcher ends every rank's whole process GROUP: the GPU-holding worker is the supervisor's child
    and must not outlive it (a killed supervisor used to leave it running with the device)."""
    import time
    import types
    pidfile = tmp_path / "grandchild.pid"
    rank = tmp_path / "rank.py"
    rank.write_text(textwrap.dedent("""
        import subprocess, sys, time
        p = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"])   # the stand-in worker
        open(%r, "w").write(str(p.pid))
        time.sleep(300)                                                                # the supervisor hangs
    """ % str(pidfile)))
    monkeypatch.setattr(bench, "launcher_plan", lambda n, argv, port, base_env=None: [([sys.executable, str(rank)], dict(os.environ))] * n)
    monkeypatch.setenv("MGCR_BENCH_LAUNCHER_DEADLINE_S", "3")
    t0 = time.time()
    rc = bench.run_launcher(types.SimpleNamespace(gpus=2), [])
    assert rc == 1 and time.time() - t0 < 30
    pid = int(pidfile.read_text())
    for _ in range(50):        # the grandchild is gone (reaped by init) — not merely orphaned
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        if open("/proc/%d/stat" % pid).read().split()[2] == "Z":
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the worker outlived its supervisor")


def test_worker_environment_drops_the_launcher_agents_variables():
    """Under torch.distributed.run the rank processes (bench.py's supervisors) carry TORCHELASTIC_USE_AGENT_STORE=True: a worker that
    inherited it would wait for the agent's store at its own fresh rendezvous port — where nobody listens — until the attempt's time
    limit (seen on the GPU box: the default attempt of `torchrun ... bench.py --gpus 2` hung for its 420 s).  The workers get none of
    the agent's variables and rendezvous over an explicit tcp:// address."""
    import bench
    base = dict(PATH="/usr/bin", RANK="1", WORLD_SIZE="2", LOCAL_RANK="1", MASTER_ADDR="10.0.0.1", MASTER_PORT="29511",
                TORCHELASTIC_USE_AGENT_STORE="True", TORCHELASTIC_RUN_ID="none", TORCHELASTIC_RESTART_COUNT="0", GROUP_RANK="0", ROLE_RANK="1")
    cmd, env = bench.worker_command(["--gpus", "2"], 40123, {"MGCR_PEER_ALLREDUCE": "0"}, base_env=base)
    assert not any(k.startswith("TORCHELASTIC_") for k in env) and "GROUP_RANK" not in env and "ROLE_RANK" not in env
    assert env["MASTER_PORT"] == "40123" and env["MASTER_ADDR"] == "127.0.0.1" and env["RANK"] == "1" and env["WORLD_SIZE"] == "2"
    assert env["MGCR_BENCH_ROLE"] == "worker" and env["MGCR_PEER_ALLREDUCE"] == "0"
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT")}
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="40123")
        assert bench.worker_init_method() == "tcp://127.0.0.1:40123"
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_workers_rendezvous_under_torch_distributed_run(tmp_path):
    """The same on CPU, end to end: two rank processes started by torch.distributed.run each start a child the way bench.py's
    supervisors start their workers (worker_command: fresh port, the agent's variables dropped); the children meet over gloo with
    bench.worker_init_method() and all-reduce a number — within seconds, not at a time limit."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        dist.init_process_group("gloo", init_method=bench.worker_init_method(), rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.tensor([float(os.environ["RANK"]) + 1.0])
        dist.all_reduce(t)
        print("child", os.environ["RANK"], float(t[0]), flush=True)
        dist.destroy_process_group()
    """ % root))
    parent = tmp_path / "parent.py"
    parent.write_text(textwrap.dedent("""
        import os, subprocess, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        rank = int(os.environ["RANK"])
        dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
        port = torch.tensor([bench.free_port() if rank == 0 else 0], dtype=torch.int64)
        dist.broadcast(port, src=0)
        cmd, env = bench.worker_command([], int(port[0]), {})
        p = subprocess.run([sys.executable, %r], env=env, capture_output=True, text=True, timeout=120)
        print("parent", rank, p.returncode, p.stdout.strip(), p.stderr[-300:] if p.returncode else "", flush=True)
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(p.returncode)
    """ % (root, str(child))))
    port = __import__("bench").free_port()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(parent)], capture_output=True, text=True, timeout=240, cwd=root)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "child 0 3.0" in p.stdout and "child 1 3.0" in p.stdout, p.stdout
