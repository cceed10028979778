"""CPU: `python bench.py --gpus N` must start its own ranks when no launcher set WORLD_SIZE (the driver's plain
invocation).  --dry-run exercises the control plane only: child processes, gloo rendezvous on 127.0.0.1,
barrier, max-over-ranks, ONE JSON line from rank 0 -- no GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1",
                        *args], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    out = _run(args=("--gpus", "2"))
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1


def test_bench_self_launches_eight_ranks():
    """The driver's N = 8 invocation, control plane only: eight children, one JSON line, every rank seen."""
    out = _run(args=("--gpus", "8"))
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8


def test_bench_launcher_fails_fast_when_a_rank_dies_at_init():
    """Rank 3 exits 1 before the rendezvous: the parent must stop the other ranks and return non-zero within seconds, not
    sit in the rendezvous until somebody's time limit."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["RFI_BENCH_FAIL_RANK"] = "3"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "8"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 60, "the launcher waited for the dead rank"
    assert "rank 3" in r.stderr and "fault injection" in r.stderr, r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_single_rank_needs_no_rendezvous():
    out = _run(args=("--gpus", "1"))
    assert out["n_gpus"] == 1


def test_bench_under_an_external_launcher():
    """RANK/WORLD_SIZE already set (torch.distributed.run): bench.py must join, not spawn."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    assert json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])["n_gpus"] == 2
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # only rank 0 prints


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT") and
           not k.startswith("RFI_BENCH_") and not k.startswith("TORCHELASTIC_")}
    env.update(extra)
    return env


def test_bench_supervisor_falls_back_when_a_rank_stalls_after_init():
    """Rank 5 hangs right after the control-plane rendezvous (what a first bucketed all-reduce that never returns looks
    like from outside): the watchdog names the phase and the rank, kills the set and a FRESH set of ranks runs with
    RFI_NO_BUCKETS=1 RFI_NO_STOP_EVENTS=1; its line says so.  Well inside a minute."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "8"],
                       env=_clean_env(RFI_BENCH_STALL_RANK="5", RFI_BENCH_STALL_PHASE="control_plane", RFI_BENCH_PHASE_TIMEOUT="12"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 60
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                   # the failed attempt printed nothing
    out = json.loads(lines[0])
    assert out["fallback"] == "unbucketed" and out["n_ranks_seen"] == 8
    assert "attempt 1 (bucketed exchange) failed" in r.stderr and "'control_plane'" in r.stderr and "rank 5" in r.stderr, r.stderr[-1500:]


def test_bench_supervisor_gives_up_after_the_fallback_failed_too():
    """A rank that stalls in BOTH attempts (here: it dies at init each time): non-zero, both reasons and the stderr tails."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "4"],
                       env=_clean_env(RFI_BENCH_FAIL_RANK="2"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "attempt 2 (unbucketed fallback) failed" in r.stderr and "both attempts failed" in r.stderr, r.stderr[-1500:]


def _torchrun(nproc, port, **extra):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
                           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", str(nproc)],
                          env=_clean_env(**extra), capture_output=True, text=True, timeout=240)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_under_torch_distributed_run():
    """The driver's N > 1 command line: every rank of torch.distributed.run is a supervisor of ONE child; they agree through
    the launcher's own store."""
    r = _torchrun(2, _free_port())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_ranks_seen"] == 2 and json.loads(lines[0])["fallback"] is None


def test_bench_under_torch_distributed_run_falls_back_when_a_rank_stalls():
    r = _torchrun(2, _free_port(), RFI_BENCH_STALL_RANK="1", RFI_BENCH_STALL_PHASE="control_plane", RFI_BENCH_PHASE_TIMEOUT="12")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["fallback"] == "unbucketed", r.stdout
    assert "another rank's supervisor ended the attempt" in r.stderr or "rank 0 has been in the phase" in r.stderr


def test_bench_supervisor_attempt_limit_runs_from_the_end_of_the_imports():
    """No phase bound is hit (defaults: 45-200 s) but the attempt's own limit is: rank 2 hangs after the rendezvous, the limit
    of 8 s -- counted from the moment every child has finished its imports, however long those took -- ends the attempt and
    the fresh unbucketed set finishes inside RFI_BENCH_TOTAL_TIMEOUT."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "4"],
                       env=_clean_env(RFI_BENCH_STALL_RANK="2", RFI_BENCH_STALL_PHASE="control_plane", RFI_BENCH_LAUNCH_TIMEOUT="8",
                                      RFI_BENCH_TOTAL_TIMEOUT="120"), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 100
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["fallback"] == "unbucketed", r.stdout
    assert "after the imports" in r.stderr and "attempt 1 (bucketed exchange) failed" in r.stderr, r.stderr[-1500:]
