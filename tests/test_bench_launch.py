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
