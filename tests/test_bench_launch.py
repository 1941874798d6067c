"""bench.py's N-rank launch path, as far as it goes without GPUs: `--gpus 2` must START two ranks (by itself when no launcher set
WORLD_SIZE, or under torch.distributed.run — the driver's form, reference README.md:127 / trainer.py:478 /
src/utils/distributed.py:82-98), the ranks must find each other over loopback and agree on the world size. The collective
transport here is gloo (CVHIP_DIST_BACKEND=gloo: control flow only); on a GPU box the same path binds RCCL through the C ABI."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    e = dict(os.environ, CVHIP_DIST_BACKEND="gloo", CVHIP_BENCH_LAUNCH_TIMEOUT="240")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_USE_AGENT_STORE"):
        e.pop(k, None)
    return e


def _line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout     # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_flag_starts_the_ranks_itself():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["dry_launch"] is True and d["world"] == 2 and d["ranks_seen"] == 2


def test_runs_under_the_drivers_launcher():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"]
    r = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["world"] == 2 and d["ranks_seen"] == 2


def test_a_dead_rank_fails_the_launch():
    """a rank that cannot start must end the whole launch with a non-zero status (not hang the others in the rendezvous)"""
    e = _env()
    e["CVHIP_DIST_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=e, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
