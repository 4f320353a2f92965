"""CPU (gloo): `python bench.py --gpus 2` started WITHOUT a launcher spawns its own ranks (the form the driver uses), the torchrun form
keeps working, and rank 0 prints exactly one JSON line.  `--dry-run` swaps the kernels for host arithmetic; rendezvous, the bucketed
reducer, barrier-bracketed timing and the MAX over ranks are the code the GPU run uses."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_bench_spawns_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["dry_run"] is True


def test_bench_under_torchrun():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_bench_dry_run_with_eight_ranks_and_reduce_scatter():
    """The launch form of the 8-GPU scaling run, on CPU: 8 ranks, the flat gradient buffer of the real config-2 parameter list in 32 MiB buckets,
    reduce-scatter + all-gather per bucket, the packed EMA statistics exchange and the CLIs' file sharding at world 8 (bench.py: dry_run)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--dry-run", "--ddp-mode",
                        "reduce_scatter"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["dry_run"] is True
    assert lines[0]["comm"]["mode"] == "reduce_scatter" and lines[0]["comm"]["buckets"] >= 3 and lines[0]["comm"]["bytes_per_step"] > 100e6
