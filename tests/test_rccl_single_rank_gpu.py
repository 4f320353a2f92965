"""GPU, RCCL ("nccl" backend) on a ONE-rank process group: a one-GPU box cannot hold two RCCL ranks (RCCL refuses duplicate devices), so
`tests/test_ddp_gpu.py` covers the two-rank arithmetic over gloo and THIS test covers what gloo cannot: that the collectives the product issues
-- `GradReducer`'s bucketed fp32 all-reduces on its side stream while weight-gradient kernels are still being queued, the quantizer's packed
[counts | dw] statistics all-reduce ahead of `sa_vq_ema_update`, `bench.py`'s barrier / MAX reductions -- are accepted by RCCL on gfx950 (buffer
views into the flat gradient buffer, stream / event ordering, communicator set-up and teardown) and leave the step BIT-identical to the run
without a process group (a one-rank SUM is the identity).  `SA_DDP_SINGLE_RANK` (debug.host("ddp_single_rank")) makes the world-size-1 short cuts
issue their collectives.

Reference behaviour: run_vqvae.py:71-77 (DDP gradient averaging), src/networks/vqvae/baseline.py:66-80 (all-reduced EMA statistics)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_CHILD = r"""
import os, sys, torch
sys.path.insert(0, os.environ["SA_ROOT"])
sys.path.insert(0, os.path.join(os.environ["SA_ROOT"], "tests"))
import torch.distributed as dist
from test_ddp_gpu import _vqvae_steps, _performer_steps
use = os.environ.get("WORLD_SIZE") == "1" and os.environ.get("SA_DDP_SINGLE_RANK") is not None
if use:
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)          # communicator set-up happens here
    dist.barrier()
    assert float(t.sum()) == 4.0
res = dict(vq32=_vqvae_steps(0, 1), vq16=_vqvae_steps(0, 1, torch.bfloat16), perf=_performer_steps(0, 1))
if use:
    assert res["vq32"]["comm"] is not None and res["vq32"]["comm"]["comm_ms"] > 0, res["vq32"]["comm"]   # bucket collectives DID run on the side stream
    dist.barrier()
    dist.destroy_process_group()
torch.save(res, sys.argv[1])
"""


def _run(tmp, name, rccl):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SA_DDP_SINGLE_RANK", "SA_DETERMINISTIC")}
    env.update(SA_ROOT=ROOT, SA_DETERMINISTIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")    # fixed-order reductions: the two runs must agree bit for bit
    if rccl:
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", SA_DDP_SINGLE_RANK="1")
    out = os.path.join(tmp, name + ".pt")
    r = subprocess.run([sys.executable, "-c", _CHILD, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return torch.load(out, weights_only=False)


def test_product_collectives_run_on_rccl_and_are_the_identity_on_one_rank(tmp_path):
    plain = _run(str(tmp_path), "plain", False)
    rccl = _run(str(tmp_path), "rccl", True)
    assert plain["vq32"]["comm"] is None and rccl["vq32"]["buckets"] >= 3
    for key in ("vq32", "vq16"):
        a, b = plain[key], rccl[key]
        for k in ("params", "N", "embed_avg", "weight"):
            assert torch.equal(a[k], b[k]), (key, k)
        for s, (ga, gb) in enumerate(zip(a["grads"], b["grads"])):
            assert torch.equal(ga, gb), (key, "grad", s)
    a, b = plain["perf"], rccl["perf"]
    assert torch.equal(a["params"], b["params"])
    for s in range(len(a["grads"])):
        assert torch.equal(a["grads"][s], b["grads"][s]) and torch.equal(a["projs"][s], b["projs"][s]), s


def test_bench_line_under_a_one_rank_rccl_group():
    """`bench.py` launched the way the driver launches it (`python -m torch.distributed.run --nproc-per-node 1`, RANK / WORLD_SIZE / MASTER_* from the
    launcher): the rendezvous, RCCL's communicator, the barrier-bracketed timing and the MAX reduction of the N > 1 path, on the one GPU this box has."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SA_DDP_SINGLE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
                        str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1", "--no-performer",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 1 and d["value"] > 0 and np.isfinite(d["final_loss"])
    assert d.get("comm") and d["comm"]["comm_ms"] > 0 and d["comm"]["backend"] == "nccl", d.get("comm")
