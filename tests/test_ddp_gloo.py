"""CPU, world_size 2 (gloo): the N>1 path -- bucketed gradient all-reduce overlapping 'backward', and the EMA statistics
exchange (sum over ranks, baseline.py:70-72) giving every rank the codebook a single rank would compute on the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from synthanatomy_amd.runtime.ddp import GradReducer, init_distributed
    from synthanatomy_amd.runtime.optim import FlatParams
    r, l, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in [(64, 3), (17,), (8, 8), (5,)]]
    flat = FlatParams(ps)
    red = GradReducer(flat, bucket_bytes=300)
    assert len(red.buckets) >= 2
    # "backward": last parameter first, each rank contributes rank+1
    for p in reversed(ps):
        red.buffer(p).add_(float(rank + 1))
        red.ready(p)
    scale = red.finish()
    ok_grad = bool(torch.allclose(flat.grad[: ps[0].numel()], torch.full((ps[0].numel(),), 3.0))) and scale == 0.5

    # the other collectives behind the same sink (reduce-scatter + all-gather, bf16 transport) against the all-reduce path:
    # seeded, rank-dependent gradients; fp32 transport must agree to rounding (a different summation tree), bf16 to bf16 rounding
    def reduced(mode, transport):
        f = FlatParams([torch.nn.Parameter(p.detach().clone()) for p in ps])
        rd = GradReducer(f, bucket_bytes=300, mode=mode, transport=transport)
        gg = torch.Generator().manual_seed(100 + rank)
        for p in reversed(f.params):
            rd.buffer(p).copy_(torch.randn(p.shape, generator=gg))
            rd.ready(p)
        assert rd.finish() == 0.5
        return f.grad.clone(), rd
    base, rd0 = reduced("all_reduce", "fp32")
    rs, rd1 = reduced("reduce_scatter", "fp32")
    lp, _ = reduced("all_reduce", "bf16")
    rslp, _ = reduced("reduce_scatter", "bf16")
    assert rd0.buckets == rd1.buckets and len(rd1.buckets) >= 2   # (views are 16-byte aligned, so only world sizes that do not divide 4 see a tail: the 8-rank dry run)
    ok_grad = ok_grad and bool(torch.allclose(rs, base, rtol=1e-6, atol=1e-6)) and bool(torch.allclose(lp, base, rtol=2e-2, atol=2e-2)) \
        and bool(torch.allclose(rslp, lp, rtol=2e-2, atol=2e-2)) and not torch.equal(lp, base)
    both = [torch.empty_like(base) for _ in range(world)]
    dist.all_gather(both, rs)
    ok_grad = ok_grad and all(torch.equal(b, both[0]) for b in both)      # replicas hold identical sums after the all-gather

    # EMA statistics: each rank quantizes its half of the batch; the packed [counts | dw] buffer is summed
    from oracle import vqvae_ref
    cfg = vqvae_ref.VQVAEConfig(n_embed=32, embed_dim=8, vq_decay=0.5)
    g = torch.Generator().manual_seed(1)
    W0 = torch.randn(32, 8, generator=g)
    z = torch.randn(4, 8, 3, 3, 3, generator=g)
    st = {"quantizer.0.impl.weight": W0.clone(), "quantizer.0.impl.N": torch.zeros(32), "quantizer.0.impl.embed_avg": W0.clone()}
    _, _, _, aux = vqvae_ref.quantize({k: v.clone() for k, v in st.items()}, cfg, z[2 * rank: 2 * rank + 2], training=False)
    packed = torch.cat([aux["counts"], aux["dw"].reshape(-1)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    vqvae_ref.quantize(st, cfg, z[2 * rank: 2 * rank + 2], training=True, world_stats=(packed[:32], packed[32:].view(32, 8)))
    full = {"quantizer.0.impl.weight": W0.clone(), "quantizer.0.impl.N": torch.zeros(32), "quantizer.0.impl.embed_avg": W0.clone()}
    vqvae_ref.quantize(full, cfg, z, training=True)
    ok_ema = all(torch.allclose(st[k], full[k], rtol=1e-5, atol=1e-6) for k in st)
    q.put((rank, ok_grad, ok_ema))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reducer_and_ema_statistics():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)], res


def _dying_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from synthanatomy_amd.runtime.ddp import GradReducer, init_distributed
    from synthanatomy_amd.runtime.optim import FlatParams
    init_distributed(backend="gloo", timeout_s=6.0)
    ps = [torch.nn.Parameter(torch.randn(32, 4)), torch.nn.Parameter(torch.randn(9))]
    flat = FlatParams(ps)
    red = GradReducer(flat, bucket_bytes=1 << 20)

    def step():
        for p in reversed(ps):
            red.buffer(p).fill_(1.0)
            red.ready(p)
        return red.finish()
    step()                       # one healthy step: both ranks take part
    if rank == 1:
        os._exit(17)             # the rank dies without leaving the group (a killed process / a lost node)
    q.put((rank, "step1", 0.0))
    t0 = time.time()
    try:
        step()
        q.put((rank, "completed", time.time() - t0))     # must not happen: the peer is gone
    except Exception as exc:      # gloo: the waiting call raises (connection closed by peer, or the collective's timeout)
        q.put((rank, "raised:" + type(exc).__name__, time.time() - t0))
    q.close()
    q.join_thread()              # (the feeder thread has written the messages before the hard exit below, which skips gloo's destructors)
    os._exit(0)


def test_dead_rank_fails_the_others_within_the_timeout():
    """runtime/ddp.init_distributed(timeout_s=...): when a rank dies, the survivors' next collective FAILS within the timeout instead of hanging for torch's
    default 10 / 30 minutes (the reference leaves deepspeed's default, run_vqvae.py:831-842)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dying_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert procs[1].exitcode == 17
    last = [g for g in got if g[0] == 0 and g[1] != "step1"]
    assert len(last) == 1 and last[0][1].startswith("raised:"), got
    assert last[0][2] < 30.0, got      # timeout 6 s (+ slack for a loaded CI box), far below the default


def test_optimizer_slices_wait_for_finish_unless_the_report_followed_flush_then_ready():
    """GradReducer.on_bucket (optimizer in backward) overwrites parameters behind a bucket's gradients.  Only a bucket whose every parameter was reported with the
    token of the flush() that opened its layer's report may step before finish(); a plain ready() (autograd hooks, third-party sink users) keeps the bucket's
    slice until finish() (nothing enforced the contract)."""
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import FlatParams
    ps = [torch.nn.Parameter(torch.randn(40)) for _ in range(4)]
    flat = FlatParams(ps)
    red = GradReducer(flat, bucket_bytes=200)      # 40 floats = 160 bytes per parameter: one bucket each
    assert len(red.buckets) == 4
    calls = []
    red.on_bucket = lambda lo, hi, scale: calls.append((lo, hi))
    tok = red.flush()
    red.ready(ps[3], tok)            # disciplined: its slice goes at the NEXT flush
    assert calls == []
    red.ready(ps[2])                 # no token: waits for finish()
    tok = red.flush()
    assert calls == [(flat.offsets[3], flat.numel)]
    red.ready(ps[1], tok - 1)        # a stale token is no proof either
    tok = red.flush()
    assert len(calls) == 1
    red.ready(ps[0], tok)
    red.finish()
    assert sorted(calls) == sorted((flat.offsets[i], flat.offsets[i + 1] if i < 3 else flat.numel) for i in range(4))
    assert calls[0] == (flat.offsets[3], flat.numel) and calls[1][0] == flat.offsets[0]   # disciplined ones first, the tainted ones at the end
