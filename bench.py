#!/usr/bin/env python3
"""Headline benchmark: VQ-VAE training-step throughput (volumes/s, 160x224x160) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + MSE loss + backward + EMA codebook update + Adam over one batch of synthetic volumes already resident
in HBM (BASELINE.json configs[1]/[2]: baseline_vqvae no_levels=4 no_channels=256 embedding_dim=32 num_embeddings=2048,
bf16 MFMA, batch 8 per GPU, batch-sharded data parallel with RCCL all-reduce of gradients and EMA statistics).  Rank 0 prints
ONE JSON line.  The oracle / CPU restatement is touched only by the `cpu_baseline` leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC: RCCL between processes needs this before the runtime loads

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

VOL = (160, 224, 160)
NET = dict(n_levels=4, downsample_parameters=((4, 2, 1, 1),) * 4, upsample_parameters=((4, 2, 1, 0, 1),) * 4, n_embed=2048, embed_dim=32,
           n_channels=256, n_res_channels=256, n_res_layers=3, p_dropout=0.0, commitment_cost=0.25, vq_decay=0.5)
FWD_TFLOP_PER_VOLUME = 4.991      # 2 x 2495.5 GMAC
STEP_TFLOP_PER_VOLUME = 14.97     # fwd + dgrad + wgrad
PEAK_BF16_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3


def _host_description():
    model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    cfg = torch.__config__.show()
    libs = "; ".join(l.strip(" -") for l in cfg.splitlines() if ("oneAPI Math Kernel" in l or "oneDNN" in l or "MKL-DNN" in l or "OpenMP" in l))[:200]
    return f"{model}, {os.cpu_count()} logical CPUs; torch {torch.__version__} ({libs})"


def _physical_cores():
    """Physical cores this process may run on: psutil's count (distinct (package, core id) pairs of /proc/cpuinfo), bounded by the affinity mask."""
    n = None
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
    except Exception:
        n = None
    if not n:
        try:
            pairs, phys = set(), "0"
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("physical id"):
                        phys = line.split(":")[1].strip()
                    elif line.startswith("core id"):
                        pairs.add((phys, line.split(":")[1].strip()))
            n = len(pairs) or None
        except OSError:
            n = None
    if not n:
        n = max(1, (os.cpu_count() or 2) // 2)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return max(1, int(n))


def cpu_baseline():
    """Reference-equivalent CPU path (the oracle restatement, fp32 torch-CPU) on this host: ONE full 160x224x160 volume, forward + MSE + backward
    of the config-2 network -- always the full volume, no extrapolation -- timed once with every physical core and once with half of
    them (on the pool's 2 x 64-core hosts the 64-thread run has been the faster one: cross-socket traffic); `value` is the better of the two and
    both are listed.  A small crop step per thread count warms the thread pool and the allocator first.  About 50-60 s of CPU work."""
    from oracle import vqvae_ref

    phys = _physical_cores()               # all physical cores of the host (SMT siblings add nothing to MKL / oneDNN kernels)
    cfg = vqvae_ref.VQVAEConfig(**NET)
    st = vqvae_ref.init_state(cfg, seed=4)
    leaf = {k: v.requires_grad_(True) for k, v in st.items() if "quantizer" not in k}
    st.update(leaf)

    def one(shape):
        for p in leaf.values():
            p.grad = None
        x = torch.rand(1, 1, *shape, generator=torch.Generator().manual_seed(4))
        t0 = time.perf_counter()
        out = vqvae_ref.forward(st, cfg, x, training=True)
        vqvae_ref.mse_loss(out, x).backward()
        return time.perf_counter() - t0

    runs = []
    for threads in sorted({phys, max(1, phys // 2)}):
        torch.set_num_threads(threads)
        one((32, 48, 32))               # thread pool / allocator warm-up
        dt = one(VOL)
        runs.append({"threads": threads, "seconds": round(dt, 2), "volumes_per_sec": round(1.0 / dt, 5)})
    best = max(runs, key=lambda r: r["volumes_per_sec"])
    torch.set_num_threads(best["threads"])
    return {"value": best["volumes_per_sec"], "unit": "volumes/s", "cores": best["threads"], "kind": "port", "runs": runs,
            "sample": f"1 training step (fwd + MSE + bwd, fp32 torch-CPU oracle) of the config-2 network on ONE full {VOL[0]}x{VOL[1]}x{VOL[2]} volume per thread "
                      f"count ({', '.join(str(r['threads']) + ' threads: ' + str(r['seconds']) + ' s' for r in runs)}); value = the faster; host: {_host_description()}"}


def cpu_baseline_performer():
    """the Performer training step (N = 1 400, batch 1) on the oracle restatement, fp32 torch-CPU, same host / thread count."""
    import numpy as np

    from oracle import performer_ref as P
    threads = torch.get_num_threads()
    spatial = PERF["spatial"]
    n = int(np.prod(spatial))
    cfg = P.PerformerConfig(num_tokens=PERF["vocab"] + 1, max_seq_len=n + 1, dim=PERF["dim"], depth=PERF["depth"], heads=PERF["heads"], dim_head=64,
                            local_attn_heads=PERF["local_heads"], local_window_size=PERF["window"], spatial_shape=spatial, use_rezero=True)
    st = P.init_state(cfg, seed=4)
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "projection_matrix" not in k and v.dtype.is_floating_point}
    stt = dict(st)
    stt.update(leaf)
    seqs = P.spatial_index_sequences(spatial, np.arange(n))
    g = torch.Generator().manual_seed(4)
    tok = torch.randint(0, PERF["vocab"], (1, n), generator=g)
    tgt = torch.randint(0, PERF["vocab"], (1, n), generator=g)
    times = []
    for _ in range(2):
        for p in leaf.values():
            p.grad = None
        t0 = time.perf_counter()
        P.ce_loss(P.forward(stt, cfg, tok, seqs), tgt).backward()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    return {"value": n / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"training step (fwd + CE + bwd, fp32 torch-CPU oracle, parity-unpinned restatement) of the README Performer on 1 sequence of {n} tokens: best of 2 = {dt:.2f} s "
                      f"({sum(times):.1f} s total), {threads} threads"}


def measured_peaks(dev):
    """What this device delivers on the two rooflines at its own clocks: a pure-MFMA loop (v_mfma_f32_32x32x16_bf16, no memory) and a
    device-to-device copy of 2 GiB (read + write bytes)."""
    from synthanatomy_amd import _ffi
    lib = _ffi.lib()
    scratch = torch.zeros(4, device=dev)
    blocks, iters = 256 * 8, 4096
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _ffi.check(lib.sa_bench_mfma_bf16(_ffi.ptr(scratch), blocks, 64, _ffi.stream()), "sa_bench_mfma_bf16")
    torch.cuda.synchronize()
    e0.record()
    _ffi.check(lib.sa_bench_mfma_bf16(_ffi.ptr(scratch), blocks, iters, _ffi.stream()), "sa_bench_mfma_bf16")
    e1.record()
    torch.cuda.synchronize()
    tf = blocks * 4 * iters * 8 * 32768.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    # cross-check at other occupancies and with the instruction shape the convolution kernels use: waves per CU x shape
    cross = {}
    for shape, flop in ((0, 32768.0), (1, 16384.0)):
        for threads in (256, 512, 1024):
            blk, it2 = 256 * 8 * 256 // threads, 8192
            _ffi.check(lib.sa_bench_mfma_bf16_ex(_ffi.ptr(scratch), blk, threads, 64, shape, _ffi.stream()), "sa_bench_mfma_bf16_ex")
            torch.cuda.synchronize()
            e0.record()
            _ffi.check(lib.sa_bench_mfma_bf16_ex(_ffi.ptr(scratch), blk, threads, it2, shape, _ffi.stream()), "sa_bench_mfma_bf16_ex")
            e1.record()
            torch.cuda.synchronize()
            cross[f"{'32x32x16' if shape == 0 else '16x16x32'}_{threads // 64}waves_per_block"] = round(blk * (threads // 64) * it2 * 4 * flop / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    tf = max(tf, max(cross.values()))
    src = torch.empty(1 << 30, dtype=torch.int16, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    gbs = 4 * 2 * src.numel() * 2 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return {"mfma_bf16_tflops": round(tf, 1), "copy_gbs": round(gbs, 1), "mfma_cross_check_tflops": cross,
            "how": "sa_bench_mfma_bf16: 2048 blocks x 4 waves x 4096 x 8 independent v_mfma_f32_32x32x16_bf16; torch device-to-device copy of 2 GiB, read + write bytes"}


PERF = dict(vocab=2048, spatial=(10, 14, 10), dim=512, depth=24, heads=16, local_heads=8, window=420)
PERFORMER_STEP_MFLOP_PER_TOKEN = 618.7


def bench_performer(args, rank, world, dev, shape=None, batch=None):
    """Secondary metric: Performer training-step tokens/s (README.md:126-141 configuration, raster-ordered 10x14x10 latents; `shape` /
    `batch` override the latent grid and the sequences per GPU: 20x28x25 = BASELINE.json's "~14k-token" variant)."""
    import numpy as np

    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam

    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    spatial = tuple(shape) if shape is not None else tuple(int(v) for v in args.performer_shape.split(","))
    N = int(np.prod(spatial))
    B = batch if batch is not None else args.performer_batch
    torch.manual_seed(4)
    order = Ordering("raster_scan", 3, (1,) + spatial, (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
    net = Performer(num_tokens=PERF["vocab"] + 1, max_seq_len=N + 1, dim=PERF["dim"], depth=PERF["depth"], heads=PERF["heads"], ordering=order,
                    local_attn_heads=PERF["local_heads"], local_window_size=PERF["window"], feature_redraw_interval=1, use_rezero=True,
                    spatial_position_emb="absolute", spatial_shape=spatial, compute_dtype=dt).to(dev).train()
    flat = FlatParams(net.parameters())
    reducer = GradReducer(flat, mode=args.ddp_mode, transport=args.grad_transport)
    net.set_grad_sink(reducer)
    if not args.opt_in_backward:
        opt = FusedAdam(flat, lr=1e-3)
        opt.on_step.append(net.invalidate_packed_weights)
    else:   # the optimizer slice + operand re-pack of a bucket run behind its gradients, in the shadow of the backward pass (runtime/optim.FusedAdam)
        opt = FusedAdam(flat, lr=1e-3, in_backward=reducer)
        rp = net.range_repacker(flat)
        opt.on_range.append(rp)
        opt.on_step.append(rp.finish)
    loss_fn = CELoss()
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    codes = torch.randint(0, PERF["vocab"], (B, N), generator=gen, device=dev)
    seq = codes[:, torch.as_tensor(order.get_sequence_ordering(), device=dev)]
    seq = torch.nn.functional.pad(seq, (1, 0), value=PERF["vocab"])
    x_in, x_tgt = seq[:, :-1].contiguous(), seq[:, 1:].contiguous()

    def step():
        flat.zero_grad()
        logits = net(x_in)
        loss = loss_fn(logits.transpose(1, 2), x_tgt)
        loss.backward()
        opt.step(grad_scale=reducer.finish())
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    _spin_until(marks[-1])
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if dist.is_initialized():
        t = torch.tensor([dtm], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtm = float(t.item())
    toks = B * N * world * args.steps / dtm
    roof = None
    if rank == 0 and world == 1 and not args.no_kernel_timer and shape is None:
        # per-kernel pass for the roofline record: two more steps on ONE stream (the timed steps above overlap the weight gradients with the data-gradient
        # chain on a second stream, which inflates every per-kernel duration) with HIP events around every dense launch and the FAVOR+ launch groups
        from synthanatomy_amd import debug, engine
        with debug.override(no_side_wgrad=True):
            step()
            torch.cuda.synchronize()
            timer = engine.KernelTimer()
            engine.TIMER = timer
            for _ in range(2):
                step()
            stats = timer.collect()
            engine.TIMER = None
        single = {k: v for k, v in stats.items() if "+" not in k and v[1] > 0} or stats
        name, dom = max(single.items(), key=lambda kv: kv[1][2])
        n, flops, ms, ab = dom[0], dom[1], dom[2], dom[4]
        ach = flops / (ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if dt == torch.bfloat16 else PEAK_F32_TFLOPS
        traffic = _pmc_lookup(name, PMC_FILE_PERFORMER, PERFORMER_KERNEL_SOURCES) if (B == 6 and dt == torch.bfloat16) else None
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                # the dense launches sit below the MFMA / HBM ridge on their OWN bytes (bf16 operands in, fp32 residual stream + bf16 copies out): the HBM fraction of the
                # same launches, from their operand bytes (input, packed weights, every output / addend once), next to the MFMA fraction
                "algorithmic_bytes_per_launch": round(ab / n) if n else None,
                "achieved_gbs": round(ab / (ms * 1e-3) / 1e9, 1) if ms > 0 else None, "frac_hbm": round(ab / (ms * 1e-3) / 1e9 / 8000.0, 4) if ms > 0 else None,
                "traffic_over_algorithmic": round(traffic / (ab / n), 3) if (traffic and ab) else None,
                "traffic_source": _pmc_source(PMC_FILE_PERFORMER), "kernel": name,
                "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2),
                "note": "one-stream pass (weight gradients not overlapped) of 2 steps; dense layers (nn.Linear) on the im2col-order MFMA kernel; brackets with '+' "
                        "cover several launches (FAVOR+ groups: algorithmic FLOPs, the kernels execute ~3x on split-bf16 products)",
                "kernels": {k: {"launches": v[0], "ms": round(v[2], 3), "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[2] > 0 and v[1] > 0 else None}
                            for k, v in sorted(stats.items(), key=lambda kv: -kv[1][2])[:10]}}
    sampling = None
    if args.sampling and shape is None:
        # "also report sampling tokens/s (B10)": autoregressive sample() of full N-token sequences, stateful O(N) path
        net.eval()
        prefix = torch.full((B, 1), PERF["vocab"], dtype=torch.long, device=dev)
        with torch.no_grad():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out = net.sample(prefix, sample=True, top_k=None, temperature=1.0)
            torch.cuda.synchronize()
            dts = time.perf_counter() - t1
        if dist.is_initialized():
            t = torch.tensor([dts], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        sampling = {"metric": "performer_sampling_tokens_per_sec", "value": round(B * N * world / dts, 1), "unit": "tokens/s", "seconds": round(dts, 3),
                    "workload": f"sample() of {B} sequences x {N} tokens per GPU, stateful O(N) decoding (HIP graph per token), graph capture included"}
        assert tuple(out.shape) == (B, *spatial)
    res = {"metric": "performer_train_tokens_per_sec", "value": round(toks, 1), "unit": "tokens/s", "ms_per_step": round(dtm / args.steps * 1e3, 3),
           "dtype": args.dtype, "scaling": "weak", "final_loss": round(float(loss.item()), 5), "steps": args.steps, "warmup": args.warmup,
           "step_ms": {"median": round(step_ms[len(step_ms) // 2], 3), "min": round(step_ms[0], 3), "max": round(step_ms[-1], 3),
                       "tokens_per_sec_at_median": round(B * N * world / (step_ms[len(step_ms) // 2] * 1e-3), 1)},
           "tflops_per_gpu": round(toks / world * PERFORMER_STEP_MFLOP_PER_TOKEN / 1e6, 2),
           "config": {"workload": f"performer n_embd=512 n_layers=24 n_head=16 local_attn_heads=8 local_window=420 vocab=2048, N={N} raster-ordered "
                                  f"{'x'.join(map(str, spatial))} latents, training step = fwd + CE + bwd + Adam, projections redrawn every other step", "batch_per_gpu": B,
                      "global_batch": B * world, "seq_len": N, "parallelism": f"dp{world}"}}
    if roof is not None:
        res["roofline"] = roof
    if sampling is not None:
        res["sampling"] = sampling
    del net, flat, opt, reducer
    torch.cuda.empty_cache()
    return res


PMC_FILE = "r06_pmc_traffic.json"   # profiles/: HBM bytes per launch from the committed PMC passes (its `_provenance` names the command and commit)
PMC_FILE_PERFORMER = "r06_pmc_traffic_performer.json"


VQVAE_KERNEL_SOURCES = ("conv1.hip", "conv_fprop.hip", "conv_fprop_f16.hip", "conv_fprop_kernels.h", "conv_fprop_common.h", "conv_wgrad.hip", "convt1.hip",
                        "elementwise.hip", "norm.hip", "vq.hip", "sa_common.h", "split_bf16.h")
PERFORMER_KERNEL_SOURCES = ("performer.hip", "favor_fused.hip", "favor_proj.hip", "local_attn.hip", "local_attn_split.h", "conv_fprop.hip", "conv_fprop_kernels.h",
                            "conv_fprop_common.h", "conv_wgrad.hip", "elementwise.hip", "norm.hip", "sa_common.h", "split_bf16.h")


def _spin_until(event):
    """Poll the last step's event before the blocking synchronize that closes a timed region.  Seen once on a pool box (round 5): three of ten runs in one call
    reported a wall time ~2.8 s longer than the sum of their per-step HIP-event times, every step normal -- the host's interrupt-driven wait woke late, not the
    GPU.  hipEventQuery reads the completion signal directly; the synchronize behind it then finds nothing left to wait for on the launch stream."""
    while not event.query():
        pass


def csrc_digest(sources=VQVAE_KERNEL_SOURCES):
    """SHA-1 over the sources of the kernels a workload launches (synthanatomy_amd/csrc/, in the given order): `tools/rocpd_tools.py traffic` stores it next to
    the counters it summarises, and `roofline.traffic` is only reported while the sources are the ones that were profiled."""
    import hashlib
    d = os.path.join(ROOT, "synthanatomy_amd", "csrc")
    h = hashlib.sha1()
    for f in sources:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


_PMC_STATE = {}


def _pmc_lookup(kernel, fname, sources):
    """HBM bytes per launch of `kernel` from the PMC passes of the same command (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate rocprofv3 --pmc runs; counters
    cannot be read from inside the process), recorded in profiles/<fname> with its provenance and the digest of the kernel sources it was collected on;
    reported only while that digest matches the sources of this tree -- a stale file gives null, never an old number."""
    try:
        st = _PMC_STATE.setdefault(fname, {})
        if "rec" not in st:
            with open(os.path.join(ROOT, "profiles", fname)) as f:
                st["rec"] = json.load(f)
            st["fresh"] = st["rec"].get("csrc_sha1") == csrc_digest(sources)
        if not st["fresh"]:
            return None
        rec = st["rec"]["kernels"].get(kernel)
        return int(rec["hbm_bytes_per_launch"]) if rec else None
    except (OSError, ValueError, KeyError):
        _PMC_STATE.setdefault(fname, {}).setdefault("fresh", False)
        return None


def _pmc_traffic(kernel, args):
    """(VQ-VAE step: only for the configuration the counters were collected on -- default batch, bf16)"""
    if args.batch != 8 or args.dtype != "bf16":
        return None
    return _pmc_lookup(kernel, PMC_FILE, VQVAE_KERNEL_SOURCES)


def _pmc_source(fname=None):
    fname = fname or PMC_FILE
    if _PMC_STATE.get(fname, {}).get("fresh"):
        return f"profiles/{fname} (separate rocprofv3 --pmc passes of this command on these kernel sources; not sampled in this run)"
    return f"null: profiles/{fname} is missing or was collected on other kernel sources (csrc digest mismatch) -- re-run the PMC passes (tools/profile_round.sh)"


def _respawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, the same command line under
    torch.distributed.run) and pass their output through; rank 0 of the children prints the JSON line."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Launch plumbing without a GPU (CPU, gloo): rendezvous, the bucketed gradient reducer over the flat buffer of the REAL config-2 parameter list
    (28.1 M trainable parameters, 32 MiB buckets in backward order; `--ddp-mode` / `--grad-transport` select the collective), the packed
    [counts | dw] EMA statistics exchange, the file sharding of the CLIs, barrier-bracketed timing, MAX over ranks and the one JSON line --
    what `tests/test_bench_spawn_cpu.py` exercises with 2 and 8 ranks.  No kernel runs, so the line says dry_run."""
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.ddp import GradReducer, all_reduce_sum, init_distributed
    from synthanatomy_amd.runtime.optim import FlatParams
    from synthanatomy_amd.utils.general import shard_for_rank

    rank, local, world = init_distributed(backend="gloo")
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.manual_seed(4)
    net = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16)     # host parameters only: nothing is launched
    ps = [p for p in net.parameters() if p.requires_grad]
    flat = FlatParams(ps)
    red = GradReducer(flat, mode=args.ddp_mode, transport=args.grad_transport)

    def step():
        flat.zero_grad()
        for p in reversed(ps):
            red.buffer(p).add_(float(rank + 1))
            red.ready(p)
        return red.finish()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scale = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    want = world * (world + 1) / 2.0
    for p in ps:   # every parameter's gradient view carries the sum over ranks (and nothing else moved)
        v = red.buffer(p)
        assert float(v.min()) == want == float(v.max()), (tuple(p.shape), float(v.min()), float(v.max()), want)
    assert abs(scale - 1.0 / world) < 1e-12
    # EMA statistics: the packed [counts(K) | dw(K, D)] buffer summed over ranks (baseline.py:70-72), 270 KB
    K, D = NET["n_embed"], NET["embed_dim"]
    packed = torch.full((K + K * D,), float(rank + 1))
    if world > 1:
        all_reduce_sum(packed)
    assert float(packed[0]) == want and float(packed[-1]) == want
    # the CLIs' file sharding at this world size: every sample appears, every rank runs the same number of steps
    n_files = 8 * world + 3
    shards = [shard_for_rank(n_files, r, world, epoch=1, seed=4) for r in range(world)]
    assert len({len(sh) for sh in shards}) == 1 and set(i for sh in shards for i in sh) == set(range(n_files))
    if rank == 0:
        print(json.dumps({"metric": "vqvae_train_volumes_per_sec", "value": None, "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "dry_run": True,
                          "config": {"workload": "launch plumbing only (CPU, gloo): no kernel ran", "parallelism": f"dp{world}"},
                          "comm": {"buckets": len(red.buckets), "bytes_per_step": int(flat.numel * (2 if red.transport == "bf16" else 4)),
                                   "mode": red.mode, "transport": red.transport, "backend": dist.get_backend() if world > 1 else None}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_end_to_end():
    """BASELINE.json configs[4] on this GPU, production size, through the CLIs (tools/end_to_end.py): wall seconds per stage, file IO and network
    construction included -- a pipeline check with a clock on it, not a throughput figure."""
    import shutil
    import tempfile

    from tools import end_to_end
    import contextlib
    tmp = tempfile.mkdtemp(prefix="sa_e2e_")
    try:
        with contextlib.redirect_stdout(sys.stderr):      # the CLIs log to stdout; bench.py's stdout carries the ONE JSON line
            res = end_to_end.run_chain(tmp, volumes=4, extract=4, samples=2)
        return {"workload": res["workload"], "seconds": res["seconds"], "total_s": res["total_s"], "codes": len(res["codes"]),
                "samples": len(res["samples"]), "decoded_volumes": len(res["decoded"]), "bos_tokens_clamped": res["bos_tokens_clamped"]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        torch.cuda.empty_cache()


def bench_fp32_mode(dev, batch=2, steps=2):
    """The same training step with compute_dtype=float32 (exact-fp32 MFMA, 1/16 of the bf16 matrix rate): the mode that meets north_star's 1e-3
    tolerance (tests/test_width_parity_gpu.py).  Small batch / few steps: it is a reference point, not the headline."""
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    torch.manual_seed(4)
    net = BaselineVQVAE(**NET, compute_dtype=torch.float32).to(dev).train()
    flat = FlatParams(net.parameters())
    opt = FusedAdam(flat, lr=1.65e-4)
    opt.on_step.append(net.invalidate_packed_weights)
    loss_fn = MSELoss()
    x = torch.rand(batch, 1, *VOL, generator=torch.Generator(device=dev).manual_seed(4), device=dev)

    def step():
        flat.zero_grad()
        loss_fn(net(x), x).backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del net, flat, opt, x
    torch.cuda.empty_cache()
    return {"metric": "vqvae_train_volumes_per_sec", "value": round(batch / dt, 3), "unit": "volumes/s", "dtype": "f32", "batch_per_gpu": batch, "steps": steps,
            "ms_per_step": round(dt * 1e3, 2), "tflops_per_gpu": round(batch / dt * STEP_TFLOP_PER_VOLUME, 2), "peak_f32_tflops": PEAK_F32_TFLOPS}


def bench_bf16_vs_fp32(dev, batch=2, state=None, what="random-init weights"):
    """The benchmarked mode against the fp32 product path (the mode pinned to the reference at 1e-3) on the real config: same weights, `batch`
    160x224x160 volumes, eval.  Index agreement of the bf16 encoder + quantizer with the fp32 one, the evidence for every index that differs
    (utils/general.index_flip_report: how far from the fp32 optimum the 16-bit path landed, against the distance change ONE half-precision ulp of z
    would cause), and the bf16 decoder against the fp32 decoder on the SAME (fp32) indices.  `state`: a state_dict to compare on (the weights and the
    EMA codebook the timed training steps left) instead of a fresh initialisation."""
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.utils.general import index_flip_report
    torch.manual_seed(4)
    ref = BaselineVQVAE(**NET, compute_dtype=torch.float32)
    if state is not None:
        ref.load_state_dict(state)
    ref = ref.to(dev).eval()
    low = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16)
    low.load_state_dict(ref.state_dict())
    low = low.to(dev).eval()
    old = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16, encoder_forward_dtype=torch.bfloat16)   # bf16 forward operands too (the round-3 mode)
    old.load_state_dict(ref.state_dict())
    old = old.to(dev).eval()
    x = torch.rand(batch, 1, *VOL, generator=torch.Generator(device=dev).manual_seed(4), device=dev)
    with torch.no_grad():
        i32, i16, ibf = ref.index_quantize(x)[0], low.index_quantize(x)[0], old.index_quantize(x)[0]
        z32f, z16f = ref.encode(x)[0].float(), low.encode(x)[0].float()
        z32, z16, zbf = z32f.double(), z16f.double(), old.encode(x)[0].double()
        r32, r16 = ref.decode_samples([i32]).double(), low.decode_samples([i32]).double()
    rep = index_flip_report(z32f, z16f, ref.quantizer[0].impl.weight.detach(), i32, i16, max_list=32)
    res = {"index_agreement_vs_fp32": round(float((i32 == i16).float().mean()), 5), "positions": int(i32.numel()),
           "z_max_rel": float(f"{float((z16 - z32).abs().max() / z32.abs().max()):.3e}"),
           "recon_max_rel_same_indices": float(f"{float((r16 - r32).abs().max() / r32.abs().max()):.3e}"),
           "encoder_forward_dtype": str(low.encoder_forward_dtype).replace("torch.", ""),
           "flips": {"count": rep["flipped"], "max_ratio_gap_over_one_f16_ulp_of_z": float(f"{rep['max_flip_ratio']:.3g}"),
                     "median_ratio_unflipped": float(f"{rep['median_ratio_of_unflipped']:.3g}"), "p01_ratio_unflipped": float(f"{rep['p01_ratio_of_unflipped']:.3g}"),
                     "codebook_norm_max": float(f"{rep['codebook_norm_max']:.3g}"), "codebook_norm_median": float(f"{rep['codebook_norm_median']:.3g}"),
                     "each": [{k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in f.items()} for f in rep["flips"]],
                     "how": "fp32 path's distance from its optimum to the code the 16-bit path chose, divided by the distance change one IEEE-half ulp of every "
                            "channel of z would cause between the two codes (utils/general.index_flip_report); unflipped: the same ratio with the runner-up code"},
           "bf16_forward_operands": {"index_agreement_vs_fp32": round(float((i32 == ibf).float().mean()), 5),
                                     "z_max_rel": float(f"{float((zbf - z32).abs().max() / z32.abs().max()):.3e}")},
           "what": f"benchmarked mode (bf16 MFMA, encoder forward on float16 operands) vs fp32 product path, {what}, {batch} volumes "
                   f"{VOL[0]}x{VOL[1]}x{VOL[2]}, eval"}
    del ref, low, old, x
    torch.cuda.empty_cache()
    return res


def bench_latency_b1(dev, dtype, steps=5):
    """config 2 at B = 1 -- latency of one training step and of one extract + decode pass on ONE volume."""
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    torch.manual_seed(4)
    net = BaselineVQVAE(**NET, compute_dtype=dtype).to(dev).train()
    flat = FlatParams(net.parameters())
    opt = FusedAdam(flat, lr=1.65e-4)
    opt.on_step.append(net.invalidate_packed_weights)
    loss_fn = MSELoss()
    x = torch.rand(1, 1, *VOL, generator=torch.Generator(device=dev).manual_seed(4), device=dev)

    def step():
        flat.zero_grad()
        loss_fn(net(x), x).backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / steps
    net.eval()
    with torch.no_grad():
        for _ in range(2):
            net.decode_samples(net.index_quantize(x))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.decode_samples(net.index_quantize(x))
        torch.cuda.synchronize()
        t_inf = (time.perf_counter() - t0) / steps
    del net, flat, opt, x
    torch.cuda.empty_cache()
    return {"batch": 1, "train_step_ms": round(t_train * 1e3, 3), "train_volumes_per_sec": round(1.0 / t_train, 3),
            "train_tflops": round(STEP_TFLOP_PER_VOLUME / t_train, 1), "extract_decode_ms": round(t_inf * 1e3, 3),
            "extract_decode_volumes_per_sec": round(1.0 / t_inf, 3), "steps": steps}


DISC_FWD_TFLOP_PER_VOLUME = 0.3108   # 155.4 GMAC forward per 160x224x160 volume


def bench_adversarial(dev, dtype, batch, steps=5, warmup=3):
    """The README's training command (reference README.md:62-67): --adversarial_component=True with baseline_discriminator (ndf 64), least-square
    criteria weight 0.005, adaptive weight off -- one G + D iteration (src/engines/trainer.py:157-256) per step.  Added work per volume: the
    discriminator runs forward on the fakes and back to the reconstruction in the G step (2 x fwd), forward + full backward on fakes and reals in
    the D step (2 x 3 x fwd): 8 x 0.311 = 2.49 TFLOP on top of the generator's 14.97.  Per-iteration HIP events and the caching allocator's
    counters ride along: a regression of this leg (round 5: 168 -> 1 431 ms, unnoticed) shows as iterations of unequal length or as
    device allocations / retries inside the timed iterations."""
    from synthanatomy_amd.engines.trainer import AdversarialTrainer
    from synthanatomy_amd.losses.adversarial import get_discriminator_loss, get_generator_loss
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.discriminator.baseline import BaselineDiscriminator
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    torch.manual_seed(4)
    net = BaselineVQVAE(**NET, compute_dtype=dtype).to(dev).train()
    disc = BaselineDiscriminator(input_nc=1, ndf=64, n_layers=3, compute_dtype=dtype).to(dev).train()
    flat, d_flat = FlatParams(net.parameters()), FlatParams(disc.parameters())
    opt, d_opt = FusedAdam(flat, lr=1.65e-4), FusedAdam(d_flat, lr=5e-5)
    opt.on_step.append(net.invalidate_packed_weights)
    d_opt.on_step.append(lambda: [st.op.invalidate() for st in disc._stages])
    tr = AdversarialTrainer(net, opt, get_generator_loss({"generator_loss": "least_square"}), MSELoss(), disc, d_opt,
                            get_discriminator_loss({"discriminator_loss": "least_square"}))
    x = torch.rand(batch, 1, *VOL, generator=torch.Generator(device=dev).manual_seed(4), device=dev)
    for _ in range(warmup):
        res = tr.iteration(x, x, 1)
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats(dev)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        res = tr.iteration(x, x, 1)
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms1 = torch.cuda.memory_stats(dev)
    it_ms = [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(steps)]
    tf = STEP_TFLOP_PER_VOLUME + 8 * DISC_FWD_TFLOP_PER_VOLUME
    out = {"metric": "vqvae_adversarial_train_volumes_per_sec", "value": round(batch / dt, 3), "unit": "volumes/s", "ms_per_step": round(dt * 1e3, 2),
           "batch_per_gpu": batch, "steps": steps, "warmup": warmup, "iteration_ms": it_ms,
           "allocator": {"device_allocs_in_timed_region": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                         "device_frees_in_timed_region": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
                         "alloc_retries_in_timed_region": int(ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0)),
                         "reserved_gb": round(ms1.get("reserved_bytes.all.current", 0) / 2 ** 30, 2),
                         "peak_allocated_gb": round(ms1.get("allocated_bytes.all.peak", 0) / 2 ** 30, 2)},
           "tflop_per_volume": round(tf, 2), "tflops_per_gpu": round(batch / dt * tf, 1),
           "losses": {k: round(float(res[k]), 6) for k in ("loss", "g_loss", "d_loss")},
           "workload": "G + D iteration: baseline_vqvae config 2 + baseline_discriminator(1, 64, 3), MSE + 0.005 x least-square GAN terms, two Adam steps"}
    del net, disc, tr, flat, d_flat, opt, d_opt, x
    torch.cuda.empty_cache()
    return out


# ---- regression guard (round 6): every headline number of the line against the newest committed final line of an EARLIER tree.  The round-5 line carried an
# "adversarial" record 8.5 x below round 4's through four committed lines because nothing compared the sub-records.
GUARDED = (("value", True), ("inference.value", True), ("adversarial.value", True), ("latency_b1.train_volumes_per_sec", True),
           ("latency_b1.extract_decode_volumes_per_sec", True), ("fp32_mode.value", True), ("secondary.value", True), ("secondary.sampling.value", True),
           ("secondary_14k.value", True), ("end_to_end.total_s", False), ("roofline.frac", True), ("roofline_hbm.frac", True), ("secondary.roofline.frac", True))
# (per-record thresholds: the configs[4] chain is five process start-ups around a few seconds of work -- its wall time moves by 15 % between boxes)
GUARD_THRESHOLD = {"end_to_end.total_s": 0.30}


def _dig(d, path):
    for k in path.split("."):
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d if isinstance(d, (int, float)) and not isinstance(d, bool) else None


def regression_report(line, threshold=0.10, ref_path=None):
    """{"against": file, "regressions": [...], "compared": {...}}: sub-records more than ``threshold`` worse than the reference line (the newest
    profiles/rNN_final_bench_line.json, or SA_BENCH_REFERENCE_LINE)."""
    import glob
    import re
    ref_path = ref_path or os.environ.get("SA_BENCH_REFERENCE_LINE")
    if ref_path is None:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_bench_line.json")),
                       key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)))
        if not cands:
            return None
        ref_path = cands[-1]
    try:
        with open(ref_path) as f:
            ref = json.loads(f.readline())
    except (OSError, ValueError) as e:
        return {"against": os.path.relpath(ref_path, ROOT), "error": str(e)}
    compared, regs = {}, []
    for path, higher in GUARDED:
        new, old = _dig(line, path), _dig(ref, path)
        if new is None or old is None or old == 0:
            continue
        ratio = new / old
        compared[path] = round(ratio, 3)
        thr = max(threshold, GUARD_THRESHOLD.get(path, 0.0))
        if (ratio < 1 - thr) if higher else (ratio > 1 + thr):
            regs.append({"record": path, "now": new, "reference": old, "ratio": round(ratio, 3)})
    return {"against": os.path.relpath(ref_path, ROOT), "threshold": threshold, "regressions": regs, "compared": compared}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (SURVEY 8(d): >= 50)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps before them (SURVEY 8(d): 20)")
    ap.add_argument("--roofline-steps", type=int, default=3, help="steps of the separate per-kernel pass behind the timed region")
    ap.add_argument("--batch", type=int, default=8, help="volumes per GPU per step (README.md:72: batch 8/GPU)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the measured-peak probes and the fp32-mode sub-record")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the configs[4] chain through the CLIs (tools/end_to_end.py)")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--performer-batch", type=int, default=6, help="sequences per GPU per step (README.md:119)")
    ap.add_argument("--no-performer", action="store_true", help="skip the secondary Performer tokens/s measurement")
    ap.add_argument("--no-sampling", dest="sampling", action="store_false", help="skip the autoregressive sampling tokens/s measurement")
    ap.add_argument("--only-performer", action="store_true", help="dev/profiling: measure only the Performer workload")
    ap.add_argument("--only-adversarial", action="store_true", help="dev/profiling: measure only the adversarial G + D iteration sub-record")
    ap.add_argument("--performer-shape", default="10,14,10", help="latent grid of the Performer workload (20,28,25 = the 14 000-token variant)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo: exercise the launch + reduction plumbing only (no GPU, no kernels)")
    ap.add_argument("--ddp-mode", default=None, choices=["all_reduce", "reduce_scatter"],
                    help="gradient collective per bucket: one all-reduce (default, what DDP issues) or reduce-scatter + all-gather (runtime/ddp.GradReducer)")
    ap.add_argument("--grad-transport", default=None, choices=["fp32", "bf16"], help="gradient bytes on the links (default fp32)")
    ap.add_argument("--opt-in-backward", action="store_true", help="A/B: per-bucket Adam slices + re-packs behind the gradients (FusedAdam in_backward) instead of one launch each after backward")
    ap.add_argument("--share-device", action="store_true",
                    help="test aid: all ranks on cuda:0 with the gloo backend (RCCL refuses two ranks per device) -- the N > 1 code path on real kernels, not a number")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # started without a launcher: spawn one process per GPU ourselves
        sys.exit(_respawn(args))
    if args.dry_run:
        return dry_run(args)

    from synthanatomy_amd import engine
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.ddp import GradReducer, init_distributed
    from synthanatomy_amd.runtime.optim import ExponentialLR, FlatParams, FusedAdam

    if args.share_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, local, world = init_distributed(backend="gloo" if args.share_device else None)
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", local)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    if args.only_adversarial:
        print(json.dumps(bench_adversarial(dev, dtype, args.batch)), flush=True)
        return
    if args.only_performer:
        res = bench_performer(args, rank, world, dev)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    torch.manual_seed(4)
    net = BaselineVQVAE(**NET, compute_dtype=dtype).to(dev).train()
    flat = FlatParams(net.parameters())
    reducer = GradReducer(flat, mode=args.ddp_mode, transport=args.grad_transport)
    net.set_grad_sink(reducer)
    if not args.opt_in_backward:
        opt = FusedAdam(flat, lr=1.65e-4)
        opt.on_step.append(net.invalidate_packed_weights)
    else:
        opt = FusedAdam(flat, lr=1.65e-4, in_backward=reducer)
        rp = net.range_repacker(flat)
        opt.on_range.append(rp)
        opt.on_step.append(rp.finish)
    sched = ExponentialLR(opt, gamma=0.99999)
    loss_fn = MSELoss()
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    x = torch.rand(args.batch, 1, *VOL, generator=gen, device=dev, dtype=torch.float32)

    def step():
        flat.zero_grad()
        out = net(x)
        loss = loss_fn(out, x)
        loss.backward()
        scale = reducer.finish()
        opt.step(grad_scale=scale)
        sched.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    # the timed region: EXACTLY args.steps steps, nothing else on the streams (no per-launch events: the roofline pass below is separate), a HIP event on the
    # launch stream at every step boundary (hipEventElapsedTime per step, median reported next to the wall-clock mean that `value` uses)
    reducer.timing = dist.is_initialized()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    _spin_until(marks[-1])
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    reducer.timing = False
    comm = reducer.comm_stats()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    # roofline pass, AFTER the timed region: a few more steps with HIP events around every launch on the launch stream (engine.KernelTimer)
    timer = None
    if not args.no_kernel_timer:          # (every rank runs the steps -- they contain collectives --, rank 0 carries the events)
        if rank == 0:
            timer = engine.KernelTimer()
            engine.TIMER = timer
        from synthanatomy_amd import debug as _dbg
        with _dbg.override(no_side_wgrad=True):      # one stream: overlapped weight gradients would inflate every per-kernel duration
            for _ in range(args.roofline_steps):
                step()
            torch.cuda.synchronize()
        if timer is not None:
            stats = timer.collect()
        engine.TIMER = None
    if dist.is_initialized():
        dist.barrier()

    roof = None
    if timer is not None:
        # brackets that cover two kernels (wgrad + its partial-tile reduce) stay in `kernels` but are not the roofline kernel:
        # its avg_launch_us has to be comparable with rocprofv3's per-kernel average
        single = {k: v for k, v in stats.items() if "+" not in k} or stats
        dom = max(single.items(), key=lambda kv: kv[1][2])
        name, (n, flops, ms, _nb, ab) = dom
        ach = flops / (ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": _pmc_traffic(name, args), "traffic_source": _pmc_source(),
                "algorithmic_bytes_per_launch": round(ab / n) if n else None,      # operand bytes of the same launches (input, weights, every output / addend / mask once)
                "kernel": name, "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2),
                "how": f"separate pass of {args.roofline_steps} steps after the timed region, HIP events around every launch on the launch stream",
                "kernels": {k: {"launches": v[0], "ms": round(v[2], 3), "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[2] > 0 else None}
                            for k, v in sorted(stats.items(), key=lambda kv: -kv[1][2])}}

    roof_hbm = None
    if timer is not None:
        # the largest HBM-bound launch of the step (bandwidth-bound sub-kernels are reported against HBM): the fused
        # 1x1x1 backward of the residual blocks.  Its bracket also covers the small partial-tile reduce (< 3 % of the time).
        hb = {k: v for k, v in stats.items() if v[3] > 0}
        if hb:
            k, v = max(hb.items(), key=lambda kv: kv[1][2])
            gbs = v[3] / (v[2] * 1e-3) / 1e9
            roof_hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                        "traffic": _pmc_traffic(k.split("+")[0], args), "kernel": k, "launches": v[0], "avg_launch_us": round(v[2] * 1e3 / v[0], 2),
                        # every launch the step reports against bandwidth (the one-channel first / last layers, the fused 1x1x1 backward)
                        "kernels": {kk: {"launches": vv[0], "avg_launch_us": round(vv[2] * 1e3 / vv[0], 2), "gbs": round(vv[3] / (vv[2] * 1e-3) / 1e9, 1),
                                         "frac": round(vv[3] / (vv[2] * 1e-3) / 1e9 / 8000.0, 4)} for kk, vv in sorted(hb.items(), key=lambda kv: -kv[1][2])}}
    if rank == 0:
        vols = args.batch * world * args.steps
        value = vols / dt
        line = {
            "metric": "vqvae_train_volumes_per_sec", "value": round(value, 4), "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "step_ms": {"median": round(step_ms[len(step_ms) // 2], 3), "min": round(step_ms[0], 3), "max": round(step_ms[-1], 3),
                        "p10": round(step_ms[len(step_ms) // 10], 3), "p90": round(step_ms[(len(step_ms) * 9) // 10 if len(step_ms) > 1 else 0], 3),
                        "volumes_per_sec_at_median": round(args.batch * world / (step_ms[len(step_ms) // 2] * 1e-3), 3),
                        "wall_minus_events_ms": round(dt * 1e3 - sum(step_ms), 3),     # host-side time the step events do not see (a late wake-up shows here)
                        "how": "hipEventElapsedTime between events recorded on the launch stream at every step boundary of the timed region (rank 0); "
                               "`value` = volumes / wall time of the whole region (barrier + synchronize on both sides, MAX over ranks)"},
            "config": {"workload": "baseline_vqvae no_levels=4 no_channels=256 embedding_dim=32 num_embeddings=2048, 160x224x160 fp32 volumes, "
                                   "training step = fwd + MSE + bwd + EMA codebook update + Adam", "batch_per_gpu": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            **({"share_device": True} if args.share_device else {}),
            "tflops_per_gpu": round(value / world * STEP_TFLOP_PER_VOLUME, 2), "final_loss": round(final_loss, 6),
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "reserved_mem_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 2),     # the caching allocator's pool (round 5: 4 x the live bytes under record_stream)
        }
        if comm is not None:
            comm["backend"] = dist.get_backend()
            line["comm"] = comm   # gradient all-reduce on the side stream, rank 0: total / exposed after backward / hidden under backward
        if roof is not None:
            line["roofline"] = roof
        if roof_hbm is not None:
            line["roofline_hbm"] = roof_hbm
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            if not args.no_performer:
                line["cpu_baseline_performer"] = cpu_baseline_performer()
    # "report also inference (index_quantize + decode_samples) volumes/s"
    net.eval()
    with torch.no_grad():
        for _ in range(max(1, min(3, args.warmup))):
            rec = net.decode_samples(net.index_quantize(x))
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rec = net.decode_samples(net.index_quantize(x))
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        dti = time.perf_counter() - t1
    if dist.is_initialized():
        t = torch.tensor([dti], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dti = float(t.item())
    if rank == 0:
        line["inference"] = {"metric": "vqvae_extract_decode_volumes_per_sec", "value": round(args.batch * world * args.steps / dti, 3), "unit": "volumes/s",
                             "ms_per_step": round(dti / args.steps * 1e3, 3), "workload": "index_quantize + decode_samples (eval), same volumes"}
    del rec
    if rank == 0 and world == 1 and not args.no_extras:
        if roof is not None:
            pk = measured_peaks(dev)
            line["roofline"]["peak_measured"] = pk
            line["roofline"]["frac_of_measured"] = round(roof["achieved"] / pk["mfma_bf16_tflops"], 4) if dtype == torch.bfloat16 else None
            if "roofline_hbm" in line:
                line["roofline_hbm"]["peak_measured_gbs"] = pk["copy_gbs"]
        if dtype == torch.bfloat16:
            trained = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}      # weights + EMA codebook after the warm-up, timed and roofline steps
            n_trained = args.warmup + args.steps + (0 if args.no_kernel_timer else args.roofline_steps)
            line["fp32_mode"] = bench_fp32_mode(dev)
            line["fp32_mode"]["bf16_vs_fp32"] = bench_bf16_vs_fp32(dev)
            line["fp32_mode"]["bf16_vs_fp32_trained"] = bench_bf16_vs_fp32(
                dev, state=trained, what=f"the weights and EMA codebook after the {n_trained} training steps of this run (synthetic volumes)")
            del trained
    secondary = secondary_14k = None
    del net, flat, opt, reducer, x
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_extras:
        line["latency_b1"] = bench_latency_b1(dev, dtype)
        line["adversarial"] = bench_adversarial(dev, dtype, args.batch)
    if not args.no_performer:
        secondary = bench_performer(args, rank, world, dev)
        torch.cuda.empty_cache()
        if args.performer_shape == "10,14,10":   # BASELINE.json configs[3] says "~14k-token" latents: also the 20x28x25 grid, one sequence per GPU
            secondary_14k = bench_performer(args, rank, world, dev, shape=(20, 28, 25), batch=1)
    if rank == 0 and world == 1 and not args.no_extras and not args.no_end_to_end:
        line["end_to_end"] = bench_end_to_end()
    if rank == 0:
        if secondary is not None:
            line["secondary"] = secondary
        if secondary_14k is not None:
            line["secondary_14k"] = secondary_14k
        if world == 1:
            rep = regression_report(line)
            if rep is not None:
                line["regression_guard"] = rep
                for r in rep.get("regressions", []):
                    print(f"[bench] REGRESSION {r['record']}: {r['now']} vs {r['reference']} in {rep['against']} (x{r['ratio']})", file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
