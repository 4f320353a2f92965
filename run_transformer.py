#!/usr/bin/env python3
"""Transformer entry point with the reference's flags and modes (reference run_transformer.py:405-635):

    python run_transformer.py run --training_subjects=<dir of *_quantization_0.npy | synthetic:64> --validation_subjects=... \\
        --project_directory=/tmp/proj/ --experiment_name=exp --mode=training|inference --vocab_size=2048 --n_embd=512 ...

Training: codes -> ordering -> BOS pad -> Performer -> cross entropy, Adam, per-iteration ExponentialLR.  Inference: sample
``prod(spatial_shape)`` tokens autoregressively, revert the ordering and write uint16 ``.npy`` code grids (postfix "sample")
that ``run_vqvae.py --mode=decoding`` consumes.  ``--n_embed`` is accepted as an alias of ``--n_embd`` (README vs. code).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synthanatomy_amd.utils.general import (REQUIRED, check_for_checkpoints, create_folder_structure, list_inputs, load_checkpoint,  # noqa: E402
                                            load_network_state, log, parse_flags, save_checkpoint, save_npy, shard_for_rank)

DEFAULTS = dict(
    training_subjects=REQUIRED, validation_subjects=REQUIRED, project_directory=REQUIRED, experiment_name=REQUIRED, mode="training",
    conditioning_path=None, conditionings=None, conditioning_type="bos_replacement", device=0, deterministic=False, cuda_benchmark=True, seed=2,
    epochs=1000000, learning_rate=1e-4, gamma="auto", log_every=25, checkpoint_every=50, eval_every=50, sample=True, temperature=1.0, top_k=None,
    batch_size=2, eval_batch_size=2, num_workers=8, prefetch_factor=6, starting_epoch=0, ordering_type="raster_scan",
    reflected_spatial_dims=(False, False, False), transpositions_axes=tuple(), rot90_axes=tuple(),
    transformation_order=("transpose", "rotate_90", "reflect"), network="performer", vocab_size=32, n_embd=256, n_layers=10, n_head=8,
    local_attn_heads=0, local_window_size=256, feature_redraw_interval=1000, generalized_attention=False, emb_dropout=0.0, ff_dropout=0.0,
    attn_dropout=0.0, use_rezero=False, position_emb="absolute", spatial_position_emb=None, evaluation_checkpoint="recent",
    # MI355X-only: latent grid for synthetic inputs, GEMM dtype
    spatial_shape=(10, 14, 10), compute_dtype="fp32", training_epoch_length=None,
)


def _load_codes(path, cfg, gen):
    if path.startswith("synthetic"):   # codes that depend on the NAME only: resumable, rank-independent
        g = torch.Generator().manual_seed(cfg["seed"] * 1000003 + int(path.split("_")[-1]))
        return torch.randint(0, cfg["vocab_size"], tuple(cfg["spatial_shape"]), generator=g)
    return torch.from_numpy(np.load(path).astype(np.int64))


def build(cfg, dims, dev):
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    if cfg["network"] != "performer":
        raise ValueError(f"Transformer unknown. Was given {cfg['network']} but choices are ['performer'].")
    if cfg["position_emb"] not in ("absolute", "fixed"):      # run_transformer.py:86-88 maps "fixed" to fixed_position_emb, everything else to the learned table
        raise NotImplementedError("position_emb other than 'absolute' / 'fixed'")
    ordering = Ordering(ordering_type=cfg["ordering_type"], spatial_dims=len(dims), dimensions=(1,) + tuple(dims),
                        reflected_spatial_dims=cfg["reflected_spatial_dims"], transpositions_axes=cfg["transpositions_axes"],
                        rot90_axes=cfg["rot90_axes"], transformation_order=cfg["transformation_order"])
    net = Performer(num_tokens=cfg["vocab_size"] + 1, max_seq_len=int(np.prod(dims)) + 1, dim=cfg["n_embd"], depth=cfg["n_layers"], heads=cfg["n_head"],
                    ordering=ordering, local_attn_heads=cfg["local_attn_heads"], local_window_size=cfg["local_window_size"],
                    feature_redraw_interval=cfg["feature_redraw_interval"], generalized_attention=cfg["generalized_attention"],
                    emb_dropout=cfg["emb_dropout"], ff_dropout=cfg["ff_dropout"], attn_dropout=cfg["attn_dropout"], use_rezero=cfg["use_rezero"],
                    fixed_position_emb=cfg["position_emb"] == "fixed", spatial_position_emb=cfg["spatial_position_emb"], spatial_shape=tuple(dims),
                    compute_dtype=torch.bfloat16 if cfg["compute_dtype"] == "bf16" else torch.float32)
    return net.to(dev), ordering


def _validation_ce(net, ordering, files, cfg, gen, dev, rank, world):
    """Mean cross entropy over the validation subjects -- the key metric of run_transformer.py:136-143 (the best-checkpoint rule keeps the
    HIGHEST score, so the score is the negated loss)."""
    import torch.distributed as dist

    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.utils.transformer import prepare_batch
    was = net.training
    net.eval()
    tot = torch.zeros(2, device=dev, dtype=torch.float64)
    loss_fn = CELoss()
    with torch.no_grad():
        order = shard_for_rank(len(files), rank, world, shuffle=False, pad=False)
        for i in range(0, len(order), cfg["eval_batch_size"]):
            q = torch.stack([_load_codes(files[k], cfg, gen) for k in order[i:i + cfg["eval_batch_size"]]])
            (x_in, _), x_tgt = prepare_batch({"quantization": q}, ordering.get_sequence_ordering(), cfg["vocab_size"], device=dev)
            tot[0] += loss_fn(net(x_in).transpose(1, 2), x_tgt).double() * q.shape[0]
            tot[1] += q.shape[0]
    if world > 1:
        dist.all_reduce(tot)
    net.train(was)
    return float(tot[0] / tot[1].clamp(min=1))


def training(cfg, rank, local, world, dev):
    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import ExponentialLR, FlatParams, FusedAdam, TrainerState
    from synthanatomy_amd.utils.transformer import prepare_batch
    gen = torch.Generator().manual_seed(cfg["seed"] + rank)
    files = list_inputs(cfg["training_subjects"], postfix="quantization_0")
    val_files = list_inputs(cfg["validation_subjects"], postfix="quantization_0")
    dims = tuple(_load_codes(files[0], cfg, gen).shape)  # the reference peeks one batch for the latent shape (run_transformer.py:54-56)
    net, ordering = build(cfg, dims, dev)
    net.train()
    flat = FlatParams(net.parameters())
    red = GradReducer(flat)
    net.set_grad_sink(red)
    from synthanatomy_amd import debug
    if debug.host("opt_in_backward"):     # SA_OPT_IN_BACKWARD=1: a bucket's Adam slice + operand re-pack run behind its gradients (runtime/optim.py)
        opt = FusedAdam(flat, lr=cfg["learning_rate"], in_backward=red)
        repacker = net.range_repacker(flat)
        opt.on_range.append(repacker)
        opt.on_step.append(repacker.finish)
    else:
        opt = FusedAdam(flat, lr=cfg["learning_rate"])
        opt.on_step.append(net.invalidate_packed_weights)
    sched = ExponentialLR(opt, gamma=float(cfg["gamma"]) if cfg["gamma"] != "auto" else 0.99999)
    loss_fn = CELoss()
    per_rank = (len(files) + world - 1) // world
    epoch_length = cfg["training_epoch_length"] or (per_rank + cfg["batch_size"] - 1) // cfg["batch_size"]
    state = TrainerState(epoch_length=epoch_length, max_epochs=cfg["epochs"])
    to_save = {"network": net, "optimizer": opt, "lr_scheduler": sched, "trainer": state}
    ckpt = check_for_checkpoints(cfg)
    if ckpt:
        load_checkpoint(ckpt, to_save, map_location=dev)
        state.rebase(epoch_length, cfg["epochs"])      # finished epochs by the CHECKPOINT's epoch length; this run's data set / --epochs decide the rest
        net.invalidate_packed_weights()
        log(rank, f"resumed from {ckpt}: epoch {state.epoch}, iteration {state.iteration}, lr {opt.lr:.6e}")
    for epoch in range(state.epoch, cfg["epochs"]):
        order = shard_for_rank(len(files), rank, world, epoch=epoch, seed=cfg["seed"])   # DistributedSampler: shared permutation, padded
        done = 0
        for i in range(0, len(order), cfg["batch_size"]):
            q = torch.stack([_load_codes(files[k], cfg, gen) for k in order[i:i + cfg["batch_size"]]])
            (x_in, _), x_tgt = prepare_batch({"quantization": q}, ordering.get_sequence_ordering(), cfg["vocab_size"], device=dev)
            flat.zero_grad()
            logits = net(x_in)
            loss = loss_fn(logits.transpose(1, 2), x_tgt)
            loss.backward()
            opt.step(grad_scale=red.finish())
            sched.step()
            state.iteration += 1
            done += 1
            if state.iteration % cfg["log_every"] == 0:
                log(rank, f"epoch {epoch} it {state.iteration} loss {loss.item():.5f} lr {opt.lr:.3e}")
            if done == epoch_length:
                break
        state.iteration = (epoch + 1) * epoch_length
        if (epoch + 1) % cfg["eval_every"] == 0 and val_files:
            ce = _validation_ce(net, ordering, val_files, cfg, gen, dev, rank, world)
            log(rank, f"epoch {epoch} validation ce {ce:.5f}")
            if rank == 0:
                save_checkpoint(cfg, epoch + 1, to_save, key_metric=-ce)
        if rank == 0 and ((epoch + 1) % cfg["checkpoint_every"] == 0 or epoch + 1 == cfg["epochs"]):
            save_checkpoint(cfg, epoch + 1, to_save)


def inference(cfg, rank, local, world, dev):
    from synthanatomy_amd.utils.transformer import prepare_inference_batch
    files = list_inputs(cfg["validation_subjects"], postfix="quantization_0")[rank::world]
    gen = torch.Generator().manual_seed(cfg["seed"] + rank)
    dims = tuple(_load_codes(files[0], cfg, gen).shape)
    net, ordering = build(cfg, dims, dev)
    path = check_for_checkpoints(cfg)     # starting_epoch > 0: that epoch; else evaluation_checkpoint = "recent" | "best"
    if path:
        load_network_state(net, path)
    net.eval()
    for i in range(0, len(files), cfg["eval_batch_size"]):
        chunk = files[i:i + cfg["eval_batch_size"]]
        q = torch.zeros(len(chunk), *dims, dtype=torch.long)
        (prefix, _), _ = prepare_inference_batch({"quantization": q}, cfg["vocab_size"], device=dev)
        out = net.sample(prefix, temperature=cfg["temperature"], sample=cfg["sample"], top_k=cfg["top_k"])
        for f, o in zip(chunk, out.cpu().numpy()):
            save_npy(o, cfg["outputs_directory"], f, "sample", np.uint16)
    log(rank, f"inference done: {len(files)} samples -> {cfg['outputs_directory']}")


def run(argv):
    from synthanatomy_amd.runtime.ddp import init_distributed
    cfg = parse_flags(argv, DEFAULTS, aliases={"n_embed": "n_embd"})
    if cfg["mode"] not in ("training", "inference"):
        raise ValueError(f"Transformer mode unknown. Was given {cfg['mode']} but choices are ['training', 'inference'].")
    rank, local, world = init_distributed()
    cfg.update(rank=rank, local_rank=local, world_size=world)
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    if cfg.get("deterministic"):
        # upstream: torch.backends.cudnn.deterministic (src/utils/general.py:336-338).  The forward, the FAVOR+ / local-attention / dense kernels and the weight
        # gradients are deterministic already; the flag replaces the remaining fp32 atomics by fixed-order sums: embedding gradients, ReZero gate gradients,
        # LayerNorm weight / bias gradients, bias gradients and the loss (csrc/deterministic.hip; tests/test_deterministic_gpu.py: bit-identical steps)
        from synthanatomy_amd import debug
        debug.set_deterministic(True)
        log(rank, "--deterministic: fixed-order reductions (csrc/deterministic.hip); slower than the default path")
    create_folder_structure(cfg)
    dev = torch.device("cuda", local)
    (training if cfg["mode"] == "training" else inference)(cfg, rank, local, world, dev)


if __name__ == "__main__":
    run(sys.argv[1:])
