#!/usr/bin/env python3
"""VQ-VAE entry point with the reference's flags and modes (reference run_vqvae.py:538-859):

    python run_vqvae.py run --training_subjects=synthetic:16 --validation_subjects=synthetic:2 --project_directory=/tmp/proj/ \\
        --experiment_name=exp --mode=training|extracting|decoding  [--no_levels=4 --no_channels=256 ...]

MONAI/ignite/fire/deepspeed are not on the target, so the loop is a minimal in-house one: Adam + per-iteration ExponentialLR
(run_vqvae.py:82-91,162), losses "mse" and "jukebox" (spectral; the LPIPS family is out of scope), optional adversarial component
(src/engines/trainer.py semantics incl. the adaptive weight; criteria vanilla / hinge / least_square), checkpoints with the reference's keys
(network, optimizer, lr_scheduler, trainer, d_*) restored on resume, uint16 ``.npy`` code files.
Inputs: ``.npy`` volumes (any of dir / glob / csv listing) or ``synthetic:<n>`` (uniform [0,1) volumes of ``--roi`` size).
Multi-GPU: launch with torchrun; one process per GPU, RCCL.
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
    no_augmented_extractions=0, device=0, distributed_port=29500, amp=True, deterministic=False, cuda_benchmark=True, seed=4, epochs=100,
    learning_rate=0.0003, gamma=0.99999, log_every=1, checkpoint_every=1, eval_every=5, augmentation_probability=0.2, augmentation_strength=0,
    loss="mse", adversarial_component=False, finetune_adversarial_component=None, finetune_patience=100,
    discriminator_network="baseline_discriminator", discriminator_learning_rate=0.0005, discriminator_loss="least_square",
    generator_loss="least_square", use_adversarial_adaptive_weight=False, adaptive_adversarial_weight_threshold=0,
    adaptive_adversarial_weight_value=1, initial_factor_value=0, initial_factor_steps=25, max_factor_steps=50, max_factor_value=5, normalize=True,
    roi=((16, 176), (16, 240), (96, 256)), batch_size=3, patch_size=None, eval_batch_size=3, eval_patch_size=None, training_epoch_length=None,
    num_workers=8, prefetch_factor=8, starting_epoch=0, network="baseline_vqvae", use_subpixel_conv=False, use_slim_residual=True, no_levels=3,
    downsample_parameters=((4, 2, 1, 1),) * 3, upsample_parameters=((4, 2, 1, 0, 1),) * 3, no_res_layers=3, no_channels=256, codebook_type="ema",
    num_embeddings=(256,), embedding_dim=(256,), embedding_init=("normal",), commitment_cost=(0.25,), decay=(0.99,), decay_warmup=None,
    max_decay_epochs=50, norm=None, dropout=0.0, act="RELU", output_act=None, evaluation_checkpoint="recent", load_nii_canonical=True,
)


def _roi_shape(cfg):
    return tuple(int(b - a) for a, b in cfg["roi"])


def _load_volume(path, cfg, gen, dev):
    if path.startswith("synthetic"):   # a volume that depends on its NAME only (not on how many were drawn before it): resumable, rank-independent
        g = torch.Generator(device=dev).manual_seed(cfg["seed"] * 1000003 + int(path.split("_")[-1]))
        return torch.rand(1, *_roi_shape(cfg), generator=g, device=dev)
    v = torch.from_numpy(np.load(path).astype(np.float32))
    if v.dim() == 3:
        v = v[None]
    if cfg["normalize"]:
        v = (v - v.min()) / (v.max() - v.min() + 1e-8)  # ScaleIntensityd(0, 1)
    return v.to(dev)


def _batches(files, order, bs, cfg, gen, dev):
    """Batches of this rank's shard (``order`` = indices into ``files`` from utils.general.shard_for_rank)."""
    for i in range(0, len(order), bs):
        chunk = [files[k] for k in order[i:i + bs]]
        yield chunk, torch.stack([_load_volume(f, cfg, gen, dev) for f in chunk])


def build_network(cfg, dev):
    from synthanatomy_amd.networks.vqvae.configure import get_vqvae_network
    cfg = dict(cfg)
    cfg["compute_dtype"] = torch.bfloat16 if cfg["amp"] else torch.float32  # --amp=True (reference: fp16 autocast) -> bf16 MFMA
    return get_vqvae_network(cfg).to(dev)


def _validation_mse(net, files, cfg, gen, dev, rank, world):
    """Mean reconstruction MSE over the validation subjects (the evaluator of run_vqvae.py:248-289 with the metric this build has: the
    reference's key metric is MS-SSIM, a MONAI metric outside the hot path; the selection rule -- keep the best one -- is the same)."""
    from synthanatomy_amd.losses.vqvae import hip_mse
    import torch.distributed as dist
    was = net.training
    net.eval()
    tot = torch.zeros(2, device=dev, dtype=torch.float64)
    with torch.no_grad():
        order = shard_for_rank(len(files), rank, world, shuffle=False, pad=False)
        for _, x in _batches(files, order, cfg["eval_batch_size"], cfg, gen, dev):
            rec = net(x)["reconstruction"][0]
            tot[0] += hip_mse(rec, x).double() * x.shape[0]
            tot[1] += x.shape[0]
    if world > 1:
        dist.all_reduce(tot)
    net.train(was)
    return float(tot[0] / tot[1].clamp(min=1))


def training(cfg, rank, local, world, dev):
    from synthanatomy_amd.engines.trainer import AdversarialTrainer
    from synthanatomy_amd.losses.adversarial import get_discriminator_loss, get_generator_loss
    from synthanatomy_amd.losses.vqvae import get_vqvae_loss
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import ExponentialLR, FlatParams, FusedAdam, TrainerState
    loss_fn = get_vqvae_loss(cfg)
    net = build_network(cfg, dev).train()
    flat = FlatParams(net.parameters())
    red = GradReducer(flat)
    net.set_grad_sink(red)
    from synthanatomy_amd import debug
    if cfg["adversarial_component"] or not debug.host("opt_in_backward"):      # (the adversarial iteration runs several backward passes per optimizer step)
        opt = FusedAdam(flat, lr=cfg["learning_rate"])
        opt.on_step.append(net.invalidate_packed_weights)
    else:                                 # SA_OPT_IN_BACKWARD=1: a bucket's Adam slice + operand re-pack run behind its gradients (runtime/optim.py)
        opt = FusedAdam(flat, lr=cfg["learning_rate"], in_backward=red)
        repacker = net.range_repacker(flat)
        opt.on_range.append(repacker)
        opt.on_step.append(repacker.finish)
    gamma = float(cfg["gamma"]) if cfg["gamma"] != "auto" else 0.99999
    sched = ExponentialLR(opt, gamma=gamma)
    files = list_inputs(cfg["training_subjects"])
    val_files = list_inputs(cfg["validation_subjects"])
    per_rank = (len(files) + world - 1) // world
    epoch_length = cfg["training_epoch_length"] or (per_rank + cfg["batch_size"] - 1) // cfg["batch_size"]
    state = TrainerState(epoch_length=epoch_length, max_epochs=cfg["epochs"])
    to_save = {"network": net, "optimizer": opt, "lr_scheduler": sched, "trainer": state}
    trainer = None
    if cfg["adversarial_component"]:
        from synthanatomy_amd.networks.discriminator.configure import get_discriminator_network
        dcfg = dict(cfg, compute_dtype=torch.bfloat16 if cfg["amp"] else torch.float32)
        disc = get_discriminator_network(dcfg).to(dev).train()
        d_flat = FlatParams(disc.parameters())
        d_opt = FusedAdam(d_flat, lr=cfg["discriminator_learning_rate"])
        d_opt.on_step.append(lambda: [st.op.invalidate() for st in disc._stages])
        d_sched = ExponentialLR(d_opt, gamma=gamma)
        trainer = AdversarialTrainer(net, opt, get_generator_loss(cfg), loss_fn, disc, d_opt, get_discriminator_loss(cfg),
                                     use_adversarial_adaptive_weight=cfg["use_adversarial_adaptive_weight"],
                                     adaptive_adversarial_weight_threshold=cfg["adaptive_adversarial_weight_threshold"],
                                     adaptive_adversarial_weight_value=cfg["adaptive_adversarial_weight_value"],
                                     g_reducer=red, d_reducer=GradReducer(d_flat), g_scheduler=sched, d_scheduler=d_sched)
        to_save.update(d_network=disc, d_optimizer=d_opt, d_lr_scheduler=d_sched)
    # resume (run_vqvae.py:328-345): everything in to_save, except the d_* entries when the adversarial component is being fine-tuned in
    ckpt = check_for_checkpoints(cfg)
    if ckpt:
        to_load = {k: v for k, v in to_save.items() if not (cfg["finetune_adversarial_component"] and k.startswith("d_"))}
        load_checkpoint(ckpt, to_load, map_location=dev)
        state.rebase(epoch_length, cfg["epochs"])      # finished epochs by the CHECKPOINT's epoch length; this run's data set / --epochs decide the rest
        net.invalidate_packed_weights()
        log(rank, f"resumed from {ckpt}: epoch {state.epoch}, iteration {state.iteration}, lr {opt.lr:.6e}")
    gen = torch.Generator(device=dev).manual_seed(cfg["seed"] + rank)
    for epoch in range(state.epoch, cfg["epochs"]):
        # DistributedSampler semantics: one epoch-seeded permutation shared by all ranks, padded so every rank runs the same number of steps
        order = shard_for_rank(len(files), rank, world, epoch=epoch, seed=cfg["seed"])
        done = 0
        for names, x in _batches(files, order, cfg["batch_size"], cfg, gen, dev):
            if trainer is not None:
                res = trainer.iteration(x, x, epoch + 1)      # ignite's state.epoch is 1 during the first epoch (trainer.py:176)
                loss = res["loss"]
            else:
                flat.zero_grad()
                out = net(x)
                loss = loss_fn(out, x)
                loss.backward()
                opt.step(grad_scale=red.finish())
                sched.step()
                res = None
            state.iteration += 1
            done += 1
            if state.iteration % cfg["log_every"] == 0:
                extra = f" g_loss {float(res['g_loss']):.6f} d_loss {float(res['d_loss']):.6f} adv_weight {float(res['adversarial_weight']):.4f}" if res else ""
                log(rank, f"epoch {epoch} it {state.iteration} loss {loss.item():.6f}{extra} perplexity {net.get_perplexity()[0].item():.2f} lr {opt.lr:.3e}")
            if done == epoch_length:
                break
        state.iteration = (epoch + 1) * epoch_length      # (a short last batch list still closes the epoch)
        if (epoch + 1) % cfg["eval_every"] == 0 and val_files:
            mse = _validation_mse(net, val_files, cfg, gen, dev, rank, world)
            log(rank, f"epoch {epoch} validation mse {mse:.6f}")
            if rank == 0:
                save_checkpoint(cfg, epoch + 1, to_save, key_metric=-mse)      # evaluator's key-metric checkpoint, key_metric_n_saved=1
        if rank == 0 and (epoch + 1) % cfg["checkpoint_every"] == 0:
            save_checkpoint(cfg, epoch + 1, to_save)                            # ignite numbers checkpoints by finished epochs
    if rank == 0:
        torch.save(net.state_dict(), os.path.join(cfg["checkpoint_directory"], f"model_state_dict_epoch={cfg['epochs']}.pt"))


def inference(cfg, rank, local, world, dev):
    net = build_network(cfg, dev).eval()
    path = check_for_checkpoints(cfg)     # starting_epoch > 0: that epoch; else evaluation_checkpoint = "recent" | "best"
    if path:
        load_network_state(net, path)
        log(rank, f"loaded {path}")
    gen = torch.Generator(device=dev).manual_seed(cfg["seed"] + rank)
    files = list_inputs(cfg["validation_subjects"] if cfg["mode"] == "extracting" else cfg["training_subjects"])
    with torch.no_grad():
        if cfg["mode"] == "extracting":
            order = shard_for_rank(len(files), rank, world, shuffle=False, pad=False)   # even_divisible=False: no duplicates, no collectives
            for names, x in _batches(files, order, cfg["eval_batch_size"], cfg, gen, dev):
                idx = net.index_quantize(x)[0]
                rec = net.decode_samples([idx])
                for n, i_, r_ in zip(names, idx.cpu().numpy(), rec.float().cpu().numpy()):
                    save_npy(i_, cfg["outputs_directory"], n, "quantization_0", np.uint16)
                    save_npy(r_[0], cfg["outputs_directory"], n, "reconstruction", np.float32)
        else:  # decoding: .npy uint16 code grids -> reconstructions (prepare_decoding_batch: .long())
            files = list_inputs(cfg["training_subjects"], postfix="sample")[rank::world]
            for f in files:
                codes = np.load(f).astype(np.int64)
                if codes.min() < 0 or codes.max() >= net.n_embed:   # e.g. a sampled BOS id (== vocab_size): torch's embedding lookup raises upstream too
                    raise ValueError(f"{f}: code {int(codes.max())} is not a codebook entry (num_embeddings={net.n_embed})")
                idx = torch.from_numpy(codes)[None].to(dev)
                rec = net.decode_samples([idx])
                save_npy(rec[0, 0].float().cpu().numpy(), cfg["outputs_directory"], f, "sample", np.float32)
    log(rank, f"{cfg['mode']} done: {len(files)} inputs -> {cfg['outputs_directory']}")


def run(argv):
    from synthanatomy_amd.runtime.ddp import init_distributed
    cfg = parse_flags(argv, DEFAULTS)
    if cfg["mode"] not in ("training", "extracting", "decoding"):
        raise ValueError(f"VQVAE mode unknown. Was given {cfg['mode']} but choices are ['training', 'extracting', 'decoding'].")
    rank, local, world = init_distributed()
    cfg.update(rank=rank, local_rank=local, world_size=world)
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    if cfg.get("deterministic"):
        # upstream: torch.backends.cudnn.deterministic (src/utils/general.py:336-338).  Here: fixed-order reductions instead of fp32 atomics (quantizer
        # statistics, bias gradients, BatchNorm sums) and the unfused first / last layer routes -- bit-identical runs (tests/test_deterministic_gpu.py)
        from synthanatomy_amd import debug
        debug.set_deterministic(True)
        log(rank, "--deterministic: fixed-order reductions (csrc/deterministic.hip); slower than the default path")
    create_folder_structure(cfg)
    dev = torch.device("cuda", local)
    (training if cfg["mode"] == "training" else inference)(cfg, rank, local, world, dev)


if __name__ == "__main__":
    run(sys.argv[1:])
