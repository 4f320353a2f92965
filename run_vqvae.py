#!/usr/bin/env python3
"""VQ-VAE entry point with the reference's flags and modes (reference run_vqvae.py:538-859):

    python run_vqvae.py run --training_subjects=synthetic:16 --validation_subjects=synthetic:2 --project_directory=/tmp/proj/ \\
        --experiment_name=exp --mode=training|extracting|decoding  [--no_levels=4 --no_channels=256 ...]

MONAI/ignite/fire/deepspeed are not on the target, so the loop is a minimal in-house one: Adam + per-iteration ExponentialLR
(run_vqvae.py:82-91,162), MSE loss ("mse"; the LPIPS/spectral losses are out of scope), optional adversarial component
(least-square GAN, src/losses/adversarial), checkpoints with the reference's keys, uint16 ``.npy`` code files.
Inputs: ``.npy`` volumes (any of dir / glob / csv listing) or ``synthetic:<n>`` (uniform [0,1) volumes of ``--roi`` size).
Multi-GPU: launch with torchrun; one process per GPU, RCCL.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synthanatomy_amd.utils.general import (REQUIRED, create_folder_structure, latest_checkpoint, list_inputs, load_network_state, log,  # noqa: E402
                                            parse_flags, save_checkpoint, save_npy)

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
    if path.startswith("synthetic"):
        return torch.rand(1, *_roi_shape(cfg), generator=gen, device=dev)
    v = torch.from_numpy(np.load(path).astype(np.float32))
    if v.dim() == 3:
        v = v[None]
    if cfg["normalize"]:
        v = (v - v.min()) / (v.max() - v.min() + 1e-8)  # ScaleIntensityd(0, 1)
    return v.to(dev)


def _batches(files, bs, cfg, gen, dev, rank, world):
    files = files[rank::world]  # DistributedSampler-style sharding
    for i in range(0, len(files), bs):
        chunk = files[i:i + bs]
        yield chunk, torch.stack([_load_volume(f, cfg, gen, dev) for f in chunk])


def build_network(cfg, dev):
    from synthanatomy_amd.networks.vqvae.configure import get_vqvae_network
    cfg = dict(cfg)
    cfg["compute_dtype"] = torch.bfloat16 if cfg["amp"] else torch.float32  # --amp=True (reference: fp16 autocast) -> bf16 MFMA
    return get_vqvae_network(cfg).to(dev)


def training(cfg, rank, local, world, dev):
    from synthanatomy_amd.losses.vqvae import MSELoss, hip_mse
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import ExponentialLR, FlatParams, FusedAdam
    if cfg["loss"] != "mse":
        raise NotImplementedError(f"loss={cfg['loss']!r}: the MI355X build implements 'mse' (SURVEY.md section 2 row 12)")
    net = build_network(cfg, dev).train()
    start = 0
    if cfg["starting_epoch"] == -1:
        path, ep = latest_checkpoint(cfg["checkpoint_directory"])
        if path:
            load_network_state(net, path)
            start = ep + 1
            log(rank, f"resumed from {path}")
    flat = FlatParams(net.parameters())
    opt = FusedAdam(flat, lr=cfg["learning_rate"])
    opt.on_step.append(net.invalidate_packed_weights)
    sched = ExponentialLR(opt, gamma=float(cfg["gamma"]) if cfg["gamma"] != "auto" else 0.99999)
    red = GradReducer(flat)
    net.set_grad_sink(red)
    loss_fn = MSELoss()
    disc = d_opt = None
    if cfg["adversarial_component"]:
        from synthanatomy_amd.networks.discriminator.configure import get_discriminator_network
        dcfg = dict(cfg, compute_dtype=torch.bfloat16 if cfg["amp"] else torch.float32)
        disc = get_discriminator_network(dcfg).to(dev).train()
        d_flat = FlatParams(disc.parameters())
        d_opt = FusedAdam(d_flat, lr=cfg["discriminator_learning_rate"])
        d_red = GradReducer(d_flat)
    gen = torch.Generator(device=dev).manual_seed(cfg["seed"] + rank)
    files = list_inputs(cfg["training_subjects"])
    it = 0
    for epoch in range(start, cfg["epochs"]):
        for names, x in _batches(files, cfg["batch_size"], cfg, gen, dev, rank, world):
            flat.zero_grad()
            out = net(x)
            loss = loss_fn(out, x)
            if disc is not None:  # generator step: least-square GAN term, weight 0.005 (losses/adversarial/configure.py:19-38)
                for p in disc.parameters():
                    p.requires_grad_(False)
                logits_fake = disc(out["reconstruction"][0])
                loss = loss + 0.005 * hip_mse(logits_fake, torch.ones_like(logits_fake))
                for p in disc.parameters():
                    p.requires_grad_(True)
            loss.backward()
            opt.step(grad_scale=red.finish())
            sched.step()
            if disc is not None:  # discriminator step
                d_flat.zero_grad()
                rec = out["reconstruction"][0].detach()
                lf, lr_ = disc(rec), disc(x)
                d_loss = 0.5 * (hip_mse(lf, torch.zeros_like(lf)) + hip_mse(lr_, torch.ones_like(lr_)))
                d_loss.backward()
                for p in d_flat.params:  # autograd delivered these grads; reduce them as one flat buffer
                    d_red.ready(p)
                d_opt.step(grad_scale=d_red.finish())
                for st in disc._stages:
                    st.op.invalidate()
            it += 1
            if it % cfg["log_every"] == 0:
                log(rank, f"epoch {epoch} it {it} loss {loss.item():.6f} perplexity {net.get_perplexity()[0].item():.2f} lr {opt.lr:.3e}")
            if cfg["training_epoch_length"] and it % cfg["training_epoch_length"] == 0:
                break
        if rank == 0 and (epoch + 1) % cfg["checkpoint_every"] == 0:
            save_checkpoint(cfg, epoch, net, opt)
    if rank == 0:
        torch.save(net.state_dict(), os.path.join(cfg["checkpoint_directory"], f"model_state_dict_epoch={cfg['epochs'] - 1}.pt"))


def inference(cfg, rank, local, world, dev):
    net = build_network(cfg, dev).eval()
    path, _ = latest_checkpoint(cfg["checkpoint_directory"])
    if path:
        load_network_state(net, path)
        log(rank, f"loaded {path}")
    gen = torch.Generator(device=dev).manual_seed(cfg["seed"] + rank)
    files = list_inputs(cfg["validation_subjects"] if cfg["mode"] == "extracting" else cfg["training_subjects"])
    with torch.no_grad():
        if cfg["mode"] == "extracting":
            for names, x in _batches(files, cfg["eval_batch_size"], cfg, gen, dev, rank, world):
                idx = net.index_quantize(x)[0]
                rec = net.decode_samples([idx])
                for n, i_, r_ in zip(names, idx.cpu().numpy(), rec.float().cpu().numpy()):
                    save_npy(i_, cfg["outputs_directory"], n, "quantization_0", np.uint16)
                    save_npy(r_[0], cfg["outputs_directory"], n, "reconstruction", np.float32)
        else:  # decoding: .npy uint16 code grids -> reconstructions (prepare_decoding_batch: .long())
            files = files[rank::world]
            for f in files:
                idx = torch.from_numpy(np.load(f).astype(np.int64))[None].to(dev)
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
    create_folder_structure(cfg)
    dev = torch.device("cuda", local)
    (training if cfg["mode"] == "training" else inference)(cfg, rank, local, world, dev)


if __name__ == "__main__":
    run(sys.argv[1:])
