"""ORACLE (test infrastructure only) -- CPU restatement of the reference's training-loss arithmetic around the VQ-VAE hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the product never does.

* ``adversarial_loss``  -- src/losses/adversarial/adversarial.py:36-105 (criteria "vanilla" / "hinge" / "least_square" with the reference's
  arithmetic under the reference's names), weight 0.005 from src/losses/adversarial/configure.py:19-38
* ``jukebox_loss``      -- src/losses/vqvae/vqvae.py:574-625 (``JukeboxLoss(dimensions=3)``: ortho ``fftn`` over dims (1,2,3,4), amplitude MSE x
  fft_factor + pixel MSE + quantization losses)
* ``adaptive_adversarial_weight`` / ``adversarial_step`` -- src/engines/trainer.py:157-294 (generator step, discriminator step on detached
  fakes, last-layer gradient-norm ratio clamped to [0, 1e4])

Parity status: the two loss functions are PINNED by ``tests/golden/losses.npz`` (values and gradients computed by the reference's own classes,
``tests/golden/make_goldens_losses.py``).  ``adversarial_step`` composes pinned pieces (``vqvae_ref.forward``, ``discriminator_forward``, the two
losses, ``torch.optim.Adam``) in the order of trainer.py; the trainer module itself needs ignite / MONAI to import and has no tests upstream, so
the composition is restated, not pinned.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import vqvae_ref


def _criterion(name: str):
    if name == "vanilla":        # adversarial.py:82-88
        return lambda logits, is_real: F.relu(1.0 + (-1 if is_real else 1) * logits)
    if name == "hinge":          # adversarial.py:90-96
        return lambda logits, is_real: F.softplus((-1 if is_real else 1) * logits)
    if name == "least_square":   # adversarial.py:98-104
        return lambda logits, is_real: (logits - (1 if is_real else 0)) ** 2
    raise ValueError(name)


def adversarial_loss(logits_fake, logits_real=None, criterion="least_square", is_discriminator=True, weight=0.005):
    crit = _criterion(criterion)
    loss = torch.mean(crit(logits_fake.float(), not is_discriminator))
    if is_discriminator:
        loss = 0.5 * (loss + torch.mean(crit(logits_real.float(), True)))
    return weight * loss


def fft_amplitude(images):
    f = torch.fft.fftn(images, dim=tuple(range(1, images.dim())), norm="ortho")
    return torch.sqrt(f.real ** 2 + f.imag ** 2)


def jukebox_loss(out, y, fft_factor=1.0, include_pixel_loss=True):
    pred = out["reconstruction"][0].float()
    y = y.float()
    spectral = F.mse_loss(fft_amplitude(pred), fft_amplitude(y)) * fft_factor
    loss = spectral
    if include_pixel_loss:
        loss = loss + F.mse_loss(pred, y)
    for q in out["quantization_losses"]:
        loss = loss + q.float()
    return loss, spectral


def adaptive_adversarial_weight(reconstruction_loss, generator_loss, last_layer, global_step, use=True, threshold=0, value=1.0):
    """trainer.py:269-294."""
    if not use:
        return 1
    nll = torch.autograd.grad(reconstruction_loss, last_layer, retain_graph=True)[0]
    gg = torch.autograd.grad(generator_loss, last_layer, retain_graph=True)[0]
    w = torch.clamp(torch.norm(nll) / (torch.norm(gg) + 1e-4), 0.0, 1e4).detach()
    if global_step < threshold:
        w = value
    return w


def adversarial_step(g_state: Dict[str, torch.Tensor], d_state: Dict[str, torch.Tensor], cfg, x, *, g_lr, d_lr, g_criterion="least_square",
                     d_criterion="least_square", use_adaptive=False, threshold=0, value=1.0, epoch=0, recon_loss="mse", g_opt=None, d_opt=None,
                     d_layers=3):
    """One iteration of trainer.py:157-256 on leaf-tensor states (modified in place by Adam).  Returns losses, the weight, both gradient sets
    and the optimizers (pass them back in for the next iteration)."""
    g_leaf = [k for k, v in g_state.items() if "quantizer" not in k]
    d_leaf = [k for k, v in d_state.items() if "running" not in k and "num_batches" not in k]
    for k in g_leaf:
        g_state[k].requires_grad_(True)
    for k in d_leaf:
        d_state[k].requires_grad_(True)
    g_opt = g_opt or torch.optim.Adam([g_state[k] for k in g_leaf], lr=g_lr)
    d_opt = d_opt or torch.optim.Adam([d_state[k] for k in d_leaf], lr=d_lr)
    # ---- generator
    g_opt.zero_grad(set_to_none=True)
    out = vqvae_ref.forward(g_state, cfg, x, training=True)
    recon = out["reconstruction"][0]
    logits_fake = vqvae_ref.discriminator_forward(d_state, recon.float().contiguous(), training=True, n_layers=d_layers)
    rl = (vqvae_ref.mse_loss(out, x) if recon_loss == "mse" else jukebox_loss(out, x)[0]).mean()
    gl = adversarial_loss(logits_fake, None, g_criterion, is_discriminator=False).mean()
    last = g_state[f"decoder.0.{2 + 3 * (cfg.n_levels - 1)}.weight"]
    w = adaptive_adversarial_weight(rl, gl, last, epoch, use_adaptive, threshold, value)
    total_g = rl + gl * w
    for k in d_leaf:
        d_state[k].grad = None
    total_g.backward()
    g_grads = {k: g_state[k].grad.clone() for k in g_leaf}
    g_opt.step()
    # ---- discriminator
    for k in d_leaf:
        d_state[k].grad = None
    lf = vqvae_ref.discriminator_forward(d_state, recon.float().contiguous().detach(), training=True, n_layers=d_layers)
    lr_ = vqvae_ref.discriminator_forward(d_state, x.contiguous().detach(), training=True, n_layers=d_layers)
    dl = adversarial_loss(lf, lr_, d_criterion, is_discriminator=True).mean() * w
    dl.backward()
    d_grads = {k: d_state[k].grad.clone() for k in d_leaf}
    d_opt.step()
    return dict(recon_loss=rl.detach(), g_loss=total_g.detach(), d_loss=dl.detach(), weight=w, g_grads=g_grads, d_grads=d_grads, g_opt=g_opt, d_opt=d_opt)
