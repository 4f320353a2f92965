"""One adversarial training iteration -- the arithmetic of reference src/engines/trainer.py:122-294 (``AdversarialTrainer._iteration`` and
``adaptive_adversarial_weight``) without the ignite engine around it.

Generator step: ``loss_G = recon_loss(G(x), x) + w * g_loss(D(G(x)))``; discriminator step on the DETACHED fakes:
``loss_D = w * d_loss(D(fake), D(x))`` with the same ``w``.  ``w`` is 1, or with ``use_adversarial_adaptive_weight`` the ratio
``|d recon_loss / dW_last| / (|d g_loss / dW_last| + 1e-4)`` clamped to [0, 1e4] (``value`` while ``epoch < threshold``), ``W_last`` = the
generator's ``get_last_layer()``.

How it maps to this build: both losses reach ``W_last`` only through the reconstruction, so the two last-layer gradients are the last decoder
stage's weight-gradient kernel applied to ``d recon_loss / d recon`` and to ``d g_loss / d recon`` (``BaselineVQVAE.last_layer_grad``) -- one
discriminator data-gradient pass and two small wgrad launches instead of two extra full decoder backward passes; and because the generator
loss is linear in the two terms, the generator's backward is ONE pass seeded with ``d recon_loss / d recon + w * d g_loss / d recon``.  The
discriminator's own weight gradients are not computed during the generator step (the reference computes and then discards them).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


class AdversarialTrainer:
    def __init__(self, g_network, g_optimizer, g_loss_function, recon_loss_function, d_network, d_optimizer, d_loss_function,
                 use_adversarial_adaptive_weight: bool = False, adaptive_adversarial_weight_threshold: int = 0,
                 adaptive_adversarial_weight_value: float = 1, g_reducer=None, d_reducer=None, g_scheduler=None, d_scheduler=None):
        self.g_network, self.g_optimizer, self.g_loss_function, self.recon_loss_function = g_network, g_optimizer, g_loss_function, recon_loss_function
        self.d_network, self.d_optimizer, self.d_loss_function = d_network, d_optimizer, d_loss_function
        self.use_adversarial_adaptive_weight = use_adversarial_adaptive_weight
        self.adaptive_adversarial_weight_threshold = adaptive_adversarial_weight_threshold
        self.adaptive_adversarial_weight_value = adaptive_adversarial_weight_value
        self.g_reducer, self.d_reducer, self.g_scheduler, self.d_scheduler = g_reducer, d_reducer, g_scheduler, d_scheduler

    # trainer.py:269-294
    def adaptive_adversarial_weight(self, d_recon_loss: Optional[torch.Tensor], d_generator_loss: Optional[torch.Tensor], global_step: int,
                                    threshold: int = 0, value: float = 1):
        """``d_*``: gradients of the two losses with respect to the reconstruction (the only path to the last layer)."""
        if not self.use_adversarial_adaptive_weight:
            return 1
        nll_grads = self.g_network.last_layer_grad(d_recon_loss)
        g_grads = self.g_network.last_layer_grad(d_generator_loss)
        weight = torch.norm(nll_grads) / (torch.norm(g_grads) + 1e-4)
        weight = torch.clamp(weight, 0.0, 1e4).detach()
        if global_step < threshold:
            weight = value
        return weight

    def iteration(self, inputs: torch.Tensor, targets: torch.Tensor, epoch: int) -> Dict[str, torch.Tensor]:
        g_net, d_net = self.g_network, self.d_network
        # ---- generator (trainer.py:157-219)
        g_net.train()
        self.g_optimizer.zero_grad()
        for p in d_net.parameters():
            p.requires_grad_(False)          # the generator step only needs the discriminator's DATA gradient
        g_predictions = g_net(inputs)
        recon = g_predictions["reconstruction"][0]
        logits_fake = d_net(recon.float().contiguous())
        reconstruction_loss = self.recon_loss_function(g_predictions, targets).mean()
        generator_loss = self.g_loss_function(logits_fake).mean()
        d_gen = torch.autograd.grad(generator_loss, recon)[0]
        d_rec = None
        if self.use_adversarial_adaptive_weight:
            d_rec = torch.autograd.grad(reconstruction_loss, recon, retain_graph=True)[0]
        adversarial_weight = self.adaptive_adversarial_weight(d_rec, d_gen, epoch, self.adaptive_adversarial_weight_threshold,
                                                              self.adaptive_adversarial_weight_value)
        total_g = reconstruction_loss.detach() + generator_loss.detach() * adversarial_weight
        torch.autograd.backward([reconstruction_loss, recon], [None, d_gen * adversarial_weight])
        for p in d_net.parameters():
            p.requires_grad_(True)
        self.g_optimizer.step(grad_scale=self.g_reducer.finish() if self.g_reducer is not None else 1.0)
        if self.g_scheduler is not None:
            self.g_scheduler.step()
        # ---- discriminator (trainer.py:221-256)
        d_net.train()
        self.d_optimizer.zero_grad()
        fakes = recon.detach().float().contiguous()
        logits_fake = d_net(fakes)
        logits_real = d_net(inputs.contiguous().detach())
        d_loss = self.d_loss_function(logits_fake, logits_real).mean() * adversarial_weight
        d_loss.backward()
        if self.d_reducer is not None:
            for p in self.d_reducer.flat.params:     # autograd delivered these gradients into the flat buffer: reduce it bucket by bucket
                self.d_reducer.ready(p)
        self.d_optimizer.step(grad_scale=self.d_reducer.finish() if self.d_reducer is not None else 1.0)
        if self.d_scheduler is not None:
            self.d_scheduler.step()
        return {"loss": reconstruction_loss.detach(), "g_loss": total_g, "d_loss": d_loss.detach(), "adversarial_weight": adversarial_weight,
                "pred": g_predictions}
