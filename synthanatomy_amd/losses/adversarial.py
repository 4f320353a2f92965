"""Adversarial losses of reference src/losses/adversarial/adversarial.py:11-105 and their factories (configure.py:10-38).

``AdversarialLoss(criterion, is_discriminator, weight)``: mean over the patch logits of the criterion on the fakes (target "real" for the
generator, "fake" for the discriminator); the discriminator form averages it with the criterion on the real logits; times ``weight`` (0.005
from the factories).  The three criteria keep the reference's names AND its arithmetic (``vanilla`` is the hinge expression relu(1 -+ x),
``hinge`` the softplus one, ``least_square`` (x - target)^2) -- the names are swapped upstream and a drop-in must not "fix" that.  The logits are
a [B, 1, 18, 26, 18] patch map (155 KB at batch 8): plain tensor arithmetic, the work is in the discriminator's convolutions.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch
import torch.nn.functional as F

CRITERIA = ("vanilla", "hinge", "least_square")   # src/losses/adversarial/utils.py


def get_criterion(criterion: str) -> Callable[[torch.Tensor, bool], torch.Tensor]:
    if criterion == "vanilla":
        return lambda logits, is_real: F.relu(1.0 + (-1 if is_real else 1) * logits)
    if criterion == "hinge":
        return lambda logits, is_real: F.softplus((-1 if is_real else 1) * logits)
    if criterion == "least_square":
        return lambda logits, is_real: (logits - (1 if is_real else 0)) ** 2
    raise ValueError(f"Unknown adversarial criterion {criterion!r}; choices are {list(CRITERIA)}")


class AdversarialLoss(torch.nn.Module):
    def __init__(self, criterion: str = "least_square", is_discriminator: bool = True, weight=None, reduction: str = "mean"):
        super().__init__()
        if reduction not in ("sum", "mean"):
            raise ValueError("Reduction must be either 'sum' or 'mean'")
        self.criterion, self.is_discriminator = criterion, is_discriminator
        self.criterion_function = get_criterion(criterion)
        self._weight = weight
        self.summaries: Dict = {"scalar": {}}

    def forward(self, logits_fake: torch.Tensor, logits_real: torch.Tensor = None) -> torch.Tensor:
        logits_fake = logits_fake.float()
        loss_fake = torch.mean(self.criterion_function(logits_fake, not self.is_discriminator))
        who = "Discriminator" if self.is_discriminator else "Generator"
        self.summaries["scalar"][f"Loss-Adversarial_{who}-Reconstruction"] = loss_fake.detach()
        loss = loss_fake
        if self.is_discriminator:
            loss_real = torch.mean(self.criterion_function(logits_real.float(), True))
            self.summaries["scalar"]["Loss-Adversarial_Discriminator-Originals"] = loss_real.detach()
            loss = 0.5 * (loss + loss_real)
        return self._weight * loss

    def get_summaries(self):
        return self.summaries

    def get_weight(self) -> float:
        return self._weight

    def set_weight(self, weight: float) -> float:
        self._weight = weight
        return self.get_weight()


def get_discriminator_loss(config: dict) -> AdversarialLoss:
    if config["discriminator_loss"] not in CRITERIA:
        raise ValueError(f"Unknown discriminator loss. Available losses are {list(CRITERIA)} but received {config['discriminator_loss']}")
    return AdversarialLoss(criterion=config["discriminator_loss"], is_discriminator=True, weight=0.005)


def get_generator_loss(config: dict) -> AdversarialLoss:
    if config["generator_loss"] not in CRITERIA:
        raise ValueError(f"Unknown generator loss. Available losses are {list(CRITERIA)} but received {config['generator_loss']}")
    return AdversarialLoss(criterion=config["generator_loss"], is_discriminator=False, weight=0.005)
