"""Autoregressive sampling shared by the transformers -- mirror of reference
``src/networks/transformers/transformer.py:8-104`` (``TransformerBase.sample_next_index`` / ``sample``).

``sample`` reproduces the reference's O(N^2) procedure (one full forward over the growing prefix per generated token,
temperature, optional top-k, multinomial or arg-max), then strips the prefix, undoes the sequence ordering and reshapes to
the latent grid.  The forward it calls is the HIP path of the concrete network.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch
from torch.nn import functional as F


def _top_k_logits(logits: torch.Tensor, k: int) -> torch.Tensor:
    v, _ = torch.topk(logits, k)
    out = logits.clone()
    out[out < v[:, [-1]]] = -float("Inf")
    return out


class TransformerBase(torch.nn.Module):
    """Abstract class for transformers."""

    @torch.no_grad()
    def sample_next_index(self, x: torch.Tensor, conditioning: torch.Tensor = None, temperature: float = 1.0, sample: bool = True,
                          top_k: Optional[int] = None) -> torch.Tensor:
        self.eval()
        logits = self(x, conditioning)
        logits = logits[:, -1, :] / temperature
        if top_k is not None:
            logits = _top_k_logits(logits, top_k)
        probs = F.softmax(logits, dim=-1)
        if sample:
            ix = torch.multinomial(probs, num_samples=1)
        else:
            _, ix = torch.topk(probs, k=1, dim=-1)
        return ix

    @torch.no_grad()
    def sample(self, prefix: torch.Tensor, conditioning: torch.Tensor = None, temperature: float = 1.0, sample: bool = True,
               top_k: Optional[int] = None) -> torch.Tensor:
        steps = int(np.prod(self.ordering.dimensions))
        x = prefix
        for _ in range(steps):
            ix = self.sample_next_index(x, conditioning=conditioning, temperature=temperature, sample=sample, top_k=top_k)
            x = torch.cat((x, ix), dim=1)
        x = x[:, prefix.shape[1]:]
        x = x[:, self.ordering.get_revert_sequence_ordering()]
        x = x.reshape(x.shape[0], *self.ordering.dimensions)
        return torch.squeeze(x, 1)  # squeezed so that it can go straight into an nn.Embedding-style lookup

    def forward(self, x: torch.Tensor) -> Any:
        return self(x)
