"""Autoregressive sampling surface of the transformers (plugin surface of reference ``src/networks/transformers/transformer.py:8-104``:
``TransformerBase.sample_next_index`` / ``sample``).

Two pieces are shared by every decoding path of this build -- the reference-faithful prefix-growing loop below and the stateful O(N) decoder of
``performer.Performer`` -- so that they cannot drift apart:

* ``choose_next``: the decision rule for ONE position (temperature, optional top-k cut, categorical draw or arg-max), with the reference's
  random-number consumption (one ``torch.multinomial`` per generated token) -- ``tests/golden/sample.npz`` pins it against the reference class;
* ``sequence_to_grid``: what happens to a finished token sequence (drop the prefix, undo the sequence ordering, reshape to the latent grid).
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch


def choose_next(last_logits: torch.Tensor, temperature: float = 1.0, sample: bool = True, top_k: Optional[int] = None) -> torch.Tensor:
    """[B, V] logits of the newest position -> [B, 1] token ids."""
    scaled = last_logits / temperature
    if top_k is not None:      # everything below the k-th largest logit of a row is cut off
        kth = torch.topk(scaled, top_k, dim=-1).values[:, -1:]
        scaled = scaled.masked_fill(scaled < kth, float("-inf"))
    probs = torch.softmax(scaled, dim=-1)
    if sample:
        return torch.multinomial(probs, num_samples=1)
    return torch.topk(probs, k=1, dim=-1).indices


def sequence_to_grid(tokens: torch.Tensor, prefix_len: int, ordering) -> torch.Tensor:
    """[B, prefix + prod(dims)] generated sequence -> [B, *dims] code grid in image order (the channel axis of ``ordering.dimensions`` squeezed, so the
    result goes straight into an embedding-style lookup / ``decode_samples``)."""
    body = tokens[:, prefix_len:]
    body = body[:, ordering.get_revert_sequence_ordering()]
    return torch.squeeze(body.reshape(body.shape[0], *ordering.dimensions), 1)


class TransformerBase(torch.nn.Module):
    """Abstract class for transformers."""

    @torch.no_grad()
    def sample_next_index(self, x: torch.Tensor, conditioning: torch.Tensor = None, temperature: float = 1.0, sample: bool = True,
                          top_k: Optional[int] = None) -> torch.Tensor:
        self.eval()
        return choose_next(self(x, conditioning)[:, -1, :], temperature, sample, top_k)

    @torch.no_grad()
    def sample(self, prefix: torch.Tensor, conditioning: torch.Tensor = None, temperature: float = 1.0, sample: bool = True,
               top_k: Optional[int] = None) -> torch.Tensor:
        """The reference procedure: every new token costs one full forward over everything generated so far (O(N^2) token-forwards)."""
        n_prefix, n_new = prefix.shape[1], int(np.prod(self.ordering.dimensions))
        seq = torch.empty(prefix.shape[0], n_prefix + n_new, dtype=prefix.dtype, device=prefix.device)
        seq[:, :n_prefix] = prefix
        for t in range(n_prefix, n_prefix + n_new):
            seq[:, t:t + 1] = self.sample_next_index(seq[:, :t], conditioning=conditioning, temperature=temperature, sample=sample, top_k=top_k)
        return sequence_to_grid(seq, n_prefix, self.ordering)

    def forward(self, x: torch.Tensor) -> Any:
        return self(x)
