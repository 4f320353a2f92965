"""Batch preparation for the transformer stage -- mirror of reference src/utils/transformer.py:239-317.

``prepare_batch``: flatten the code grid, re-order it with the ordering's index sequence, left-pad the begin-of-sequence
token (== ``vocab_size``), and split into (input, target) shifted by one.  ``prepare_inference_batch``: ``[B, 1]`` of BOS.
Integer host/device glue; bit-exact with the reference by construction (pinned in tests/test_host_logic.py).
"""
from __future__ import annotations

from enum import Enum

import numpy as np
import torch
import torch.nn.functional as F


class TransformerConditioningType(Enum):
    NONE = "none"
    BOSREPLACEMENT = "bos_replacement"
    PREPENDING = "prepending"


def _to(t, device, non_blocking):
    return t.to(device=device, non_blocking=non_blocking) if device is not None else t


def _conditionings(batch, conditionings, device, non_blocking):
    if not conditionings:
        return None
    out = []
    for label in conditionings:
        c = batch[label]
        if len(c.shape) == 1:
            c = c[..., None]
        out.append(_to(c.long(), device, non_blocking))
    return out


def prepare_batch(batch, index_sequence, vocab_size, conditionings=None, device=None, non_blocking=False):
    encoded = batch["quantization"]
    encoded = encoded.reshape(encoded.shape[0], -1)
    encoded = encoded[:, index_sequence]
    encoded = F.pad(encoded, (1, 0), "constant", vocab_size)
    encoded = encoded.long()
    conditioned = _conditionings(batch, conditionings, device, non_blocking)
    x_input = _to(encoded[:, :-1], device, non_blocking)
    x_target = _to(encoded[:, 1:], device, non_blocking)
    return (x_input, conditioned), x_target


def prepare_inference_batch(batch, num_embeddings, conditionings=None, device=None, non_blocking=False):
    no_samples = batch["quantization"].shape[0]
    initial = torch.from_numpy(np.repeat(np.array([[num_embeddings]]), no_samples, axis=0)).long()
    conditioned = _conditionings(batch, conditionings, device, non_blocking)
    return (_to(initial, device, non_blocking), conditioned), _to(initial, device, non_blocking)
