"""Network factory keyed on ``config["network"]`` -- mirror of reference src/networks/vqvae/configure.py:14-39.

The EMA-decay warm-up handlers of the reference (configure.py:42-86) are MONAI/ignite plumbing and only work with
``decay_warmup=None``; they are out of scope for the hot path.
"""
from __future__ import annotations

from enum import Enum

import torch

from .baseline import BaselineVQVAE
from .vqvae import VQVAEBase


class VQVAENetworks(Enum):
    BASELINE_VQVAE = "baseline_vqvae"


def get_vqvae_network(config: dict) -> VQVAEBase:
    if config["network"] == VQVAENetworks.BASELINE_VQVAE.value:
        return BaselineVQVAE(
            n_levels=config["no_levels"],
            downsample_parameters=config["downsample_parameters"],
            upsample_parameters=config["upsample_parameters"],
            n_embed=config["num_embeddings"][0],
            embed_dim=config["embedding_dim"][0],
            commitment_cost=config["commitment_cost"][0],
            n_channels=config["no_channels"],
            n_res_channels=config["no_channels"],
            n_res_layers=config["no_res_layers"],
            p_dropout=config["dropout"],
            vq_decay=config["decay"][0],
            use_subpixel_conv=config["use_subpixel_conv"],
            # MI355X-only knob (not in the reference): bf16 MFMA throughput mode or exact-fp32 MFMA parity mode
            compute_dtype=config.get("compute_dtype", torch.bfloat16),
        )
    raise ValueError(f"VQVAE unknown. Was given {config['network']} but choices are {[v.value for v in VQVAENetworks]}.")
