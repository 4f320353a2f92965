"""Factory keyed on ``config["discriminator_network"]`` -- mirror of reference src/networks/discriminator/configure.py:7-20."""
import torch
import torch.nn as nn

from .baseline import BaselineDiscriminator
from .utils import DiscriminatorNetworks


def get_discriminator_network(config: dict) -> nn.Module:
    if config["discriminator_network"] == DiscriminatorNetworks.BASELINE_DISCRIMINATOR.value:
        return BaselineDiscriminator(input_nc=1, ndf=64, n_layers=3, compute_dtype=config.get("compute_dtype", torch.bfloat16))
    raise ValueError(
        f"Discriminator unknown. Was given {config['discriminator_network']} but choices are"
        f" {[d.value for d in DiscriminatorNetworks]}."
    )
