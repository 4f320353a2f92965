from enum import Enum


class DiscriminatorNetworks(Enum):  # reference src/networks/discriminator/utils.py
    BASELINE_DISCRIMINATOR = "baseline_discriminator"
