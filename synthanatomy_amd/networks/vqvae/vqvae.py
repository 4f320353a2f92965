"""Abstract plugin surface every VQ-VAE network implements (mirror of reference src/networks/vqvae/vqvae.py:8-192)."""
from __future__ import annotations

import abc
from typing import Dict, List, Sequence, Union

import torch
import torch.nn as nn


class VQVAEBase(nn.Module, metaclass=abc.ABCMeta):
    @abc.abstractmethod
    def forward(self, images: torch.Tensor) -> Dict[str, List[torch.Tensor]]:
        raise NotImplementedError

    @abc.abstractmethod
    def encode(self, images: torch.Tensor) -> List[torch.Tensor]:
        raise NotImplementedError

    @abc.abstractmethod
    def quantize(self, encodings: List[torch.Tensor]) -> List[torch.Tensor]:
        raise NotImplementedError

    @abc.abstractmethod
    def decode(self, quantizations: List[torch.Tensor]) -> torch.Tensor:
        raise NotImplementedError

    @abc.abstractmethod
    def index_quantize(self, images: torch.Tensor) -> List[torch.Tensor]:
        raise NotImplementedError

    @abc.abstractmethod
    def decode_samples(self, embedding_indices: List[torch.Tensor]) -> torch.Tensor:
        raise NotImplementedError

    @abc.abstractmethod
    def construct_encoder(self) -> nn.ModuleList:
        raise NotImplementedError

    @abc.abstractmethod
    def construct_quantizer(self) -> nn.ModuleList:
        raise NotImplementedError

    @abc.abstractmethod
    def construct_decoder(self) -> nn.ModuleList:
        raise NotImplementedError

    @abc.abstractmethod
    def get_ema_decay(self) -> Sequence[float]:
        raise NotImplementedError

    @abc.abstractmethod
    def set_ema_decay(self, decay: Union[Sequence[float], float]) -> Sequence[float]:
        raise NotImplementedError

    @abc.abstractmethod
    def get_commitment_cost(self) -> Sequence[float]:
        raise NotImplementedError

    @abc.abstractmethod
    def set_commitment_cost(self, commitment_factor: Union[Sequence[float], float]) -> Sequence[float]:
        raise NotImplementedError

    @abc.abstractmethod
    def get_perplexity(self) -> Sequence[float]:
        raise NotImplementedError

    @abc.abstractmethod
    def get_last_layer(self) -> nn.parameter.Parameter:
        raise NotImplementedError
