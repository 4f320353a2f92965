"""``MSELoss`` -- the minimal reconstruction loss of the reference (src/losses/vqvae/vqvae.py:14-71):
``mse(reconstruction[0], y) + sum(quantization_losses)``, with the same ``summaries`` side-channel.  The squared-error
reduction and its gradient are one fused HIP kernel (csrc/elementwise.hip: mse_kernel)."""
from __future__ import annotations

from typing import Dict, List

import torch

from .. import _ffi, debug


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        _ffi.require_gpu()
        a = pred.float().contiguous()
        b = target.float().contiguous().to(a.device)
        n = a.numel()
        acc = torch.zeros(1, dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a) if pred.requires_grad else None
        if debug.deterministic():   # --deterministic: the squared errors are summed in a fixed order (the loss value decides the key-metric checkpoint)
            ws = torch.empty(2048, dtype=torch.float32, device=a.device)
            _ffi.check(_ffi.lib().sa_mse_det(_ffi.ptr(a), _ffi.ptr(b), n, _ffi.ptr(acc), _ffi.ptr(grad), 1.0, _ffi.ptr(ws), _ffi.stream()), "sa_mse_det")
        else:
            _ffi.check(_ffi.lib().sa_mse(_ffi.ptr(a), _ffi.ptr(b), n, _ffi.ptr(acc), _ffi.ptr(grad), 1.0, _ffi.stream()), "sa_mse")
        ctx.grad = grad
        return (acc / n).reshape(())

    @staticmethod
    def backward(ctx, g):
        # re-entrant (the adaptive adversarial weight differentiates the reconstruction loss twice, engines/trainer.py): the stored
        # d mse / d pred is neither consumed nor scaled in place
        grad = ctx.grad
        return (grad * g if grad is not None else None), None


def hip_mse(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _MSEFn.apply(pred, target)


class MSELoss(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.summaries: Dict = {"scalar": {}}

    def forward(self, network_output: Dict[str, List[torch.Tensor]], y: torch.Tensor) -> torch.Tensor:
        rec = hip_mse(network_output["reconstruction"][0], y)
        q = network_output["quantization_losses"]
        loss = rec
        for ql in q:
            loss = loss + ql
        self.summaries["scalar"]["Loss-MSE-Reconstruction"] = rec.detach()
        for i, ql in enumerate(q):
            self.summaries["scalar"][f"Loss-MSE-Quantization_{i}"] = ql.detach()
        return loss

    def get_summaries(self):
        return self.summaries


class JukeboxLoss(torch.nn.Module):
    """``JukeboxLoss(dimensions=3)`` of the reference (src/losses/vqvae/vqvae.py:522-638, selected by ``--loss=jukebox``):
    ``mse(|fftn(pred)|, |fftn(y)|) * fft_factor + mse(pred, y) + sum(quantization_losses)`` with the orthonormal FFT over dims (1, 2, 3, 4).
    The transforms run in rocFFT through ``torch.fft.fftn`` on the device the volumes live on (a [8, 1, 160, 224, 160] batch is 367 MB of
    complex64 per transform; the amplitude / difference passes are a few HBM sweeps, ~1 % of a training step); the pixel term is the fused
    ``sa_mse`` kernel.  Same ``summaries`` keys and ``get/set_fft_factor`` as upstream."""

    def __init__(self, dimensions: int = 3, include_pixel_loss: bool = True, fft_kwargs: Dict = None, reduction: str = "mean"):
        super().__init__()
        self.dimensions, self.include_pixel_loss, self.reduction = dimensions, include_pixel_loss, reduction
        self.fft_factor: float = 1.0
        self.fft_kwargs = {"s": None, "dim": tuple(range(1, dimensions + 2)), "norm": "ortho"} if fft_kwargs is None else fft_kwargs
        self.summaries: Dict = {"scalar": {}}

    def _get_fft_amplitude(self, images: torch.Tensor) -> torch.Tensor:
        f = torch.fft.fftn(images, **self.fft_kwargs)
        return torch.sqrt(f.real ** 2 + f.imag ** 2)

    def forward(self, network_output: Dict[str, List[torch.Tensor]], y: torch.Tensor) -> torch.Tensor:
        y = y.float()
        y_pred = network_output["reconstruction"][0].float()
        with torch.no_grad():
            y_amp = self._get_fft_amplitude(y)
        loss = torch.nn.functional.mse_loss(self._get_fft_amplitude(y_pred), y_amp) * self.fft_factor
        self.summaries["scalar"]["Loss-Spectral-Reconstruction"] = loss.detach()
        self.summaries["scalar"]["Auxiliary-FFT_Factor"] = self.fft_factor
        if self.include_pixel_loss:
            l2 = hip_mse(y_pred, y)
            self.summaries["scalar"]["Loss-MSE-Reconstruction"] = l2.detach()
            loss = loss + l2
        for i, ql in enumerate(network_output["quantization_losses"]):
            ql = ql.float()
            self.summaries["scalar"][f"Loss-MSE-VQ{i}_Commitment_Cost"] = ql.detach()
            loss = loss + ql
        return loss

    def get_summaries(self):
        return self.summaries

    def get_fft_factor(self) -> float:
        return self.fft_factor

    def set_fft_factor(self, fft_factor: float) -> float:
        self.fft_factor = fft_factor
        return self.get_fft_factor()


VQVAE_LOSSES = ("mse", "jukebox")   # of the reference's src/losses/vqvae/utils.py list; the LPIPS / Hartley / WaveGAN families are out of scope


def get_vqvae_loss(config: dict) -> torch.nn.Module:
    """src/losses/vqvae/configure.py:22-52 for the losses this build implements."""
    if config["loss"] == "mse":
        return MSELoss()
    if config["loss"] == "jukebox":
        return JukeboxLoss(dimensions=3)
    raise ValueError(f"Loss function unknown. Was given {config['loss']} but choices are {list(VQVAE_LOSSES)}.")
