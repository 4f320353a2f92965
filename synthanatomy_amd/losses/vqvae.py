"""``MSELoss`` -- the minimal reconstruction loss of the reference (src/losses/vqvae/vqvae.py:14-71):
``mse(reconstruction[0], y) + sum(quantization_losses)``, with the same ``summaries`` side-channel.  The squared-error
reduction and its gradient are one fused HIP kernel (csrc/elementwise.hip: mse_kernel)."""
from __future__ import annotations

from typing import Dict, List

import torch

from .. import _ffi


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        _ffi.require_gpu()
        a = pred.float().contiguous()
        b = target.float().contiguous().to(a.device)
        n = a.numel()
        acc = torch.zeros(1, dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a) if pred.requires_grad else None
        _ffi.check(_ffi.lib().sa_mse(_ffi.ptr(a), _ffi.ptr(b), n, _ffi.ptr(acc), _ffi.ptr(grad), 1.0, _ffi.stream()), "sa_mse")
        ctx.grad = grad
        return (acc / n).reshape(())

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        return (grad.mul_(g) if grad is not None else None), None


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
