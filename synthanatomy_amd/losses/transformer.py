"""``CELoss`` -- mirror of reference src/losses/transformer/transformer.py:10-36: cross entropy over logits ``[B, V, N]``
(the training inferer transposes the network output, src/inferer/transformer.py:28-29), reduction "mean" or "sum", with the
``summaries`` side channel.  log-softmax, NLL and the gradient are one fused HIP kernel (csrc/performer.hip: ce_kernel)."""
from __future__ import annotations

from typing import Dict

import torch

from .. import _ffi, debug


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_bvn, target, mean):
        _ffi.require_gpu()
        rows = logits_bvn.float().transpose(1, 2).contiguous()  # [B,N,V]; free when it is our own transposed output
        B, N, V = rows.shape
        R = B * N
        tgt = target.to(rows.device).long().contiguous().view(-1)
        acc = torch.zeros(1, dtype=torch.float32, device=rows.device)
        d = torch.empty_like(rows) if logits_bvn.requires_grad else None
        scale = (1.0 / R) if mean else 1.0
        if debug.deterministic():   # --deterministic: per-row losses, then ONE block adds them in a fixed order (the default kernel adds them with fp32 atomics)
            row_loss = torch.empty(R, dtype=torch.float32, device=rows.device)
            _ffi.check(_ffi.lib().sa_cross_entropy_rows(_ffi.ptr(rows), _ffi.ptr(tgt), R, V, _ffi.ptr(row_loss), _ffi.ptr(d), _ffi.SA_F32, scale, _ffi.stream()),
                       "sa_cross_entropy_rows")
            _ffi.check(_ffi.lib().sa_sum_det(_ffi.ptr(row_loss), R, _ffi.ptr(acc), 0, _ffi.stream()), "sa_sum_det")
        else:
            _ffi.check(_ffi.lib().sa_cross_entropy(_ffi.ptr(rows), _ffi.ptr(tgt), R, V, _ffi.ptr(acc), _ffi.ptr(d), _ffi.SA_F32, scale, _ffi.stream()), "sa_cross_entropy")
        ctx.d = d
        return (acc * scale).reshape(())

    @staticmethod
    def backward(ctx, g):
        d = ctx.d
        ctx.d = None
        return (d.mul_(g).transpose(1, 2) if d is not None else None), None, None


class _WeightedCEFn(torch.autograd.Function):
    """``F.cross_entropy(..., weight=w)`` (reference transformer.py:28-30 with a class-weight vector): per-row log-sum-exp / NLL / gradient from the same fused
    kernel in its per-row form (sa_cross_entropy_rows), then loss = sum_r w[y_r] nll_r (/ sum_r w[y_r] for "mean"; targets equal to -100 carry weight 0, as
    torch's ignore_index does) and d logits_r scaled by w[y_r] (/ the same denominator).  The per-row scaling is a broadcast multiply on the device."""

    @staticmethod
    def forward(ctx, logits_bvn, target, weight, mean):
        _ffi.require_gpu()
        rows = logits_bvn.float().transpose(1, 2).contiguous()
        B, N, V = rows.shape
        R = B * N
        tgt = target.to(rows.device).long().contiguous().view(-1)
        row_loss = torch.empty(R, dtype=torch.float32, device=rows.device)
        d = torch.empty_like(rows) if logits_bvn.requires_grad else None
        _ffi.check(_ffi.lib().sa_cross_entropy_rows(_ffi.ptr(rows), _ffi.ptr(tgt), R, V, _ffi.ptr(row_loss), _ffi.ptr(d), _ffi.SA_F32, 1.0, _ffi.stream()),
                   "sa_cross_entropy_rows")
        w = weight.to(device=rows.device, dtype=torch.float32)
        wr = torch.where(tgt >= 0, w[tgt.clamp(min=0, max=V - 1)], torch.zeros((), device=rows.device))     # (-100: ignored rows weigh nothing)
        if mean:
            wr = wr / wr.sum()
        if d is not None:
            d.mul_(wr.view(B, N, 1))
        ctx.d = d
        terms = (row_loss * wr).contiguous()
        if debug.deterministic():     # fixed-order sum, like every other reduction of the deterministic mode
            out = torch.empty((), dtype=torch.float32, device=rows.device)
            _ffi.check(_ffi.lib().sa_sum_det(_ffi.ptr(terms), terms.numel(), _ffi.ptr(out), 0, _ffi.stream()), "sa_sum_det")
            return out
        return terms.sum()

    @staticmethod
    def backward(ctx, g):
        d = ctx.d
        ctx.d = None
        return (d.mul_(g).transpose(1, 2) if d is not None else None), None, None, None


class CELoss(torch.nn.Module):
    def __init__(self, weight=None, size_average: bool = None, reduce: bool = None, reduction: str = "mean"):
        super().__init__()
        if reduction not in ["sum", "mean"]:
            raise ValueError("Reduction must be either 'sum' or 'mean'")
        # a buffer like torch.nn.CrossEntropyLoss.weight: follows .to(device) and is moved once, not per step.  ("mean" with every target ignored divides by a
        # zero weight sum and returns NaN, as torch does.)
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32).clone())
        self.reduction = reduction
        self.summaries: Dict = {"scalar": {}}

    def forward(self, y_pred: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if self.weight is not None:
            if self.weight.device != y_pred.device:
                self.weight = self.weight.to(y_pred.device)        # one copy, then resident
            loss = _WeightedCEFn.apply(y_pred, y, self.weight, self.reduction == "mean")
        else:
            loss = _CEFn.apply(y_pred, y, self.reduction == "mean")
        self.summaries["scalar"]["Loss-CE-Prediction"] = loss.detach()
        return loss

    def get_summaries(self):
        return self.summaries
