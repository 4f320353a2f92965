"""Flat parameter / gradient storage and the fused Adam step (csrc/elementwise.hip: adam_kernel).

The reference builds ``torch.optim.Adam`` + ``ExponentialLR`` stepped per iteration (run_vqvae.py:82-91,162;
run_transformer.py:108-117).  Here every trainable parameter becomes a view into ONE fp32 buffer (and its ``.grad`` a
view into a second one), so the optimizer step is a single HIP kernel and data-parallel gradient reduction works on large
contiguous ranges of that buffer (runtime/ddp.py).
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from .. import _ffi


class FlatParams:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        # (the buffers live in HBM in production; host tensors are accepted so that the bucketing / reduction logic can be
        #  exercised with the gloo backend -- FusedAdam itself is HIP-only)
        self.offsets = []
        n = 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            v = self.data[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.index = {id(p): i for i, p in enumerate(self.params)}

    def grad_view(self, p):
        i = self.index[id(p)]
        return self.grad[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:  # somebody may have replaced .grad (optimizer.zero_grad(set_to_none=True))
            if p.grad is None or p.grad.data_ptr() != self.grad_view(p).data_ptr():
                p.grad = self.grad_view(p)


class FusedAdam:
    """torch.optim.Adam semantics (no amsgrad) in one launch over the flat buffer."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.step_count = 0
        self.on_step = []  # callbacks, e.g. network.invalidate_packed_weights

    def step(self, grad_scale: float = 1.0):
        self.step_count += 1
        f = self.flat
        _ffi.check(_ffi.lib().sa_adam(_ffi.ptr(f.data), _ffi.ptr(f.grad), _ffi.ptr(self.m), _ffi.ptr(self.v), f.numel, self.lr, self.betas[0],
                                      self.betas[1], self.eps, self.weight_decay, self.step_count, grad_scale, _ffi.stream()), "sa_adam")
        for cb in self.on_step:
            cb()

    def zero_grad(self):
        self.flat.zero_grad()

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.lr}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count, self.lr = sd["step"], sd["lr"]


class ExponentialLR:
    """lr <- lr * gamma per call (the reference steps it every iteration, run_vqvae.py:162)."""

    def __init__(self, opt: FusedAdam, gamma: float):
        self.opt, self.gamma = opt, gamma

    def step(self):
        self.opt.lr *= self.gamma
