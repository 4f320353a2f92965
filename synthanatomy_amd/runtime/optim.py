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
        every = list(params)
        self.params: List[torch.nn.Parameter] = [p for p in every if p.requires_grad]
        assert self.params, "no trainable parameters"
        # position of each trainable parameter in the iterable the caller passed -- torch.optim.Adam(network.parameters()) numbers ALL
        # parameters (the frozen codebook included), and checkpoints use those numbers (FusedAdam.state_dict)
        self.positions = [i for i, p in enumerate(every) if p.requires_grad]
        self.n_all = len(every)
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
    """torch.optim.Adam semantics (no amsgrad) in one launch over the flat buffer.

    ``in_backward=<GradReducer>`` ("optimizer in the shadow of the backward pass"): the reducer calls back as soon as a bucket's gradients are final (all
    its weight-gradient kernels queued and, with several ranks, its collective done) and THAT range is stepped right there, on the reducer's side stream,
    followed by the ``on_range`` callbacks (re-packing the GEMM operands of the layers in the range); ``step()`` then only covers what no bucket reported,
    waits for the side stream and runs ``on_step``.  Same arithmetic per element as the one-launch step (Adam is element-wise), so parameters and moments
    are bit-identical; only the last bucket's slice is still serial after backward (Performer: Adam 0.55 ms + re-pack 0.53 ms = 3.5 % of the step before)."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, in_backward=None):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.step_count = 0
        self.on_step = []   # callbacks, e.g. network.invalidate_packed_weights
        self.on_range = []  # in_backward mode: callbacks(lo, hi) after a range of the flat buffer was stepped (element offsets)
        self._reducer = in_backward
        self._stepped = []  # ranges stepped since the last step()
        if in_backward is not None:
            assert in_backward.flat is flat
            in_backward.on_bucket = self._bucket_step

    def _adam(self, lo: int, hi: int, grad_scale: float):
        f = self.flat
        _ffi.check(_ffi.lib().sa_adam(_ffi.ptr(f.data[lo:hi]), _ffi.ptr(f.grad[lo:hi]), _ffi.ptr(self.m[lo:hi]), _ffi.ptr(self.v[lo:hi]), hi - lo, self.lr,
                                      self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count, grad_scale, _ffi.stream()), "sa_adam")

    def _bucket_step(self, lo: int, hi: int, grad_scale: float):
        if not self._stepped:
            self.step_count += 1        # the first range of a step opens it (bias correction uses the step number)
        self._adam(lo, hi, grad_scale)
        self._stepped.append((lo, hi))
        for cb in self.on_range:
            cb(lo, hi)

    def step(self, grad_scale: float = 1.0):
        f = self.flat
        if self._reducer is None:
            self.step_count += 1
            self._adam(0, f.numel, grad_scale)
        else:
            if not self._stepped:
                self.step_count += 1
            covered, pos = sorted(self._stepped), 0
            for lo, hi in covered + [(f.numel, f.numel)]:      # whatever no bucket reported (normally nothing: GradReducer.finish() launches every bucket)
                if lo > pos:
                    self._adam(pos, lo, grad_scale)
                    for cb in self.on_range:
                        cb(pos, lo)
                pos = max(pos, hi)
            self._stepped = []
        for cb in self.on_step:
            cb()

    def zero_grad(self):
        self.flat.zero_grad()

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (the reference checkpoints ``optimizer`` / ``d_optimizer`` with it, run_vqvae.py:312-326):
        ``state[i] = {step, exp_avg, exp_avg_sq}`` keyed by the parameter's position in ``network.parameters()``, one param group."""
        f = self.flat
        state = {}
        for p, o, pos in zip(f.params, f.offsets, f.positions):
            n = p.numel()
            state[pos] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[o:o + n].view_as(p).clone(),
                          "exp_avg_sq": self.v[o:o + n].view_as(p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(f.n_all))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the torch.optim.Adam layout (ours and the reference's checkpoints) and the round-1 private layout {m, v, step, lr}."""
        if "param_groups" not in sd:
            self.m.copy_(sd["m"])
            self.v.copy_(sd["v"])
            self.step_count, self.lr = int(sd["step"]), float(sd["lr"])
            return
        f = self.flat
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
        ids = list(g["params"])
        steps = set()
        for p, o, pos in zip(f.params, f.offsets, f.positions):
            key = ids[pos] if pos < len(ids) else pos
            ent = sd["state"].get(key)
            n = p.numel()
            if ent is None:        # a parameter that never received a gradient has no state in torch's optimizer
                self.m[o:o + n].zero_()
                self.v[o:o + n].zero_()
                continue
            self.m[o:o + n].view_as(p).copy_(ent["exp_avg"])
            self.v[o:o + n].view_as(p).copy_(ent["exp_avg_sq"])
            steps.add(int(float(ent["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused Adam kernel keeps one bias-correction step for all")
        self.step_count = steps.pop() if steps else 0


class ExponentialLR:
    """lr <- lr * gamma per call (the reference steps it every iteration, run_vqvae.py:162)."""

    def __init__(self, opt: FusedAdam, gamma: float):
        self.opt, self.gamma = opt, gamma
        self.base_lr = opt.lr
        self.last_epoch = 0

    def step(self):
        self.opt.lr *= self.gamma
        self.last_epoch += 1

    def state_dict(self):
        """The keys of ``torch.optim.lr_scheduler.ExponentialLR.state_dict()`` (checkpoint key ``lr_scheduler``)."""
        return {"gamma": self.gamma, "base_lrs": [self.base_lr], "last_epoch": self.last_epoch, "verbose": False, "_step_count": self.last_epoch + 1,
                "_get_lr_called_within_step": False, "_last_lr": [self.opt.lr]}

    def load_state_dict(self, sd):
        self.gamma = float(sd["gamma"])
        self.base_lr = float(sd["base_lrs"][0])
        self.last_epoch = int(sd["last_epoch"])
        self.opt.lr = float(sd["_last_lr"][0]) if sd.get("_last_lr") else self.base_lr * self.gamma ** self.last_epoch


class TrainerState:
    """What ignite's ``Engine.state_dict()`` puts under the checkpoint key ``trainer``: ``iteration``, ``epoch_length``, ``max_epochs``."""

    def __init__(self, epoch_length: int, max_epochs: int):
        self.iteration, self.epoch_length, self.max_epochs = 0, epoch_length, max_epochs

    @property
    def epoch(self) -> int:
        return self.iteration // max(1, self.epoch_length)

    def state_dict(self):
        return {"iteration": self.iteration, "epoch_length": self.epoch_length, "max_epochs": self.max_epochs}

    def rebase(self, epoch_length: int, max_epochs: int):
        """After a resume: THIS run's epoch length / --epochs replace the restored ones (upstream's MaxEpochsHandler,
        src/handlers/general.py:593-610).  The number of FINISHED epochs is taken with the checkpoint's own epoch length first -- the world
        size, batch size or data set may differ from the saved run -- and the iteration counter is re-expressed in the new epoch length, so
        the start epoch, the shard seed and the checkpoint numbering stay those of the saved run."""
        finished = self.epoch
        self.epoch_length, self.max_epochs = int(epoch_length), int(max_epochs)
        self.iteration = finished * self.epoch_length
        return finished

    def load_state_dict(self, sd):
        self.epoch_length = int(sd.get("epoch_length", self.epoch_length))     # ignite restores all three
        self.max_epochs = int(sd.get("max_epochs", self.max_epochs))
        if "iteration" in sd:
            self.iteration = int(sd["iteration"])
        elif "epoch" in sd:      # ignite also accepts {epoch, epoch_length, max_epochs}; round-1 checkpoints stored {"epoch": e} = last finished epoch
            self.iteration = (int(sd["epoch"]) + (0 if "epoch_length" in sd else 1)) * self.epoch_length
