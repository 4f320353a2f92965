"""``baseline_discriminator`` (PatchGAN-3D) on MI355X -- mirror of reference src/networks/discriminator/baseline.py:12-88.

Same constructor (``input_nc``, ``ndf``, ``n_layers``), ``weights_init`` (N(0, 0.02) conv weights, N(1, 0.02) BatchNorm
weights) and ``state_dict`` keys (``main.{0,2,3,5,6,8,9,11}.*``); ``nn.Conv3d`` / ``nn.BatchNorm3d`` are parameter holders.
Forward = implicit-GEMM conv launches (LeakyReLU fused into the first conv's epilogue) + fused BatchNorm/LeakyReLU kernels
(csrc/norm.hip); backward is hand-scheduled with the LeakyReLU derivative fused into the dgrad epilogues.  BatchNorm
statistics are per rank, like the reference's plain ``BatchNorm3d`` under DDP.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn

from ... import _ffi
from ..._ffi import ACT_LRELU, ACT_NONE, MASK_LRELU
from ...engine import ConvOp, cast_pad, vec_of

SLOPE = 0.2


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


class _GradCtx:
    def __init__(self):
        self.grads = {}

    def buf(self, p):
        if p is None:
            return None
        t = torch.zeros_like(p)
        self.grads[p] = t
        return t


class _Stage:
    """conv (+bias) [+ BatchNorm] [+ LeakyReLU]; ``in_lrelu``: the input is a post-LeakyReLU tensor."""

    def __init__(self, conv: nn.Conv3d, bn, act: bool, in_lrelu: bool, dtype, first: bool, last: bool):
        self.conv, self.bn, self.act, self.in_lrelu, self.dtype, self.first, self.last = conv, bn, act, in_lrelu, dtype, first, last
        self.op = ConvOp("conv", conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.weight, conv.bias, dtype)

    def params(self):
        ps = [self.conv.weight] + ([self.conv.bias] if self.conv.bias is not None else [])
        if self.bn is not None:
            ps += [self.bn.weight, self.bn.bias]
        return ps

    def fwd(self, x, tape, training):
        self.op.weight, self.op.bias = self.conv.weight, self.conv.bias
        lib, st = _ffi.lib(), _ffi.stream()
        if self.bn is None:
            y = self.op.fprop(x, act=ACT_LRELU if self.act else ACT_NONE, slope=SLOPE, out_dtype=torch.float32 if self.last else self.dtype,
                              out_channels_stride=self.op.cout)
            if tape is not None:
                tape.append((x, None, None, None))
            return y
        c = self.op.fprop(x, use_bias=self.conv.bias is not None)
        C = self.op.cout
        M = c.numel() // C
        bn = self.bn
        mean = torch.empty(C, dtype=torch.float32, device=c.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(int(lib.sa_bn_sums_ws_floats(C)), dtype=torch.float32, device=c.device)   # 2 C totals (+ 64 per-block partials each in deterministic mode)
        y = torch.empty_like(c)
        use_batch = training or bn.running_mean is None
        _ffi.check(lib.sa_bn_forward(_ffi.ptr(c), _ffi.dtype_id(c.dtype), M, C, _ffi.ptr(bn.weight), _ffi.ptr(bn.bias), _ffi.ptr(bn.running_mean),
                                     _ffi.ptr(bn.running_var), bn.momentum if bn.momentum is not None else 0.1, bn.eps, int(use_batch), SLOPE, _ffi.ptr(y),
                                     _ffi.ptr(mean), _ffi.ptr(rstd), _ffi.ptr(ws), st), "sa_bn_forward")
        if training and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        if tape is not None:
            tape.append((x, c, (mean, rstd), use_batch))
        return y

    def bwd(self, G, saved, gc, need_w=True, need_dx=True):
        """G: gradient wrt this stage's pre-LeakyReLU output (the conv output, or the BatchNorm output).  ``need_w=False`` (generator step: the
        discriminator is frozen) skips the weight / bias gradient launches; ``need_dx=False`` (discriminator step: the image needs no gradient)
        skips the first stage's data gradient."""
        x, c, stats, use_batch = saved
        self.op.weight, self.op.bias = self.conv.weight, self.conv.bias
        lib, st = _ffi.lib(), _ffi.stream()
        vec = vec_of(self.dtype)
        if G.dtype != self.dtype or G.shape[-1] % vec:
            G = cast_pad(G, self.dtype, (G.shape[-1] + vec - 1) // vec * vec)
        if self.bn is not None:
            C = self.op.cout
            M = c.numel() // C
            dc = torch.empty_like(c)
            ws = torch.empty(int(lib.sa_bn_sums_ws_floats(C)), dtype=torch.float32, device=c.device)   # 2 C totals (+ 64 per-block partials each in deterministic mode)
            _ffi.check(lib.sa_bn_backward(_ffi.ptr(c), _ffi.ptr(G), _ffi.dtype_id(c.dtype), M, C, _ffi.ptr(self.bn.weight), _ffi.ptr(stats[0]), _ffi.ptr(stats[1]),
                                          int(use_batch), _ffi.ptr(dc), _ffi.ptr(gc.buf(self.bn.weight)), _ffi.ptr(gc.buf(self.bn.bias)), _ffi.ptr(ws), st),
                       "sa_bn_backward")
            G = dc
        if need_w:
            self.op.wgrad(x, G, gc.buf(self.conv.weight), gc.buf(self.conv.bias))
        if self.first:
            return self.op.dgrad(G, tuple(x.shape[1:4]), out_dtype=torch.float32) if need_dx else None
        return self.op.dgrad(G, tuple(x.shape[1:4]), mask=x if self.in_lrelu else None, mask_mode=MASK_LRELU, slope=SLOPE)


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, record, x, *params):
        _ffi.require_gpu()
        dt = net.compute_dtype
        vec = vec_of(dt)
        h = cast_pad(x.float().permute(0, 2, 3, 4, 1).contiguous(), dt, (x.shape[1] + vec - 1) // vec * vec)
        tape = [] if record else None
        for s in net._stages:
            h = s.fwd(h, tape, net.training)
        ctx.net, ctx.tape, ctx.cin, ctx.need_dx = net, tape, x.shape[1], x.requires_grad
        return h.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, gy):
        net = ctx.net
        gc = _GradCtx()
        G = gy.permute(0, 2, 3, 4, 1).contiguous()
        need_w = any(ctx.needs_input_grad[3:])
        for s, saved in zip(reversed(net._stages), reversed(ctx.tape)):
            G = s.bwd(G, saved, gc, need_w, ctx.need_dx)
        ctx.tape = None
        gx = G[..., : ctx.cin].permute(0, 4, 1, 2, 3) if ctx.need_dx else None
        return (None, None, gx, *[gc.grads.get(p) for p in net._params()])


class BaselineDiscriminator(nn.Module):
    def __init__(self, input_nc=1, ndf=64, n_layers=3, compute_dtype: torch.dtype = torch.bfloat16):
        """PatchGAN discriminator: ``input_nc`` image channels, ``ndf`` filters in the first conv, ``n_layers`` strided convs."""
        super().__init__()
        norm_layer = nn.BatchNorm3d
        use_bias = (norm_layer.func != nn.BatchNorm3d) if type(norm_layer) == functools.partial else (norm_layer != nn.BatchNorm3d)
        kw, padw = 4, 1
        seq = [nn.Conv3d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(SLOPE, True)]
        mult = 1
        for n in range(1, n_layers):
            prev, mult = mult, min(2 ** n, 8)
            seq += [nn.Conv3d(ndf * prev, ndf * mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias), norm_layer(ndf * mult), nn.LeakyReLU(SLOPE, True)]
        prev, mult = mult, min(2 ** n_layers, 8)
        seq += [nn.Conv3d(ndf * prev, ndf * mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias), norm_layer(ndf * mult), nn.LeakyReLU(SLOPE, True)]
        seq += [nn.Conv3d(ndf * mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.main = nn.Sequential(*seq)
        self.apply(weights_init)
        self.compute_dtype = compute_dtype
        if ndf % 8:
            raise NotImplementedError("ndf must be a multiple of 8 (16-byte channels-last vectors)")
        mods = list(self.main)
        stages, i, first = [], 0, True
        while i < len(mods):
            conv = mods[i]
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm3d) else None
            j = i + (2 if bn is not None else 1)
            act = j < len(mods) and isinstance(mods[j], nn.LeakyReLU)
            last = not act
            stages.append(_Stage(conv, bn, act, in_lrelu=not first, dtype=compute_dtype, first=first, last=last))
            first = False
            i = j + (1 if act else 0)
        self._stages = stages

    def _params(self):
        return [p for s in self._stages for p in s.params()]

    def forward(self, input):
        """Standard forward."""
        params = self._params()
        record = torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in params))
        return _DiscFn.apply(self, record, input, *params)
