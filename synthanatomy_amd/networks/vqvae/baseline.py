"""``baseline_vqvae`` on MI355X: the reference's plugin surface over hand-written HIP kernels.

Mirrors reference ``src/networks/vqvae/baseline.py`` -- ``Quantizer_impl`` (:24-91), ``Quantizer`` (:94-147),
``ResidualLayer`` (:150-160), ``BaselineVQVAE`` (:163-362): same constructor arguments, methods, return types (dicts of
1-element lists) and ``state_dict`` keys (``encoder.0.{0,2,3,..}``, ``quantizer.0.impl.{weight,N,embed_avg,
embedding.weight}``, ``decoder.0.{0,1,2,4,..}``), so checkpoints and callers are interchangeable.

What differs is everything underneath: the ``nn.Conv3d``/``nn.ConvTranspose3d`` objects are only parameter holders; the
arithmetic is a chain of implicit-GEMM MFMA launches (``synthanatomy_amd.engine``) over channels-last activations in
``compute_dtype`` (``torch.bfloat16`` = throughput mode, ``torch.float32`` = exact-f32 MFMA parity mode), and the
quantizer is one fused HIP kernel + one RCCL all-reduce of the packed EMA statistics on a side stream.  Autograd sees
three nodes (encoder, quantizer, decoder) whose backward passes are hand-scheduled dgrad/wgrad launches with the ReLU
masks and residual adds fused into the GEMM epilogues.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist
import torch.nn as nn

from ... import _ffi, debug
from ..._ffi import ACT_NONE, ACT_RELU, MASK_NONE, MASK_POS
from ...engine import ConvOp, PackSet, _launch, _ru, cast_pad, vec_of
from .vqvae import VQVAEBase


# ------------------------------------------------------------------------------------------------ quantizer
class Quantizer_impl(nn.Module):
    def __init__(self, n_embed: int, embed_dim: int, eps: float):
        super().__init__()
        self.embed_dim, self.n_embed, self.eps = embed_dim, n_embed, eps
        self.embedding = nn.Embedding(n_embed, embed_dim)
        self.embedding.weight.requires_grad = False
        self.weight = self.embedding.weight  # same Parameter under a second name, as the reference (baseline.py:33)
        self.register_buffer("N", torch.zeros(n_embed))
        self.register_buffer("embed_avg", self.weight.data.clone())
        self._stats = None      # packed [K + K*D] fp32: counts | dw  (one all-reduce instead of the reference's two)
        self._scratch = None    # wnorm[K] | sqerr[1] | perplexity[1]
        self._side = None
        self._ema_done = None
        self.process_group = None

    # ---- device buffers -------------------------------------------------------------------------------------
    def _work_buffers(self, dev):
        K, D = self.n_embed, self.embed_dim
        if self._stats is None or self._stats.device != dev:
            self._stats = torch.zeros(K + K * D, dtype=torch.float32, device=dev)
            self._scratch = torch.zeros(K + 2, dtype=torch.float32, device=dev)
        return self._stats, self._scratch

    def wait_ema(self):
        """Make the current stream wait for an EMA update still running on the side stream."""
        if self._ema_done is not None:
            torch.cuda.current_stream().wait_event(self._ema_done)
            self._ema_done = None

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float(): readers of the buffers the side stream may still be writing
        if self._ema_done is not None:
            self.wait_ema()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        if self._ema_done is not None:
            self.wait_ema()
        return super()._load_from_state_dict(*args, **kwargs)

    # ---- forward (Quantizer_impl.forward, baseline.py:38-87) --------------------------------------------------
    def forward(self, x: torch.Tensor, decay: float, commitment_cost: float):
        zq, loss, idx, _ = _VQFn.apply(x, self, float(decay), float(commitment_cost), self.training)
        return zq, loss, idx

    def embed(self, embedding_indices: torch.Tensor) -> torch.Tensor:
        """indices [B,h,w,d] -> codes [B,D,h,w,d] (baseline.py:89-91)."""
        _ffi.require_gpu()
        self.wait_ema()
        idx = embedding_indices.to(device=self.weight.device, dtype=torch.int64).contiguous()
        out = torch.empty((*idx.shape, self.embed_dim), dtype=torch.float32, device=idx.device)
        _ffi.check(_ffi.lib().sa_vq_embed(_ffi.ptr(self.weight), _ffi.ptr(idx), idx.numel(), self.n_embed, self.embed_dim, _ffi.ptr(out),
                                          _ffi.SA_F32, _ffi.stream()), "sa_vq_embed")
        return out.permute(0, 4, 1, 2, 3)


class _VQFn(torch.autograd.Function):
    """x [B,D,h,w,d] (any strides) -> (zq_st [B,D,h,w,d], loss, idx [B,h,w,d], perplexity)."""

    @staticmethod
    def forward(ctx, x, q: Quantizer_impl, decay, beta, training):
        _ffi.require_gpu()
        lib, st = _ffi.lib(), _ffi.stream()
        K, D = q.n_embed, q.embed_dim
        b = x.shape[0]
        sp = tuple(x.shape[2:])
        rows = x.float().permute(0, 2, 3, 4, 1).contiguous()  # no copy when x is the encoder's channels-last output
        M = rows.numel() // D
        dev = rows.device
        q.wait_ema()
        stats, scratch = q._work_buffers(dev)
        stats.zero_()
        scratch.zero_()
        counts, dw = stats[:K], stats[K:]
        wnorm, sqerr, ppl = scratch[:K], scratch[K:K + 1], scratch[K + 1:K + 2]
        idx = torch.empty((b, *sp), dtype=torch.int64, device=dev)
        zq = torch.empty_like(rows)
        cb = q.weight.detach()
        cb_used = cb.clone() if training else cb  # backward needs the pre-update codebook (baseline.py:63)
        _ffi.check(lib.sa_vq_assign(_ffi.ptr(rows), _ffi.ptr(cb), M, K, D, _ffi.ptr(idx), _ffi.ptr(zq), None, _ffi.ptr(counts), _ffi.ptr(dw),
                                    _ffi.ptr(sqerr), _ffi.ptr(wnorm), st), "sa_vq_assign")
        if debug.deterministic():   # counts / dw / commitment error again, summed in a fixed order (the launch above used fp32 atomics)
            err_ws = torch.empty(K, dtype=torch.float32, device=dev)
            _ffi.check(lib.sa_vq_stats_det(_ffi.ptr(rows), _ffi.ptr(cb), _ffi.ptr(idx), M, K, D, _ffi.ptr(counts), _ffi.ptr(dw), _ffi.ptr(sqerr), _ffi.ptr(err_ws), st),
                       "sa_vq_stats_det")
        _ffi.check(lib.sa_vq_perplexity(_ffi.ptr(counts), K, M, _ffi.ptr(ppl), st), "sa_vq_perplexity")
        loss = (sqerr * (beta / float(M * D))).reshape(())
        perplexity = ppl.clone().reshape(())
        if training:
            # EMA update (baseline.py:66-80).  The statistics are SUMMED over ranks; the all-reduce and the update run on a
            # side stream because nothing downstream in this step reads the new codebook.
            use_dist = dist.is_available() and dist.is_initialized() and (dist.get_world_size(q.process_group) > 1 or debug.host("ddp_single_rank"))
            if q._side is None:
                q._side = torch.cuda.Stream(device=dev)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(q._side):
                q._side.wait_event(ready)
                if use_dist:
                    from ...runtime.ddp import all_reduce_sum
                    all_reduce_sum(stats, q.process_group)
                _ffi.check(lib.sa_vq_ema_update(_ffi.ptr(q.N), _ffi.ptr(q.embed_avg), _ffi.ptr(cb), _ffi.ptr(counts), _ffi.ptr(dw), K, D, decay, q.eps,
                                                _ffi.stream()), "sa_vq_ema_update")
                done = torch.cuda.Event()
                done.record()
            q._ema_done = done
        ctx.save_for_backward(rows, cb_used, idx)
        ctx.beta, ctx.q = beta, q
        ctx.mark_non_differentiable(idx, perplexity)
        return zq.permute(0, 4, 1, 2, 3), loss, idx, perplexity

    @staticmethod
    def backward(ctx, g_zq, g_loss, _gi, _gp):
        # the EMA update of this step ran on the side stream during the decoder / loss; from here on (optimizer step, DDP buffer broadcasts,
        # handlers reading impl.weight / N / embed_avg) the main stream is ordered after it
        ctx.q.wait_ema()
        rows, cb, idx = ctx.saved_tensors
        D = rows.shape[-1]
        M = rows.numel() // D
        gz = None
        if g_zq is not None:
            gz = g_zq.permute(0, 2, 3, 4, 1).contiguous()
            if gz.dtype not in (torch.float32, torch.bfloat16):
                gz = gz.float()
        gl = g_loss.float().reshape(1).contiguous() if g_loss is not None else None
        dz = torch.empty_like(rows)
        _ffi.check(_ffi.lib().sa_vq_backward(_ffi.ptr(rows), _ffi.ptr(cb), _ffi.ptr(idx), _ffi.ptr(gz), _ffi.dtype_id(gz.dtype) if gz is not None else 0,
                                             _ffi.ptr(gl), ctx.beta, M, D, _ffi.ptr(dz), _ffi.SA_F32, _ffi.stream()), "sa_vq_backward")
        return dz.permute(0, 4, 1, 2, 3), None, None, None, None


class Quantizer(nn.Module):
    def __init__(self, n_embed, embed_dim, commitment_cost=0.25, decay=0.99, eps=1e-5):
        super().__init__()
        self.impl = Quantizer_impl(n_embed, embed_dim, eps)
        self.n_embed = n_embed
        self.commitment_cost = commitment_cost
        self.decay = decay
        self.perplexity_code: torch.Tensor = torch.rand(1)

    def forward(self, x):
        zq, loss, idx, ppl = _VQFn.apply(x, self.impl, float(self.decay), float(self.commitment_cost), self.impl.training)
        self.perplexity_code = ppl
        return zq, loss

    def get_ema_decay(self) -> float:
        return self.decay

    def set_ema_decay(self, decay: float) -> float:
        self.decay = decay
        return self.get_ema_decay()

    def get_commitment_cost(self) -> float:
        return self.commitment_cost

    def set_commitment_cost(self, commitment_cost) -> float:
        self.commitment_cost = commitment_cost
        return self.get_commitment_cost()

    def get_perplexity(self) -> torch.Tensor:
        return self.perplexity_code

    def embed(self, embedding_indices: torch.Tensor) -> torch.Tensor:
        return self.impl.embed(embedding_indices=embedding_indices)

    def quantize(self, encodings: torch.Tensor):
        return self.impl(encodings, self.decay, self.commitment_cost)


# ------------------------------------------------------------------------------------------------ parameter holders
class ResidualLayer(nn.Sequential):
    """relu(x + conv1x1(dropout(relu(conv3x3(x)))))  (baseline.py:150-160).  Holder of the two convs' parameters; the
    arithmetic runs in ``_ResStage``.  p_dropout > 0 (not the README configuration): training steps take the two-launch form of the block with the
    Dropout3d channel mask applied between the launches (``_ResStage._dropout_mask``); p == 0 and eval: the fused one-launch block."""

    def __init__(self, n_channels, n_res_channels, p_dropout):
        super().__init__(nn.Conv3d(n_channels, n_res_channels, kernel_size=3, padding=1), nn.ReLU(True), nn.Dropout3d(p_dropout),
                         nn.Conv3d(n_res_channels, n_channels, kernel_size=1))
        if not 0.0 <= p_dropout < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), but got {p_dropout}")
        self.p_dropout = float(p_dropout)


# ------------------------------------------------------------------------------------------------ chain stages
class _Act:
    """An activation of an f16 FORWARD chain (the encoder in throughput mode): ``f`` (float16) feeds the next forward launch, ``s`` (bfloat16, written
    by the same launch: ``sa_epilogue.out_lp``) is what the bf16 backward pass reads -- None when nothing is recorded (eval)."""
    __slots__ = ("f", "s")

    def __init__(self, f, s):
        self.f, self.s = f, s


def _fs(x):
    return (x.f, x.s) if isinstance(x, _Act) else (x, x)


class _ConvStage:
    """conv / convT (+ReLU).  ``in_act``: the stage's input is a post-ReLU tensor, so the data gradient it hands back is
    masked with (x > 0) in the dgrad epilogue."""

    def __init__(self, mod: nn.Module, kind, act, in_act, dtype, out_f32=False, need_dx=True, fwd_dtype=None):
        k, s, p = mod.kernel_size[0], mod.stride[0], mod.padding[0]
        self.mod, self.act, self.in_act, self.out_f32, self.need_dx = mod, act, in_act, out_f32, need_dx
        self.op = ConvOp(kind, mod.in_channels, mod.out_channels, k, s, p, mod.weight, mod.bias, dtype, fwd_dtype=fwd_dtype)
        self.dtype = dtype
        self.mixed = self.op.fwd_dtype != dtype

    def params(self):
        return [self.mod.weight, self.mod.bias]

    def _sync(self):
        self.op.weight, self.op.bias = self.mod.weight, self.mod.bias

    def fwd(self, x, tape):
        self._sync()
        cout = self.op.cout
        xf, xs = _fs(x)
        if tape is not None:
            tape.append((xs,))
        if self.mixed and not self.out_f32:
            if tape is None:
                return _Act(self.op.fprop(xf, act=self.act, out_channels_stride=cout), None)
            y, _, ys = self.op.fprop(xf, act=self.act, out_channels_stride=cout, want_lp=True)
            return _Act(y, ys)
        return self.op.fprop(xf, act=self.act, out_dtype=torch.float32 if self.out_f32 else self.op.fwd_dtype, out_channels_stride=cout)

    def bwd(self, G, saved, grads, wgrad_only=False):
        (x,) = saved
        self._sync()
        vec = vec_of(self.dtype)
        if G.shape[-1] % vec or G.dtype != self.dtype:
            G = cast_pad(G, self.dtype, (G.shape[-1] + vec - 1) // vec * vec)
        dw, db = grads.buf(self.mod.weight), grads.buf(self.mod.bias)
        grads.wgrad(self.op, x, G, dw, db)
        grads.done(self.mod.weight, self.mod.bias)
        if not self.need_dx or wgrad_only:
            return None
        return self.op.dgrad(G, tuple(x.shape[1:4]), mask=x if self.in_act else None, mask_mode=MASK_POS)


class _Conv1Stage:
    """First encoder layer Conv3d(1 -> C, k4 s2 p1) (+ReLU; baseline.py:218-226).  With ONE input channel the implicit GEMM pads the reduction
    8x (channels travel in 16-byte vectors) and spends 2.1 ms per launch at 160x224x160 / batch 8 plus 0.8 ms padding the volume.  Here the 64
    taps are the whole reduction.  bf16, 128 channels: csrc/conv1.hip gathers the taps of a tile of cells from the fp32 volume into LDS and
    runs the MFMA from there, forward (+bias, ReLU) and weight / bias gradient -- only the output / the incoming gradient touch HBM.
    Other widths / fp32: sa_convt1_im2col writes Xc[cell][tap] = x[2 cell - 1 + tap] (the gather the last decoder layer's backward uses)
    and the dense kernels treat the taps as the channels of a 1x1x1 convolution.  No data gradient: this is the network input.  Odd extents
    or tiny volumes take the generic stage."""

    GEMM_MIN_CELLS = 4096

    def __init__(self, mod: nn.Conv3d, act, dtype, fwd_dtype=None):
        self.mod, self.act, self.dtype = mod, act, dtype
        self.op = ConvOp("conv", 64, mod.out_channels, 1, 1, 0, mod.weight.view(mod.out_channels, 64, 1, 1, 1), mod.bias, dtype, fwd_dtype=fwd_dtype)
        self.mixed = self.op.fwd_dtype != dtype
        self.fallback = _ConvStage(mod, "conv", act, in_act=False, dtype=dtype, need_dx=False, fwd_dtype=fwd_dtype)
        self.fallback_op = self.fallback.op

    @staticmethod
    def applicable(mod) -> bool:
        return (isinstance(mod, nn.Conv3d) and mod.in_channels == 1 and mod.kernel_size == (4, 4, 4) and mod.stride == (2, 2, 2)
                and mod.padding == (1, 1, 1) and mod.dilation == (1, 1, 1) and mod.out_channels % 8 == 0)

    def params(self):
        return [self.mod.weight, self.mod.bias]

    def _sync(self):
        self.op.weight, self.op.bias = self.mod.weight.view(self.mod.out_channels, 64, 1, 1, 1), self.mod.bias

    def fwd(self, x, tape):
        """x: the raw fp32 volume [N, D, H, W]."""
        N, D, H, W = x.shape
        Do, Ho, Wo, cout = D // 2, H // 2, W // 2, self.op.cout
        fused = (self.dtype == torch.bfloat16 and cout == 128 and not debug.host("no_conv1_fused") and not debug.deterministic())   # (its weight gradient ends in fp32 atomics)
        if (D % 2 or H % 2 or W % 2 or Wo < 2 or N * Do * Ho * Wo < self.GEMM_MIN_CELLS or debug.host("no_conv1_gemm")   # generic stage
                or (self.mixed and not fused)):      # (an f16 forward chain has the fused kernel and the generic stage, not the im2col route)
            vec = vec_of(self.dtype)
            xc = cast_pad(x.unsqueeze(-1), self.op.fwd_dtype, vec)
            return self.fallback.fwd(_Act(xc, cast_pad(x.unsqueeze(-1), self.dtype, vec) if tape is not None else None) if self.mixed else xc, tape)
        self._sync()
        lib, st = _ffi.lib(), _ffi.stream()
        if fused:
            # csrc/conv1.hip: taps gathered into LDS straight from the volume; nothing but the output touches HBM
            wpk = self.op.packed_fwd_operand(N, (Do, Ho, Wo))
            y = torch.empty((N, Do, Ho, Wo, cout), dtype=self.op.fwd_dtype, device=x.device)
            if self.mixed:   # f16 taps / weights / output + the bf16 copy the first residual block's backward reads
                ys = torch.empty((N, Do, Ho, Wo, cout), dtype=self.dtype, device=x.device) if tape is not None else None
                _launch("conv1_fwd_f16_kernel", 2.0 * y.numel() * 64, nbytes=y.numel() * 2.0 * (2 if ys is not None else 1) + x.numel() * 4.0,
                        fn=lambda: _ffi.check(lib.sa_conv1_fwd_f16(_ffi.ptr(x), _ffi.ptr(wpk), _ffi.ptr(self.mod.bias), _ffi.ptr(y), _ffi.ptr(ys), N, Do, Ho, Wo, cout,
                                                                  self.act, st), "sa_conv1_fwd_f16"))
            else:
                _launch("conv1_fwd_kernel", 2.0 * y.numel() * 64, nbytes=y.numel() * 2.0 + x.numel() * 4.0,
                        fn=lambda: _ffi.check(lib.sa_conv1_fwd(_ffi.ptr(x), _ffi.ptr(wpk), _ffi.ptr(self.mod.bias), _ffi.ptr(y), N, Do, Ho, Wo, cout, self.act, st),
                                              "sa_conv1_fwd"))
            if tape is not None:
                tape.append((x, "fused"))
            return _Act(y, ys) if self.mixed else y
        Xc = torch.empty((N, Do, Ho, Wo, 64), dtype=self.dtype, device=x.device)
        _ffi.check(lib.sa_convt1_im2col(_ffi.ptr(x), _ffi.dtype_id(self.dtype), _ffi.ptr(Xc), None, N, Do, Ho, Wo, st), "sa_convt1_im2col")
        y = self.op.fprop(Xc, act=self.act, out_dtype=self.dtype, out_channels_stride=cout)
        if tape is not None:
            tape.append((Xc, "im2col"))
        return y

    def bwd(self, G, saved, grads):
        if len(saved) == 1:
            return self.fallback.bwd(G, saved, grads)
        self._sync()
        vec = vec_of(self.dtype)
        if G.shape[-1] % vec or G.dtype != self.dtype:
            G = cast_pad(G, self.dtype, (G.shape[-1] + vec - 1) // vec * vec)
        dw, db = grads.buf(self.mod.weight), grads.buf(self.mod.bias)
        if saved[1] == "fused":
            x = saved[0]
            N, D, H, W = x.shape
            _launch("conv1_wgrad_kernel", 2.0 * G.numel() * 64,
                    lambda: _ffi.check(_ffi.lib().sa_conv1_wgrad(_ffi.ptr(x), _ffi.ptr(G), _ffi.ptr(dw), _ffi.ptr(db), N, D // 2, H // 2, W // 2, self.op.cout,
                                                                 _ffi.stream()), "sa_conv1_wgrad"))
        else:
            self.op.wgrad(saved[0], G, dw.view(self.mod.out_channels, 64, 1, 1, 1), db)
        grads.done(self.mod.weight, self.mod.bias)
        return None


class _ConvT1Stage:
    """Final ConvTranspose3d(128 -> 1, k4 s2 p1) (csrc/convt1.hip).  At scale the 64 taps become the channels of a 1x1x1 convolution on
    the MFMA kernels (P = x . w per cell, then a gather onto the output grid; backward: im2col of the gradient, then the 1x1x1 dgrad /
    wgrad); tiny inputs use the direct kernels."""

    GEMM_MIN_CELLS = 4096

    def __init__(self, mod: nn.ConvTranspose3d, in_act, dtype):
        self.mod, self.in_act, self.dtype = mod, in_act, dtype
        # taps as output channels: weight[c][tap] read with (cout stride, cin stride) = (1, 64)
        self.taps_fwd = ConvOp("conv", 128, 64, 1, 1, 0, mod.weight, None, dtype, w_strides=(1, 64))
        self.taps_bwd = ConvOp("conv", 64, 128, 1, 1, 0, mod.weight, None, dtype)

    @staticmethod
    def applicable(mod) -> bool:
        return (isinstance(mod, nn.ConvTranspose3d) and mod.in_channels == 128 and mod.out_channels == 1 and mod.kernel_size == (4, 4, 4)
                and mod.stride == (2, 2, 2) and mod.padding == (1, 1, 1))

    def params(self):
        return [self.mod.weight, self.mod.bias]

    def _sync(self):
        self.taps_fwd.weight = self.taps_bwd.weight = self.mod.weight

    def _gemm(self, x):
        return (x.numel() // 128 >= self.GEMM_MIN_CELLS and not debug.host("convt1_direct")) or debug.deterministic()   # (the direct kernels accumulate with atomics)

    def fwd(self, x, tape):
        N, D, H, W, C = x.shape
        out = torch.empty((N, 2 * D, 2 * H, 2 * W, 1), dtype=torch.float32, device=x.device)
        lib, st = _ffi.lib(), _ffi.stream()
        if self._gemm(x) and x.dtype == torch.bfloat16 and not debug.host("no_convt1_fused_fwd"):
            # one launch: the per-cell tap products stay in LDS (csrc/convt1.hip: convt1_fused_fwd_kernel)
            self._sync()
            wpk = self.taps_fwd.packed_fwd_operand(N, (D, H, W))      # [taps (padded to 128 rows)][128 channels] bf16
            _launch("convt1_fused_fwd_kernel", 2.0 * x.numel() * 64, nbytes=x.numel() * 2.0 + out.numel() * 4.0,   # algorithmic bytes: input once, output once
                    fn=lambda: _ffi.check(lib.sa_convt1_fused_fwd(_ffi.ptr(x), _ffi.ptr(wpk), _ffi.ptr(self.mod.bias), _ffi.ptr(out), N, D, H, W, st), "sa_convt1_fused_fwd"))
        elif self._gemm(x):
            self._sync()
            P = self.taps_fwd.fprop(x, out_dtype=torch.float32, use_bias=False)          # [N, D, H, W, 64]
            _ffi.check(lib.sa_convt1_gather(_ffi.ptr(P), _ffi.ptr(self.mod.bias), _ffi.ptr(out), N, D, H, W, st), "sa_convt1_gather")
        else:
            _ffi.check(lib.sa_convt1_fwd(_ffi.ptr(x), _ffi.dtype_id(x.dtype), _ffi.ptr(self.mod.weight), _ffi.ptr(self.mod.bias), _ffi.ptr(out), N, D, H, W, C, st),
                       "sa_convt1_fwd")
        if tape is not None:
            tape.append((x,))
        return out

    def bwd(self, G, saved, grads, wgrad_only=False):
        """wgrad_only: the weight / bias gradient alone (`BaselineVQVAE.last_layer_grad`: im2col of the gradient + ONE wgrad launch, no data gradient)."""
        (x,) = saved
        N, D, H, W, C = x.shape
        G = G.float().contiguous()
        dw, db = grads.buf(self.mod.weight), grads.buf(self.mod.bias)
        lib, st = _ffi.lib(), _ffi.stream()
        if self._gemm(x) and W >= 2 and x.dtype == torch.bfloat16 and not debug.host("no_convt1_fused_bwd") and not debug.deterministic():
            # csrc/conv1.hip: the data gradient is the FIRST layer's forward on the volume G, the weight gradient its weight gradient with x in the
            # role of the output gradient; no [cells][64] matrix in HBM
            self._sync()
            wpk = self.taps_bwd.packed_fwd_operand(N, (D, H, W))
            dx = torch.empty_like(x)
            _launch("convt1_backward(conv1_fwd_kernel+conv1_wgrad_kernel)", 4.0 * x.numel() * 64,
                    lambda: _ffi.check(lib.sa_convt1_backward(_ffi.ptr(G), _ffi.ptr(x), _ffi.ptr(wpk), 1 if self.in_act else 0, _ffi.ptr(dx), _ffi.ptr(dw), _ffi.ptr(db),
                                                              N, D, H, W, st), "sa_convt1_backward"))
            if wgrad_only:
                dx = None
        elif self._gemm(x):
            self._sync()
            Gc = torch.empty((N, D, H, W, 64), dtype=x.dtype, device=x.device)
            if debug.deterministic():
                from ...engine import colsum_det
                colsum_det(G.view(-1, 1), 1, db)     # bias gradient = sum of the gradient volume, in a fixed order
                db = None
            _ffi.check(lib.sa_convt1_im2col(_ffi.ptr(G), _ffi.dtype_id(x.dtype), _ffi.ptr(Gc), _ffi.ptr(db), N, D, H, W, st), "sa_convt1_im2col")
            dx = None
            if not wgrad_only:
                dx = self.taps_bwd.fprop(Gc, mask=x if self.in_act else None, mask_mode=MASK_POS if self.in_act else MASK_NONE, use_bias=False)
            self.taps_fwd.wgrad(x, Gc, dw, None)
        else:
            dx = torch.empty_like(x)
            _ffi.check(lib.sa_convt1_bwd(_ffi.ptr(x), _ffi.dtype_id(x.dtype), _ffi.ptr(self.mod.weight), _ffi.ptr(G), _ffi.ptr(x) if self.in_act else None,
                                         _ffi.ptr(dx), _ffi.ptr(dw), _ffi.ptr(db), N, D, H, W, C, st), "sa_convt1_bwd")
        grads.done(self.mod.weight, self.mod.bias)
        return dx


class SubpixelUpsample(nn.Module):
    """Parameter holder with the state_dict keys of ``monai.networks.blocks.SubpixelUpsample(dimensions=3, in_channels, out_channels, scale_factor,
    apply_pad_pool=True, bias=True)`` -- the last decoder layer with ``use_subpixel_conv=True`` (reference baseline.py:274-282): ``conv_block`` =
    Conv3d(in_channels -> out_channels * scale_factor^3, k 3, p 1) with MONAI's ICNR initialisation (every group of scale_factor^3 output channels starts
    from one Kaiming-normal kernel), then pixelshuffle -> ConstantPad3d((scale_factor - 1, 0) x 3) -> AvgPool3d(scale_factor, stride 1) (``pad_pool``,
    no parameters).  MONAI is absent offline: restated from its published source (oracle/vqvae_ref.subpixel_upsample)."""

    def __init__(self, dimensions: int, in_channels: int, out_channels: int, scale_factor: int = 2, apply_pad_pool: bool = True, bias: bool = True):
        super().__init__()
        if dimensions != 3 or scale_factor != 2 or not apply_pad_pool or out_channels != 1:
            raise NotImplementedError("SubpixelUpsample on MI355X: the reference's use (3-D, scale factor 2, one output channel, pad + pool)")
        self.dimensions, self.scale_factor = dimensions, scale_factor
        self.conv_block = nn.Conv3d(in_channels, out_channels * scale_factor ** 3, kernel_size=3, stride=1, padding=1, bias=bias)
        with torch.no_grad():      # monai.networks.utils.icnr_init
            oc, ic = self.conv_block.weight.shape[:2]
            sf = scale_factor ** 3
            k = nn.init.kaiming_normal_(torch.zeros(oc // sf, ic, 3, 3, 3)).transpose(0, 1)
            k = k.reshape(oc // sf, ic, -1).repeat(1, 1, sf)
            self.conv_block.weight.copy_(k.reshape(ic, oc, 3, 3, 3).transpose(0, 1))
        self.pad_pool = nn.Sequential(nn.ConstantPad3d((scale_factor - 1, 0) * 3, 0.0), nn.AvgPool3d(kernel_size=scale_factor, stride=1))


class _SubpixelStage:
    """conv_block on the implicit-GEMM kernels (eight output channels, fp32 out) + one gather launch for pixelshuffle / pad / pool (csrc/elementwise.hip:
    sa_subpixel_pool_fwd); backward: the gather's adjoint, then the conv_block's weight and data gradient launches."""

    def __init__(self, mod: SubpixelUpsample, in_act, dtype):
        conv = mod.conv_block
        self.mod, self.conv, self.in_act, self.dtype = mod, conv, in_act, dtype
        self.op = ConvOp("conv", conv.in_channels, conv.out_channels, 3, 1, 1, conv.weight, conv.bias, dtype)

    def params(self):
        return [self.conv.weight, self.conv.bias]

    def _sync(self):
        self.op.weight, self.op.bias = self.conv.weight, self.conv.bias

    def fwd(self, x, tape):
        self._sync()
        N, D, H, W, _ = x.shape
        c = self.op.fprop(x, act=ACT_NONE, out_dtype=torch.float32, out_channels_stride=8)
        out = torch.empty((N, 2 * D, 2 * H, 2 * W, 1), dtype=torch.float32, device=x.device)
        _ffi.check(_ffi.lib().sa_subpixel_pool_fwd(_ffi.ptr(c), _ffi.ptr(out), N, D, H, W, _ffi.stream()), "sa_subpixel_pool_fwd")
        if tape is not None:
            tape.append((x,))
        return out

    def bwd(self, G, saved, grads, wgrad_only=False):
        (x,) = saved
        self._sync()
        N, D, H, W, _ = x.shape
        G = G.float().contiguous()
        dc = torch.empty((N, D, H, W, 8), dtype=self.dtype, device=x.device)
        _ffi.check(_ffi.lib().sa_subpixel_pool_bwd(_ffi.ptr(G), _ffi.ptr(dc), _ffi.dtype_id(self.dtype), N, D, H, W, _ffi.stream()), "sa_subpixel_pool_bwd")
        grads.wgrad(self.op, x, dc, grads.buf(self.conv.weight), grads.buf(self.conv.bias))
        grads.done(self.conv.weight, self.conv.bias)
        if wgrad_only:
            return None
        return self.op.dgrad(dc, (D, H, W), mask=x if self.in_act else None, mask_mode=MASK_POS)


class _ResStage:
    def __init__(self, mod: ResidualLayer, in_act, dtype, fwd_dtype=None):
        c3, c1 = mod[0], mod[3]
        self.mod = mod
        self.c3m, self.c1m, self.in_act, self.dtype = c3, c1, in_act, dtype
        self.c3 = ConvOp("conv", c3.in_channels, c3.out_channels, 3, 1, 1, c3.weight, c3.bias, dtype, fwd_dtype=fwd_dtype)
        self.c1 = ConvOp("conv", c1.in_channels, c1.out_channels, 1, 1, 0, c1.weight, c1.bias, dtype, fwd_dtype=fwd_dtype)
        self.mixed = self.c3.fwd_dtype != dtype

    def params(self):
        return [self.c3m.weight, self.c3m.bias, self.c1m.weight, self.c1m.bias]

    def _sync(self):
        self.c3.weight, self.c3.bias = self.c3m.weight, self.c3m.bias
        self.c1.weight, self.c1.bias = self.c1m.weight, self.c1m.bias

    def _fused_ok(self, x):
        return (self.dtype == torch.bfloat16 and self.c3.cin == 128 and self.c3.cout == 128 and self.c1.cout == 128
                and x.numel() * 2 < 0xfffffff0 - 4096 and not debug.host("no_fused_res"))

    def _fwd_fused(self, x, need_h):
        """3x3x3 conv + ReLU + 1x1x1 conv + residual + ReLU in one launch (csrc/conv_fprop.hip, FUSE=true)."""
        import ctypes
        N, D, H, W, C = x.shape
        p3 = self.c3._get_plans(N, (D, H, W), 128, 128)
        p1 = self.c1._get_plans(N, (D, H, W), 128, 128)
        fdt = self.c3.fwd_dtype      # == self.dtype, or float16 in an f16 forward chain (x, y f16; h and the copy of y for the backward pass bf16)
        self.c3._ensure_packed(p3["fwd"], fdt)
        self.c1._ensure_packed(p1["fwd"], fdt)
        y = torch.empty_like(x)
        h = torch.empty(x.shape, dtype=self.dtype, device=x.device) if need_h else None
        ys = torch.empty(x.shape, dtype=self.dtype, device=x.device) if (need_h and self.mixed) else None
        ep = ConvOp._epilogue(self.c1._bias_padded(), x, None, None, ACT_RELU, 0, True, fdt, 0.2, None, ys)
        from ... import engine
        b1 = self.c3._bias_padded()
        st = _ffi.stream()
        engine._launch(None, engine._geom_flops(p3["fwd"][0].geom) + engine._geom_flops(p1["fwd"][0].geom),
                       lambda: _ffi.check(_ffi.lib().sa_resblock_fprop(ctypes.byref(p3["fwd"][0].geom), _ffi.dtype_id(fdt), _ffi.ptr(x),
                                                                       _ffi.ptr(p3["fwd"][0].wpk), _ffi.ptr(b1), _ffi.ptr(p1["fwd"][0].wpk), _ffi.ptr(h), _ffi.ptr(y),
                                                                       ctypes.byref(ep), st), "sa_resblock_fprop"),
                       abytes=engine._tbytes(x, y, h, ys, p3["fwd"][0].wpk, p1["fwd"][0].wpk))     # x in (operand and addend: once), both weight operands, every output
        return (_Act(y, ys) if self.mixed else y), h

    def _dropout_mask(self, N, C, dev, stride=None):
        """nn.Dropout3d (baseline.py:155): whole channels of a sample are zeroed with probability p, the others scaled by 1 / (1 - p); [N, 1, 1, 1, stride] fp32
        (``stride`` = the channel stride of the hidden tensor: padding channels beyond C carry ones)."""
        p = self.mod.p_dropout
        keep = torch.bernoulli(torch.full((N, 1, 1, 1, C), 1.0 - p, device=dev)) / (1.0 - p)
        if stride is not None and stride > C:
            keep = torch.cat([keep, torch.ones((N, 1, 1, 1, stride - C), device=dev)], dim=-1)
        return keep

    def _fwd_dropout(self, xf, xs, tape):
        """Training step with p_dropout > 0: relu(conv3(x)) -> channel mask -> conv1 + x -> relu as two launches with the mask (a broadcast multiply on the device)
        between them.  Saved for the backward pass: x, the MASKED hidden activation (the operand of the 1x1x1 weight gradient; its sign pattern is the ReLU mask
        of the surviving channels) and the mask."""
        N, C = xf.shape[0], self.c3.cout
        m = self._dropout_mask(N, C, xf.device, _ru(C, vec_of(self.dtype)))
        if self.mixed:
            hf, _, h = self.c3.fprop(xf, act=ACT_RELU, want_lp=True)
            hf = (hf * m).to(hf.dtype)
            h = (h * m).to(h.dtype)
            yf, _, ys = self.c1.fprop(hf, act=ACT_RELU, addend=xf, add_before_act=True, want_lp=True)
            y = _Act(yf, ys)
        else:
            h = self.c3.fprop(xf, act=ACT_RELU)
            h = (h * m).to(h.dtype)
            y = self.c1.fprop(h, act=ACT_RELU, addend=xf, add_before_act=True)
        if tape is not None:
            tape.append((xs, h, m))
        return y

    def fwd(self, x, tape):
        self._sync()
        xf, xs = _fs(x)
        rec = tape is not None
        if self.mod.p_dropout > 0.0 and self.mod.training:      # nn.Dropout3d is active whenever the module trains, recorded or not (no_grad forward in train())
            return self._fwd_dropout(xf, xs, tape)
        if self._fused_ok(xf):
            y, h = self._fwd_fused(xf, rec)
        elif self.mixed:     # two launches; each writes its f16 output and, when recording, the bf16 copy the backward pass reads
            if rec:
                hf, _, h = self.c3.fprop(xf, act=ACT_RELU, want_lp=True)
                yf, _, ys = self.c1.fprop(hf, act=ACT_RELU, addend=xf, add_before_act=True, want_lp=True)
            else:
                hf, h, ys = self.c3.fprop(xf, act=ACT_RELU), None, None
                yf = self.c1.fprop(hf, act=ACT_RELU, addend=xf, add_before_act=True)
            y = _Act(yf, ys)
        else:
            h = self.c3.fprop(xf, act=ACT_RELU)
            y = self.c1.fprop(h, act=ACT_RELU, addend=xf, add_before_act=True)
        if rec:
            tape.append((xs, h))
        return y

    def bwd(self, G, saved, grads):
        x, h = saved[0], saved[1]
        drop = saved[2] if len(saved) > 2 else None      # Dropout3d mask of this step: the gradient wrt the hidden activation is scaled by it
        self._sync()
        dims = tuple(x.shape[1:4])
        from ...engine import conv1x1_backward
        dp = conv1x1_backward(self.c1, h, G, grads.buf(self.c1m.weight), grads.buf(self.c1m.bias))   # dw, db and the masked dgrad in one launch
        if dp is None:
            self.c1.wgrad(h, G, grads.buf(self.c1m.weight), grads.buf(self.c1m.bias))
            dp = self.c1.dgrad(G, dims, mask=h, mask_mode=MASK_POS)
        grads.done(self.c1m.weight, self.c1m.bias)
        if drop is not None:
            dp = (dp * drop).to(dp.dtype)
        grads.wgrad(self.c3, x, dp, grads.buf(self.c3m.weight), grads.buf(self.c3m.bias))
        grads.done(self.c3m.weight, self.c3m.bias)
        return self.c3.dgrad(dp, dims, addend=G, mask=x if self.in_act else None, mask_mode=MASK_POS)


class _GradCtx:
    """Where the weight-gradient kernels accumulate.  Without a sink: fresh zeroed tensors handed back to autograd.  With a
    sink (runtime.ddp.GradReducer): views of the flat gradient buffer, and the sink is told as soon as a parameter's
    gradient kernels are queued so that its bucket's all-reduce can start while backward continues."""

    def __init__(self, sink=None, side=None):
        self.sink, self.grads, self.side = sink, {}, side

    def wgrad(self, op, x, g, dw, db):
        """op.wgrad(x, g, dw, db); on the second stream when there is one (networks/transformers/performer._SideWgrad: weight gradients are leaves of the pass)"""
        if self.side is None:
            op.wgrad(x, g, dw, db)
        else:
            self.side.run(lambda: op.wgrad(x, g, dw, db), x, g, dw, db)

    def buf(self, p):
        if self.sink is not None:
            b = self.sink.buffer(p)
            if b is not None:
                return b
        t = torch.zeros_like(p)
        self.grads[p] = t
        return t

    def done(self, *params):
        if self.sink is None:
            return
        if self.side is not None and getattr(self.sink, "active", True):
            # DDP: the bucket's all-reduce waits for an event of the CURRENT stream; these gradients were queued on the main and on the weight-gradient stream
            self.side.side.wait_stream(self.side.main)
            with torch.cuda.stream(self.side.side):
                self._report(params)
            return
        self._report(params)

    def _report(self, params):
        if hasattr(self.sink, "flush"):
            tok = self.sink.flush()       # optimizer slices deferred by EARLIER reports: everything that reads their parameters is queued (runtime/ddp.GradReducer.flush)
            for p in params:
                self.sink.ready(p, tok)
        else:
            for p in params:
                self.sink.ready(p)


class _Chain:
    def __init__(self, stages, dtype, in_channels, fwd_dtype=None):
        self.stages, self.dtype, self.in_channels = stages, dtype, in_channels
        self.fwd_dtype = fwd_dtype or dtype     # float16 with dtype = bfloat16: the forward launches run on f16 operands (_Act)
        self.grad_sink = None
        self.last_tape = None

    def params(self) -> List[nn.Parameter]:
        return [p for s in self.stages for p in s.params()]

    def ops(self):
        return [op for s in self.stages
                for op in (getattr(s, "op", None), getattr(s, "c3", None), getattr(s, "c1", None), getattr(s, "taps_fwd", None), getattr(s, "taps_bwd", None),
                           getattr(s, "fallback_op", None))
                if op is not None]

    def invalidate(self):
        for op in self.ops():
            op.invalidate()

    def forward(self, x_ncdhw: torch.Tensor, record: bool):
        """x [B,C,D,H,W] fp32 (any strides) -> y channels-last [B,D,H,W,C'] (+ tape)."""
        _ffi.require_gpu()
        vec = vec_of(self.dtype)
        if isinstance(self.stages[0], _Conv1Stage) and x_ncdhw.shape[1] == 1:
            x = x_ncdhw.float().contiguous().view(x_ncdhw.shape[0], *x_ncdhw.shape[2:])   # the first stage reads the fp32 volume itself
        else:
            x = x_ncdhw.float().permute(0, 2, 3, 4, 1).contiguous()
            cs = (self.in_channels + vec - 1) // vec * vec
            if self.fwd_dtype != self.dtype:
                x = _Act(cast_pad(x, self.fwd_dtype, cs), cast_pad(x, self.dtype, cs) if record else None)
            else:
                x = cast_pad(x, self.dtype, cs)
        tape = [] if record else None
        for s in self.stages:
            x = s.fwd(x, tape)
        if isinstance(x, _Act):
            x = x.f
        self.last_tape = tape     # for last_stage_wgrad (adaptive adversarial weight); dropped when the backward pass consumes the tape
        return x, tape

    def last_stage_wgrad(self, G: torch.Tensor) -> torch.Tensor:
        """Weight gradient of the LAST stage alone for an output gradient G [B, D, H, W, C] (channels-last), from the input the last recorded
        forward saved.  Does not touch the gradient sink."""
        if self.last_tape is None:
            raise RuntimeError("last_stage_wgrad needs a recorded forward whose backward has not run yet")
        st = self.stages[-1]
        gc = _GradCtx(None)
        if isinstance(st, (_ConvStage, _ConvT1Stage, _SubpixelStage)):
            st.bwd(G, self.last_tape[-1], gc, wgrad_only=True)   # no data gradient: the caller only wants d loss / d W_last
        else:
            st.bwd(G, self.last_tape[-1], gc)
        return gc.grads[st.params()[0]]

    def backward(self, G: torch.Tensor, tape):
        side = None
        # Weight gradients are leaves of the backward pass: they run on a second HIP stream beside the data-gradient chain (round 5: default, +0.9 % on the
        # step -- 113.5 -> 112.5 ms in alternating same-box runs; overlapping launches inflate every per-kernel duration, which is why bench.py's roofline record
        # comes from a separate pass with SA_NO_SIDE_WGRAD semantics).  SA_NO_SIDE_WGRAD=1 / --deterministic: one stream.
        if (self.dtype != torch.float32 and not debug.host("no_side_wgrad") and not debug.host("no_side_wgrad_vqvae")
                and not debug.deterministic() and G.is_cuda):
            from ..transformers.performer import _SideWgrad
            side = _SideWgrad(G.device)
        gc = _GradCtx(self.grad_sink, side)
        for s, saved in zip(reversed(self.stages), reversed(tape)):
            G = s.bwd(G, saved, gc)
        if side is not None:
            side.join()
        return G, gc.grads


class _ChainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, chain: _Chain, record: bool, x, *params):
        y, tape = chain.forward(x, record=record)
        ctx.chain, ctx.tape, ctx.nparams = chain, tape, len(params)
        ctx.x_needs = x.requires_grad
        return y.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, gy):
        chain: _Chain = ctx.chain
        G = gy.permute(0, 2, 3, 4, 1).contiguous()
        Gin, grads = chain.backward(G, ctx.tape)
        if chain.last_tape is ctx.tape:
            chain.last_tape = None
        ctx.tape = None
        gx = None
        if ctx.x_needs and Gin is not None:
            gx = Gin[..., : chain.in_channels].float().permute(0, 4, 1, 2, 3)
        return (None, None, gx, *[grads.get(p) for p in chain.params()])


# ------------------------------------------------------------------------------------------------ the network
class BaselineVQVAE(VQVAEBase, nn.Module):
    def __init__(
        self,
        n_levels: int = 3,
        downsample_parameters: Tuple[Tuple[int, int, int, int], ...] = ((4, 2, 1, 1), (4, 2, 1, 1), (4, 2, 1, 1)),
        upsample_parameters: Tuple[Tuple[int, int, int, int, int], ...] = ((4, 2, 1, 0, 1), (4, 2, 1, 0, 1), (4, 2, 1, 0, 1)),
        n_embed: int = 256,
        embed_dim: int = 256,
        n_channels: int = 144,
        n_res_channels: int = 144,
        n_res_layers: int = 3,
        p_dropout: float = 0.0,
        commitment_cost: float = 0.25,
        vq_decay: float = 0.5,
        use_subpixel_conv: bool = False,
        compute_dtype: torch.dtype = torch.bfloat16,
        encoder_forward_dtype: Optional[torch.dtype] = None,
    ):
        super().__init__()
        assert n_levels == len(downsample_parameters) and n_levels == len(upsample_parameters), (
            f"downsample_parameters, upsample_parameters must have the same number of elements as n_levels. "
            f"But got {len(downsample_parameters)} and {len(upsample_parameters)}, instead of {n_levels}."
        )
        if use_subpixel_conv and (n_levels < 2 or n_channels // 2 % 8):
            # (baseline.py:274-282 builds SubpixelUpsample(in_channels=n_channels // 2): with one level the residual stack in front of it is n_channels wide and
            #  the reference itself fails in the first forward)
            raise NotImplementedError("use_subpixel_conv=True needs n_levels >= 2 and n_channels // 2 a multiple of 8")
        for dp in downsample_parameters:
            if dp[3] != 1:
                raise NotImplementedError("dilation != 1")
        for up in upsample_parameters:
            if tuple(up) != (4, 2, 1, 0, 1):
                raise NotImplementedError("upsample_parameters other than (4, 2, 1, 0, 1)")
        if n_res_channels != n_channels:
            raise NotImplementedError("n_res_channels != n_channels (configure.py:27-28 always passes them equal)")
        self.n_levels, self.downsample_parameters, self.upsample_parameters = n_levels, downsample_parameters, upsample_parameters
        self.n_embed, self.embed_dim, self.use_subpixel_conv = n_embed, embed_dim, use_subpixel_conv
        self.n_channels, self.n_res_channels, self.n_res_layers, self.p_dropout = n_channels, n_res_channels, n_res_layers, p_dropout
        self.commitment_cost, self.vq_decay = commitment_cost, vq_decay
        self.compute_dtype = compute_dtype
        # Throughput mode (bf16 MFMA): the ENCODER's forward launches run on float16 activations and weights -- the reference's AMP dtype
        # (src/engines/trainer.py:161-163; three more mantissa bits than bf16 at the same MFMA rate), so the code indices follow the fp32 path; every
        # gradient, the decoder and the saved activations stay bf16.  SA_NO_F16_FORWARD=1 / encoder_forward_dtype=torch.bfloat16: bf16 forward too.
        if encoder_forward_dtype is None:
            encoder_forward_dtype = torch.float16 if (compute_dtype == torch.bfloat16 and not debug.host("no_f16_forward")) else compute_dtype
        if encoder_forward_dtype != compute_dtype and not (encoder_forward_dtype == torch.float16 and compute_dtype == torch.bfloat16):
            raise ValueError("encoder_forward_dtype: torch.float16 with compute_dtype=torch.bfloat16, or the compute dtype itself")
        self.encoder_forward_dtype = encoder_forward_dtype
        if (n_channels // 2) % 8 or embed_dim % 8:
            raise NotImplementedError("channel counts must be multiples of 8 (16-byte channels-last vectors): n_channels//2 and embed_dim")

        self.encoder = self.construct_encoder()
        self.quantizer = self.construct_quantizer()
        self.decoder = self.construct_decoder()
        self._enc_chain = self._build_encoder_chain()
        self._dec_chain = self._build_decoder_chain()

    # ---------------------------------------------------------------- parameter trees (same numbering as the reference)
    def _level_width(self, level: int, decoder: bool) -> int:
        full = (level == 0) if decoder else (level == self.n_levels - 1)
        return self.n_channels if full else self.n_channels // 2

    def _res_stack(self, width: int) -> nn.Sequential:
        return nn.Sequential(*[ResidualLayer(width, width, self.p_dropout) for _ in range(self.n_res_layers)])

    def construct_encoder(self) -> nn.ModuleList:
        seq: List[nn.Module] = []
        cin = 1
        for lvl, (k, s, p, dil) in enumerate(self.downsample_parameters):
            width = self._level_width(lvl, decoder=False)
            seq += [nn.Conv3d(cin, width, kernel_size=k, stride=s, padding=p, dilation=dil), nn.ReLU(), self._res_stack(width)]
            cin = width
        seq.append(nn.Conv3d(self.n_channels, self.embed_dim, 3, stride=1, padding=1))
        return nn.ModuleList([nn.Sequential(*seq)])

    def construct_quantizer(self) -> nn.ModuleList:
        return nn.ModuleList([Quantizer(self.n_embed, self.embed_dim, commitment_cost=self.commitment_cost, decay=self.vq_decay)])

    def construct_decoder(self) -> nn.ModuleList:
        seq: List[nn.Module] = [nn.Conv3d(self.embed_dim, self.n_channels, 3, stride=1, padding=1)]
        for lvl, (k, s, p, op, dil) in enumerate(self.upsample_parameters):
            width = self._level_width(lvl, decoder=True)
            last = lvl == self.n_levels - 1
            seq.append(self._res_stack(width))
            if last and self.use_subpixel_conv:
                seq.append(SubpixelUpsample(dimensions=3, in_channels=self.n_channels // 2, out_channels=1, scale_factor=s, apply_pad_pool=True, bias=True))
            else:
                seq.append(nn.ConvTranspose3d(width, 1 if last else self.n_channels // 2, kernel_size=k, stride=s, padding=p, output_padding=op, dilation=dil))
            if not last:
                seq.append(nn.ReLU())
        return nn.ModuleList([nn.Sequential(*seq)])

    # ---------------------------------------------------------------- launch chains over those parameters
    def _build_encoder_chain(self, fdt: Optional[torch.dtype] = None) -> _Chain:
        dt, fdt = self.compute_dtype, fdt or self.encoder_forward_dtype
        mods = list(self.encoder[0])
        stages = []
        for lvl in range(self.n_levels):
            conv, res = mods[3 * lvl], mods[3 * lvl + 2]
            if lvl == 0 and _Conv1Stage.applicable(conv):
                stages.append(_Conv1Stage(conv, ACT_RELU, dt, fwd_dtype=fdt))
            else:
                stages.append(_ConvStage(conv, "conv", ACT_RELU, in_act=lvl > 0, dtype=dt, need_dx=lvl > 0, fwd_dtype=fdt))
            stages += [_ResStage(r, in_act=True, dtype=dt, fwd_dtype=fdt) for r in res]
        stages.append(_ConvStage(mods[3 * self.n_levels], "conv", ACT_NONE, in_act=True, dtype=dt, out_f32=True, fwd_dtype=fdt))
        return _Chain(stages, dt, in_channels=1, fwd_dtype=fdt)

    def _build_decoder_chain(self) -> _Chain:
        dt = self.compute_dtype
        mods = list(self.decoder[0])
        stages = [_ConvStage(mods[0], "conv", ACT_NONE, in_act=False, dtype=dt)]
        i = 1
        for lvl in range(self.n_levels):
            last = lvl == self.n_levels - 1
            res, up = mods[i], mods[i + 1]
            for j, r in enumerate(res):
                stages.append(_ResStage(r, in_act=not (lvl == 0 and j == 0), dtype=dt))
            if isinstance(up, SubpixelUpsample):
                stages.append(_SubpixelStage(up, in_act=True, dtype=dt))
            elif last and _ConvT1Stage.applicable(up):
                stages.append(_ConvT1Stage(up, in_act=True, dtype=dt))
            else:
                stages.append(_ConvStage(up, "convT", ACT_NONE if last else ACT_RELU, in_act=True, dtype=dt, out_f32=last))
            i += 2 if last else 3
        return _Chain(stages, dt, in_channels=self.embed_dim)

    def _chains(self) -> List[_Chain]:
        return [c for c in (self._enc_chain, getattr(self, "_enc_chain_lp", None), self._dec_chain) if c is not None]

    def _encoder_chain_for(self, images: torch.Tensor) -> _Chain:
        """The f16 forward instances exist in LDS-DMA form only (32-bit buffer offsets): when the largest encoder activation reaches 4 GiB (first-level
        output, 80x112x80x128 at batch ~24) or SA_NO_DMA is set, the encoder runs the all-bf16 chain (the round-3 mode, which has register-staged
        kernels for such operands) over the same parameters instead of failing with SA_EUNSUPPORTED."""
        if self.encoder_forward_dtype == self.compute_dtype:
            return self._enc_chain
        k, s_, p_, _ = self.downsample_parameters[0]
        vox = images.shape[0]
        for d in images.shape[2:]:
            vox *= (d + 2 * p_ - k) // s_ + 1
        too_big = vox * self._level_width(0, decoder=False) * 2 >= 0xfffffff0 - 4096
        if not (too_big or (_ffi.lib().sa_get_debug_flags() & debug.LIB_FLAGS["no_dma"])):
            return self._enc_chain
        if getattr(self, "_enc_chain_lp", None) is None:
            self._enc_chain_lp = self._build_encoder_chain(self.compute_dtype)
            self._enc_chain_lp.grad_sink = self._enc_chain.grad_sink
        return self._enc_chain_lp

    def set_grad_sink(self, sink):
        """Route weight gradients into a ``runtime.ddp.GradReducer`` (flat buffer + overlapped RCCL all-reduce)."""
        for c in self._chains():
            c.grad_sink = sink

    def invalidate_packed_weights(self):
        """Tell the launch chains that parameters were modified through raw pointers (fused Adam kernel)."""
        for c in self._chains():
            c.invalidate()
        if getattr(self, "_packset", None) is None:
            self._packset = PackSet()
        self._packset.repack([op for c in self._chains() for op in c.ops()])   # all packed operands again, in one launch

    def range_repacker(self, flat):
        """For ``FusedAdam(in_backward=reducer)`` (see networks/transformers/performer.Performer.range_repacker)."""
        from ...engine import RangeRepacker
        return RangeRepacker(flat, lambda: [op for c in self._chains() for op in c.ops()])

    # ---------------------------------------------------------------- accessors (baseline.py:301-327)
    def get_ema_decay(self) -> Sequence[float]:
        return [self.quantizer[0].get_ema_decay()]

    def set_ema_decay(self, decay: Union[Sequence[float], float]) -> Sequence[float]:
        self.quantizer[0].set_ema_decay(decay[0] if isinstance(decay, list) else decay)
        return self.get_ema_decay()

    def get_commitment_cost(self) -> Sequence[float]:
        return [self.quantizer[0].get_commitment_cost()]

    def set_commitment_cost(self, commitment_factor: Union[Sequence[float], float]) -> Sequence[float]:
        self.quantizer[0].set_commitment_cost(commitment_factor[0] if isinstance(commitment_factor, list) else commitment_factor)
        return self.get_commitment_cost()

    def get_perplexity(self) -> Sequence[float]:
        return [self.quantizer[0].get_perplexity()]

    def get_last_layer(self) -> nn.parameter.Parameter:
        if self.use_subpixel_conv:     # (baseline.py:322 would pick the parameter-free AvgPool3d of pad_pool and raise: the conv_block is the last layer with weights)
            return list(self.decoder[0])[-1].conv_block.weight
        return list(self.decoder.modules())[-1].weight

    def last_layer_grad(self, d_recon: torch.Tensor) -> torch.Tensor:
        """``torch.autograd.grad(loss, self.get_last_layer())`` of the reference's adaptive adversarial weight (src/engines/trainer.py:278-285)
        for a loss that reaches the last layer through the reconstruction only, given ``d_recon = d loss / d reconstruction`` [B,1,D,H,W]:
        the last decoder stage's weight-gradient launch on the input saved by the last recorded forward -- not a second decoder backward."""
        return self._dec_chain.last_stage_wgrad(d_recon.permute(0, 2, 3, 4, 1).contiguous())

    # ---------------------------------------------------------------- hot path (baseline.py:329-362)
    @staticmethod
    def _run(chain: _Chain, x: torch.Tensor) -> torch.Tensor:
        params = chain.params()
        record = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _ChainFn.apply(chain, record, x, *params)

    def encode(self, images: torch.Tensor) -> List[torch.Tensor]:
        return [self._run(self._encoder_chain_for(images), images)]

    def quantize(self, encodings: List[torch.Tensor]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        x, x_loss = self.quantizer[0](encodings[0])
        return [x], [x_loss]

    def decode(self, quantizations: List[torch.Tensor]) -> torch.Tensor:
        return self._run(self._dec_chain, quantizations[0])

    def index_quantize(self, images: torch.Tensor) -> List[torch.Tensor]:
        encodings = self.encode(images)
        _, _, encoding_indices = self.quantizer[0].quantize(encodings[0])
        return [encoding_indices]

    def decode_samples(self, embedding_indices: List[torch.Tensor]) -> torch.Tensor:
        samples_codes = self.quantizer[0].embed(embedding_indices[0])
        return self.decode([samples_codes])

    def forward(self, images: torch.Tensor) -> Dict[str, List[torch.Tensor]]:
        encodings = self.encode(images)
        quantizations, quantization_losses = self.quantize(encodings)
        reconstruction = self.decode(quantizations)
        return {"reconstruction": [reconstruction], "quantization_losses": quantization_losses}

    def state_dict(self, *args, **kwargs):
        if torch.cuda.is_available():
            self.quantizer[0].impl.wait_ema()
        return super().state_dict(*args, **kwargs)
