"""``performer`` on MI355X: the reference's ``Performer`` plugin surface over hand-written HIP kernels.

Mirrors reference ``src/networks/transformers/performer.py`` (``AbsoluteSpatialPositionalEmbedding`` :23-40, ``Performer``
:70-288): same keyword-only constructor, ``forward(x, conditionings=None, return_encodings=False)``,
``check_redraw_projections`` / ``fix_projection_matrices_``, ``.ordering`` and ``TransformerBase.sample``.  The layer
stack the reference delegates to ``performer_pytorch.Performer`` (1.0.11, third-party) is rebuilt here with the same
module / parameter names (``performer.net.layers.{i}.0.fn.to_q.weight`` ...), so state_dicts are interchangeable.

Underneath, autograd sees FOUR nodes -- embeddings, the whole ReZero / pre-LayerNorm layer stack, the final LayerNorm and
the vocabulary projection -- each a hand-scheduled sequence of HIP launches: 1-tap implicit-GEMM MFMA launches for
every Linear (``compute_dtype`` fp32 = exact-f32 MFMA as the reference's amp=False, or bf16), fp32 FAVOR+ feature maps
and running-state scans, fp32 rotary + banded local attention.  See oracle/performer_ref.py for the restated spec.
"""
from __future__ import annotations

import collections
import ctypes
import math
import os
from enum import Enum
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from ... import _ffi, debug
from ..._ffi import MASK_GELU
from ... import engine as _engine
from ...engine import ConvOp, PackSet
from .img2seq_ordering import Ordering
from .transformer import TransformerBase, sequence_to_grid


class TransformerConditioningType(Enum):  # src/utils/transformer.py:21-24
    NONE = "none"
    BOSREPLACEMENT = "bos_replacement"
    PREPENDING = "prepending"


def _ru(x, m):
    return (x + m - 1) // m * m


def _ck(rc, what):
    _ffi.check(rc, what)


# ------------------------------------------------------------------------------------------------ parameter holders
class AbsolutePositionalEmbedding(nn.Module):  # performer_pytorch.AbsolutePositionalEmbedding
    def __init__(self, dim, max_seq_len):
        super().__init__()
        self.emb = nn.Embedding(max_seq_len, dim)


class FixedPositionalEmbedding(nn.Module):
    """performer_pytorch.FixedPositionalEmbedding (1.0.11): rows sin | cos of position x inverse frequency, a buffer named ``emb`` (no parameters);
    `fixed_position_emb=True` of the wrapper (performer.py:138-140)."""

    def __init__(self, dim, max_seq_len):
        super().__init__()
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        position = torch.arange(0, max_seq_len, dtype=torch.float)
        sinusoid_inp = torch.einsum("i,j->ij", position, inv_freq)
        self.register_buffer("emb", torch.cat((sinusoid_inp.sin(), sinusoid_inp.cos()), dim=-1).contiguous())


class AxialPositionalEmbedding(nn.Module):
    """axial_positional_embedding.AxialPositionalEmbedding(dim, axial_shape) in its summed form (the wrapper's `axial_position_emb=True`, performer.py:141-145; the
    package is un-pinned and absent offline -- restated): parameters ``weights_0`` [1, s0, 1, dim] and ``weights_1`` [1, 1, s1, dim], N(0, 1); position t of the
    flattened (s0, s1) grid receives weights_0[t // s1] + weights_1[t % s1].  The embedding-sum kernel gathers the two tables with those two index rows."""

    def __init__(self, dim, axial_shape):
        super().__init__()
        assert len(axial_shape) == 2, "axial_position_shape must have two axes (performer.py:142-144: (ceil(max_seq_len / 64), 64))"
        self.dim, self.shape = dim, tuple(int(a) for a in axial_shape)
        self.max_seq_len = self.shape[0] * self.shape[1]
        self.weights_0 = nn.Parameter(torch.zeros(1, self.shape[0], 1, dim).normal_(0, 1))
        self.weights_1 = nn.Parameter(torch.zeros(1, 1, self.shape[1], dim).normal_(0, 1))


class AbsoluteSpatialPositionalEmbedding(nn.Module):  # performer.py:23-40
    def __init__(self, dim: int, spatial_indices_sequence: torch.Tensor):
        super().__init__()
        self.register_buffer("spatial_indices_sequence", spatial_indices_sequence)
        self.spatial_indices_sequence = self.spatial_indices_sequence[:-1]  # the last element is the predicted one
        self.emb = nn.Embedding(len(self.spatial_indices_sequence), dim)


class FixedSpatialPositionalEmbedding(nn.Module):
    """performer.py:43-66: sinusoid of the coordinate value, rows in sequence order, no parameters.  (Upstream's einsum("d,j->ij") at :51 is a
    subscript typo that raises; the outer product "i,j->ij" is the evident intent.)"""

    def __init__(self, dim, spatial_indices_sequence):
        super().__init__()
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        position = torch.arange(0, int(torch.max(spatial_indices_sequence)) + 1, dtype=torch.float)
        sinusoid_inp = torch.einsum("i,j->ij", position, inv_freq)[spatial_indices_sequence.long(), :]
        self.register_buffer("emb", torch.cat((sinusoid_inp.sin(), sinusoid_inp.cos()), dim=-1)[:-1].contiguous())


class _SinusoidalEmbeddings(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim)))


class LocalAttention(nn.Module):
    def __init__(self, window_size, dim_head):
        super().__init__()
        self.window_size = window_size
        self.rel_pos = _SinusoidalEmbeddings(dim_head)


class FastAttention(nn.Module):
    def __init__(self, dim_heads, nb_features=None):
        super().__init__()
        self.dim_heads = dim_heads
        self.nb_features = nb_features if nb_features is not None else int(dim_heads * math.log(dim_heads))
        self.register_buffer("projection_matrix", torch.zeros(self.nb_features, dim_heads))
        self.redraw_projection_matrix(None, None)  # global RNG at construction, like performer_pytorch

    @torch.no_grad()
    def redraw_projection_matrix(self, device, generator):
        """gaussian_orthogonal_random_matrix(nb_rows, nb_cols, scaling=0): orthonormalised Gaussian blocks with rows
        rescaled by chi-distributed norms.  On a HIP device the Gram-Schmidt runs in csrc (sa_favor_projection); every
        rank passes the same generator seed, so the matrices agree without the reference's DDP buffer broadcast."""
        m, d = self.nb_features, self.dim_heads
        nblk = (m + d - 1) // d
        pm = self.projection_matrix
        if pm.is_cuda:
            blocks = torch.randn(nblk, d, d, generator=generator, device=pm.device)
            rows = torch.randn(m, d, generator=generator, device=pm.device)
            _ck(_ffi.lib().sa_favor_projection(_ffi.ptr(blocks), _ffi.ptr(rows), _ffi.ptr(pm), 1, nblk, m, d, _ffi.stream()), "sa_favor_projection")
        else:  # construction time on the host (before .to(device)); plain torch, not a compute fallback of the hot path
            blocks = torch.randn(nblk, d, d, generator=generator)
            q = torch.linalg.qr(blocks.transpose(1, 2))[0].transpose(1, 2).reshape(nblk * d, d)[:m]
            pm.copy_(torch.randn(m, d, generator=generator).norm(dim=1)[:, None] * q)
        pm._sa_epoch = getattr(pm, "_sa_epoch", 0) + 1


class SelfAttention(nn.Module):
    def __init__(self, dim, heads, dim_head, local_heads, local_window_size, nb_features, qkv_bias, attn_out_bias):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.global_heads, self.dim_head = heads, heads - local_heads, dim_head
        self.fast_attention = FastAttention(dim_head, nb_features)
        self.local_attn = LocalAttention(local_window_size, dim_head) if local_heads > 0 else None
        self.to_q = nn.Linear(dim, inner, bias=qkv_bias)
        self.to_k = nn.Linear(dim, inner, bias=qkv_bias)
        self.to_v = nn.Linear(dim, inner, bias=qkv_bias)
        self.to_out = nn.Linear(inner, dim, bias=attn_out_bias)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.w1 = nn.Linear(dim, dim * mult)
        self.w2 = nn.Linear(dim * mult, dim)


class Chunk(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class ReZero(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.g = nn.Parameter(torch.tensor(1e-3))
        self.fn = fn


class PreLayerNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class SequentialSequence(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = layers


class ProjectionUpdater(nn.Module):
    def __init__(self, instance, feature_redraw_interval):
        super().__init__()
        self.instance = [instance]  # not registered (avoids a parameter cycle)
        self.feature_redraw_interval = feature_redraw_interval
        self.register_buffer("calls_since_last_redraw", torch.tensor(0))
        self._calls = 0
        self._redraws = 0
        self.base_seed = torch.initial_seed() % (2 ** 31)

    def fix_projections_(self):
        self.feature_redraw_interval = None

    def redraw_projections(self):
        if not self.training:
            return
        if self.feature_redraw_interval is not None and self._calls >= self.feature_redraw_interval:
            self._redraws += 1
            mods = [m for m in self.instance[0].modules() if isinstance(m, FastAttention)]
            dev = mods[0].projection_matrix.device
            gen = torch.Generator(device=dev).manual_seed((self.base_seed + 7919 * self._redraws) % (2 ** 31))
            same = all(m.nb_features == mods[0].nb_features and m.dim_heads == mods[0].dim_heads for m in mods)
            if dev.type == "cuda" and same:
                # all layers in ONE launch (one wave per 64x64 block); every rank uses the same seed -> identical matrices
                nl, m_, d_ = len(mods), mods[0].nb_features, mods[0].dim_heads
                nblk = (m_ + d_ - 1) // d_
                blocks = torch.randn(nl * nblk, d_, d_, generator=gen, device=dev)
                rows = torch.randn(nl * m_, d_, generator=gen, device=dev)
                out = torch.empty(nl, m_, d_, device=dev)
                _ck(_ffi.lib().sa_favor_projection(_ffi.ptr(blocks), _ffi.ptr(rows), _ffi.ptr(out), nl, nblk, m_, d_, _ffi.stream()), "sa_favor_projection")
                for i, mod in enumerate(mods):
                    mod.projection_matrix.copy_(out[i])
                    mod.projection_matrix._sa_epoch = getattr(mod.projection_matrix, "_sa_epoch", 0) + 1
            else:
                for mod in mods:
                    mod.redraw_projection_matrix(dev, gen)
            self._calls = 0
            self.calls_since_last_redraw.zero_()
            return
        self._calls += 1
        self.calls_since_last_redraw += 1


class BasePerformer(nn.Module):
    """Parameter tree of performer_pytorch.Performer (the argument order of performer.py:194-219)."""

    def __init__(self, dim, depth, heads, dim_head, local_attn_heads, local_window_size, causal, ff_mult, nb_features, feature_redraw_interval,
                 reversible, ff_chunks, generalized_attention, kernel_fn, use_scalenorm, use_rezero, ff_glu, ff_dropout, attn_dropout, cross_attend,
                 no_projection, auto_check_redraw, qkv_bias, attn_out_bias):
        super().__init__()
        unsupported = dict(reversible=reversible, generalized_attention=generalized_attention, use_scalenorm=use_scalenorm, ff_glu=ff_glu,
                           cross_attend=cross_attend, no_projection=no_projection)
        bad = [k for k, v in unsupported.items() if v]
        if bad or not causal or ff_dropout or attn_dropout or ff_chunks != 1:
            raise NotImplementedError(f"performer on MI355X implements the causal ReZero / pre-LayerNorm FAVOR+ configuration; unsupported: {bad}")
        if isinstance(local_attn_heads, int):
            local_attn_heads = (local_attn_heads,)
        local_attn_heads = tuple(local_attn_heads) * depth if len(local_attn_heads) == 1 else tuple(local_attn_heads)
        assert len(local_attn_heads) == depth and all(0 <= n <= heads for n in local_attn_heads)
        wrap = (lambda fn: ReZero(fn)) if use_rezero else (lambda fn: PreLayerNorm(dim, fn))
        layers = nn.ModuleList([])
        for lh in local_attn_heads:
            layers.append(nn.ModuleList([
                wrap(SelfAttention(dim, heads, dim_head, lh, local_window_size, nb_features, qkv_bias, attn_out_bias)),
                wrap(Chunk(FeedForward(dim, ff_mult))),
            ]))
        self.net = SequentialSequence(layers)
        self.auto_check_redraw = auto_check_redraw
        self.use_rezero = use_rezero
        self.proj_updater = ProjectionUpdater(self.net, feature_redraw_interval)

    def fix_projection_matrices_(self):
        self.proj_updater.fix_projections_()

    def check_redraw_projections(self):
        self.proj_updater.redraw_projections()


# ------------------------------------------------------------------------------------------------ launch helpers
def _lin(mod: nn.Linear, dtype) -> ConvOp:
    return ConvOp("conv", mod.in_features, mod.out_features, 1, 1, 0, mod.weight, mod.bias, dtype)


def _as5(t):  # [R, C] -> [1,1,1,R,C]
    return t.view(1, 1, 1, *t.shape)


def _cast(t, dtype):
    if t.dtype == dtype:
        return t
    from ...engine import cast_pad
    return cast_pad(t, dtype, t.shape[-1])


_SIDE_STREAMS = {}


def FAVOR_FWD_FLOP_PER_HEAD_ROW(m: int, dh: int) -> float:
    """ALGORITHMIC forward FLOPs of causal FAVOR+ per (batch, position, head): two feature maps 2 x 2 dh m, the state update k' (x) v and
    the read-out q' . S 2 x 2 m dh, the normaliser 2 x 2 m.  (The kernels execute about 3 x that on the matrix cores: split-bf16 products, and the key
    features are rebuilt by the state AND the output launch.)"""
    return 2.0 * 2 * dh * m + 2.0 * 2 * m * dh + 2.0 * 2 * m


def _favor_bracket_begin():
    """bench.py's live kernel timer (engine.TIMER): HIP events around the FAVOR+ launch groups of a layer; None when no timer is attached"""
    if _engine.TIMER is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _favor_bracket_end(e0, key, flops):
    if e0 is None or _engine.TIMER is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    _engine.TIMER.pending.append((key, flops, e0, e1, 0.0))


class _SideWgrad:
    """Weight gradients on a second HIP stream.  They are leaves of the backward pass: nothing on the main chain reads a dW before the optimizer, and the dense
    data-gradient launches they sit between run one block per CU (§4.4 (iii)) -- the weight-gradient blocks of the other queue share those CUs.  Every launch is
    ordered behind an event of the main stream (its operands are complete) and the main stream joins once at the end of the backward pass.

    Operand lifetime (round 6): the operands of the last ``LAG`` side launches are kept alive HERE, and before the launch that pushes one out of that window
    the main stream waits for its completion event -- so an operand returns to the main stream's allocator pool only behind a stream-ordered wait, and the next
    main-stream kernel that is handed the same block cannot overtake the side launch still reading it.  Rounds 3-5 used ``Tensor.record_stream`` instead: the
    caching allocator then parks every freed operand until an event recorded AT FREE TIME on the side stream has completed, and with the host a whole iteration
    ahead of the device no block ever became reusable -- the pool of the adversarial iteration grew to 113 GB for 29 GB of live tensors, with GB-sized
    ``hipMalloc``s inside the first dozen iterations (the 1 431 ms "adversarial" record of round 5: one warm-up + three timed iterations, all of them allocating)."""

    try:
        LAG = max(1, int(os.environ.get("SA_SIDE_WGRAD_LAG", "3")))
    except ValueError:
        raise ValueError(f"SA_SIDE_WGRAD_LAG={os.environ.get('SA_SIDE_WGRAD_LAG')!r}: expected the number of side-stream launches whose operands stay alive (an integer >= 1)") from None

    def __init__(self, dev):
        self.main = torch.cuda.current_stream(dev)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
        self.side = _SIDE_STREAMS[key]
        self.inflight = collections.deque()

    def run(self, fn, *operands):
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            fn()                                  # (workspaces allocated in here come from the side stream's own pool: stream-ordered reuse)
            done = torch.cuda.Event()
            done.record(self.side)
        self.inflight.append((done, operands))
        while len(self.inflight) > self.LAG:
            old, ops = self.inflight.popleft()
            self.main.wait_event(old)
            del ops

    def join(self):
        self.main.wait_stream(self.side)
        self.inflight.clear()


class _GradCtx:
    def __init__(self, sink=None, side=None):
        self.sink, self.grads, self.side = sink, {}, side

    def wgrad(self, op, x, g, dw, db):
        """op.wgrad(x, g, dw, db), on the side stream when there is one"""
        if self.side is None:
            op.wgrad(x, g, dw, db)
        else:
            self.side.run(lambda: op.wgrad(x, g, dw, db), x, g, dw, db)

    def buf(self, p):
        if p is None:
            return None
        if self.sink is not None:
            b = self.sink.buffer(p)
            if b is not None:
                return b
        t = self.grads.get(p)
        if t is None:
            t = torch.zeros_like(p)
            self.grads[p] = t
        return t

    def done(self, *params):
        if self.sink is None:
            return
        if self.side is not None and getattr(self.sink, "active", True):
            # DDP: a bucket's all-reduce is ordered behind an event of the CURRENT stream (runtime/ddp.GradReducer._launch); the gradients of these parameters
            # were queued on the main stream (bias / gate gradients) and on the weight-gradient stream, so report them from the latter after it has caught up
            self.side.side.wait_stream(self.side.main)
            with torch.cuda.stream(self.side.side):
                self._report(params)
            return
        self._report(params)

    def _report(self, params):
        params = [p for p in params if p is not None]
        if hasattr(self.sink, "flush"):
            tok = self.sink.flush()       # optimizer slices deferred by EARLIER reports: everything that reads their parameters is queued (runtime/ddp.GradReducer.flush)
            for p in params:
                self.sink.ready(p, tok)
        else:
            for p in params:
                self.sink.ready(p)


class _LayerEngine:
    """Hand-scheduled forward / backward of ONE (attention, feed-forward) block."""

    def __init__(self, attn_wrap, ff_wrap, dim, dtype, use_rezero):
        self.aw, self.fw, self.dim, self.dtype, self.rezero = attn_wrap, ff_wrap, dim, dtype, use_rezero
        sa: SelfAttention = attn_wrap.fn
        ff: FeedForward = ff_wrap.fn.fn
        self.sa, self.ff = sa, ff
        self.H, self.G, self.dh = sa.heads, sa.global_heads, sa.dim_head
        self.L = self.H - self.G
        self.m = sa.fast_attention.nb_features
        self.LDF = _ru(self.m, 16)
        if self.G > 0 and (self.LDF > 272 or self.dh != 64):
            raise NotImplementedError("FAVOR+ kernels are built for dim_head 64 (nb_features <= 272)")
        self.W = sa.local_attn.window_size if sa.local_attn is not None else 0
        self.ops = {n: _lin(getattr(sa, n), dtype) for n in ("to_q", "to_k", "to_v", "to_out")}
        self.ops["w1"], self.ops["w2"] = _lin(ff.w1, dtype), _lin(ff.w2, dtype)
        self._pscaled = None
        # state_flags bit 2 of the fused scans: fp32 (parity) mode keeps every product on the exact-fp32 MFMA; the bf16 throughput mode uses the
        # split-bf16 kernels (~1e-5 relative, far below the rounding of its dense layers)
        self._xf = 4 if dtype == torch.float32 else 0
        self._pop = None
        self._ptiles = None
        self._rot = None
        self._one = None
        self._ws = None
        self.layer_pos = None      # rotary_position_emb=True: the wrapper's FixedPositionalEmbedding(dim_head) -- q / k of the GLOBAL heads are rotated pairwise

    def _rotate_global(self, q, k, B, N, transpose):
        """performer_pytorch 1.0.11 apply_rotary_pos_emb on the FAVOR+ heads (columns [0, G dh) of q and k), in place; transpose=1: the adjoint on dq / dk"""
        if self.layer_pos is None or self.G == 0:
            return
        lib, st = _ffi.lib(), _ffi.stream()
        tab = self.layer_pos.emb
        assert N <= tab.shape[0] and tab.shape[1] == self.dh and tab.is_contiguous()
        R, inner = B * N, self.H * self.dh
        if k.data_ptr() - q.data_ptr() == inner * 4 and q.stride(0) == k.stride(0):   # column blocks of one matrix: one launch
            _ck(lib.sa_rotary_pairs(_ffi.ptr(q), q.stride(0), 0, self.G, self.dh, _ffi.ptr(tab), _ffi.ptr(q), q.stride(0), 0, N, R, transpose, 2, inner, inner, st),
                "sa_rotary_pairs(q|k)")
        else:
            for t in (q, k):
                _ck(lib.sa_rotary_pairs(_ffi.ptr(t), t.stride(0), 0, self.G, self.dh, _ffi.ptr(tab), _ffi.ptr(t), t.stride(0), 0, N, R, transpose, 1, 0, 0, st),
                    "sa_rotary_pairs")

    def invalidate(self):
        for op in self.ops.values():
            op.invalidate()

    @staticmethod
    def _stacked(ts):
        """One [sum rows, cols] view over equally shaped 2-D tensors that sit back to back in memory (the flat parameter / gradient buffers of
        runtime.optim.FlatParams), or None."""
        t0 = ts[0]
        if any(t is None or t.shape != t0.shape or t.dtype != t0.dtype or not t.is_contiguous() for t in ts):
            return None
        nb = t0.numel() * t0.element_size()
        if any(t.data_ptr() != t0.data_ptr() + i * nb for i, t in enumerate(ts)):
            return None
        if t0.storage_offset() + len(ts) * t0.numel() > t0.untyped_storage().nbytes() // t0.element_size():
            return None
        return torch.as_strided(t0, (len(ts) * t0.shape[0], t0.shape[1]), (t0.shape[1], 1))

    def _sync(self):
        sa, ff = self.sa, self.ff
        for n in ("to_q", "to_k", "to_v", "to_out"):
            self.ops[n].weight, self.ops[n].bias = getattr(sa, n).weight, getattr(sa, n).bias
        # to_q / to_k / to_v as ONE dense layer (one forward, one data-gradient and one weight-gradient launch instead of three each) when the
        # three bias-free weights are adjacent in memory
        wqkv = None
        if not debug.host("no_fused_qkv") and sa.to_q.bias is None and sa.to_k.bias is None and sa.to_v.bias is None:
            wqkv = self._stacked([sa.to_q.weight.detach(), sa.to_k.weight.detach(), sa.to_v.weight.detach()])
        if wqkv is None:
            self.ops.pop("to_qkv", None)
        elif "to_qkv" not in self.ops:
            self.ops["to_qkv"] = ConvOp("conv", wqkv.shape[1], wqkv.shape[0], 1, 1, 0, wqkv.view(*wqkv.shape, 1, 1, 1), None, self.dtype)
        else:
            self.ops["to_qkv"].weight = wqkv.view(*wqkv.shape, 1, 1, 1)
        self.ops["w1"].weight, self.ops["w1"].bias = ff.w1.weight, ff.w1.bias
        self.ops["w2"].weight, self.ops["w2"].bias = ff.w2.weight, ff.w2.bias

    def _proj_op(self):
        pm = self.sa.fast_attention.projection_matrix
        ver = (pm.data_ptr(), pm._version, getattr(pm, "_sa_epoch", 0), debug.host("no_proj_bf16"))
        if self._pop is None or self._pop[0] != ver:
            c = self.dh ** -0.25
            ps = (pm.detach() * c).contiguous()  # data_normalizer folded into the operand
            if self.dtype == torch.bfloat16 and not debug.host("no_proj_bf16"):
                # throughput mode: the projection OPERAND is a bf16 copy of the folded matrix, as every dense weight of this mode is (fp32 master in the buffer
                # `projection_matrix`, bf16 operand in the launches) -- the split-bf16 kernels then see lo(P) = 0 and run two products instead of three per
                # feature-map / adjoint GEMM and move half the projection bytes (csrc/favor_fused.hip: PT_FLAG_OFF).  SA_NO_PROJ_BF16=1 keeps the fp32 operand.
                ps = ps.to(torch.bfloat16).float().contiguous()
            op = ConvOp("conv", self.dh, self.m, 1, 1, 0, ps, None, torch.float32)
            self._pop = (ver, op, ps)
        return self._pop[1]

    def _proj_tiles(self):
        """the projection matrix as split-bf16 slab tiles for the fused FAVOR+ kernels (rebuilt when the matrix is redrawn)"""
        self._proj_op()
        ver, ps = self._pop[0], self._pop[2]
        if self._ptiles is None or self._ptiles[0] != ver:
            tiles = torch.empty(5 * 16384, dtype=torch.uint8, device=ps.device)
            _ck(_ffi.lib().sa_favor_fused_proj_tiles(_ffi.ptr(ps), self.m, _ffi.ptr(tiles), _ffi.stream()), "sa_favor_fused_proj_tiles")
            self._ptiles = (ver, tiles)
        return self._ptiles[1], ps

    def _fused_favor(self):
        """throughput mode: FAVOR+ with the feature maps recomputed on chip (csrc/favor_fused.hip) -- no [B N G, 272] tensor in HBM"""
        return not self._xf and self.G > 0 and self.dh == 64 and self.m <= 272 and not debug.host("no_fused_favor")

    def _rot_tables(self, N, dev):
        if self._rot is None or self._rot[0] != (N, dev):
            inv = self.sa.local_attn.rel_pos.inv_freq.to(dev)
            fr = torch.einsum("i,j->ij", torch.arange(N, device=dev, dtype=torch.float32), inv)
            fr = torch.cat((fr, fr), dim=-1)
            self._rot = ((N, dev), fr.cos().contiguous(), fr.sin().contiguous())
        return self._rot[1], self._rot[2]

    def _scan_ws(self, B, N, G, dev):
        """scratch for the segment-parallel scans (states of <= 16 segments per (batch, head)); reused by every call of this layer"""
        n = _ffi.lib().sa_favor_scan_workspace_bytes(B, N, G, self.LDF, self.dh) // 4
        if self._ws is None or self._ws.numel() < n or self._ws.device != dev:
            self._ws = torch.empty(n, dtype=torch.float32, device=dev)
        return self._ws

    def _gate(self, wrap, dev):
        if self.rezero:
            return wrap.g
        if self._one is None or self._one.device != dev:
            self._one = torch.ones((), device=dev)
        return self._one

    def _pre(self, wrap, x, R):
        """input of the wrapped fn: x itself (ReZero) or LayerNorm(x) (PreLayerNorm); returns (xin fp32, stats)"""
        if self.rezero:
            return x, None
        lib, st = _ffi.lib(), _ffi.stream()
        y = torch.empty_like(x)
        stats = torch.empty(2 * R, dtype=torch.float32, device=x.device)
        _ck(lib.sa_layernorm_fwd(_ffi.ptr(x), _ffi.ptr(wrap.norm.weight), _ffi.ptr(wrap.norm.bias), _ffi.ptr(y), None, 0, _ffi.ptr(stats), R, self.dim,
                                 wrap.norm.eps, st), "sa_layernorm_fwd")
        return y, stats

    # ---------------------------------------------------------------------------------------------- forward
    def fwd(self, x, B, N, tape, x_lp=None):
        """x_lp: the compute-dtype copy of x when the producer already wrote one (the previous block's ReZero kernel); the copy of this block's
        output is left in ``self.out_lp`` for the next block (saves two cast launches per block on the ReZero path)."""
        self._sync()
        lib, st, dev, T = _ffi.lib(), _ffi.stream(), x.device, self.dtype
        R, H, G, L, dh, m, LDF = B * N, self.H, self.G, self.L, self.dh, self.m, self.LDF
        inner = H * dh
        f32 = torch.float32
        lp = self.rezero and T != f32            # the residual kernel can emit the low-precision copy the next GEMM wants
        xa, st_a = self._pre(self.aw, x, R)
        xaT = x_lp if (lp and x_lp is not None) else _cast(xa, T)
        if "to_qkv" in self.ops:
            qkv = self.ops["to_qkv"].fprop(_as5(xaT), out_dtype=f32).view(R, 3 * inner)
            q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
        else:
            q = self.ops["to_q"].fprop(_as5(xaT), out_dtype=f32).view(R, inner)
            k = self.ops["to_k"].fprop(_as5(xaT), out_dtype=f32).view(R, inner)
            v = self.ops["to_v"].fprop(_as5(xaT), out_dtype=f32).view(R, inner)
        qs = q.stride(0)   # row stride of q / k / v (3 * inner when they are column blocks of one matrix)
        self._rotate_global(q, k, B, N, 0)       # (rotary_position_emb=True only; the tape keeps the rotated rows -- what the FAVOR+ backward differentiates)
        attn = torch.empty(R, inner, dtype=f32, device=dev)
        # throughput mode: the attention kernels write the bf16 operand of to_out next to the fp32 rows (no cast launch); needs every head on a kernel that can
        attn_lp = (torch.empty(R, inner, dtype=T, device=dev)
                   if (T == torch.bfloat16 and (G == 0 or self._fused_favor()) and not debug.host("no_lp_mirrors")) else None)
        sv = dict(x=x, xa=xa, xaT=xaT, st_a=st_a, q=q, k=k, v=v, attn=attn)
        # throughput mode with both kinds of heads: the local-window heads' blocks ride in the FAVOR+ launches (sa_local_attn_args): rotary first, no launch of their own
        la_args = None
        if G > 0 and L > 0 and self._fused_favor() and not debug.host("no_attn_colaunch"):
            la_args = self._local_fwd_prep(q, k, v, qs, attn, attn_lp, B, N, R, dev, sv)
        if G > 0 and self._fused_favor():
            tiles, ps = self._proj_tiles()
            offq = torch.empty(R * G, dtype=f32, device=dev)
            offk = torch.empty(R * G, dtype=f32, device=dev)
            amq = torch.empty(R * G, dtype=torch.int32, device=dev)
            gws = torch.empty(1, dtype=torch.int64, device=dev)
            inv = torch.empty(R * G, dtype=f32, device=dev)
            nst = lib.sa_favor_fused_state_bytes(B, N, G, m) // 4
            if tape is not None:     # training: the chunk prefixes (sum k' (x) v | sum k') are kept for the dq' scan of the backward pass
                state = torch.empty(nst, dtype=f32, device=dev)
            else:
                if self._ws is None or self._ws.numel() < nst or self._ws.device != dev:
                    self._ws = torch.empty(nst, dtype=f32, device=dev)
                state = self._ws
            t0 = _favor_bracket_begin()
            _ck(lib.sa_favor_fused_prepass(_ffi.ptr(q), _ffi.ptr(k), qs, G, _ffi.ptr(tiles), _ffi.ptr(offq), _ffi.ptr(amq), _ffi.ptr(offk), _ffi.ptr(gws), R * G, m, dh, st),
                "sa_favor_fused_prepass")
            rc = lib.sa_favor_fused_fwd(_ffi.ptr(q), _ffi.ptr(k), _ffi.ptr(v), qs, _ffi.ptr(tiles), _ffi.ptr(ps), _ffi.ptr(offq), _ffi.ptr(offk), _ffi.ptr(gws),
                                        _ffi.ptr(attn), inner, _ffi.ptr(inv), 1e-6, B, N, G, m, _ffi.ptr(state), _ffi.ptr(attn_lp),
                                        ctypes.byref(la_args) if la_args is not None else None, st)
            if rc == _ffi.SA_EUNSUPPORTED and la_args is not None:      # (exact-fp32 local attention: its kernels cannot share the launch) -> separate launches
                la_args = None
                rc = lib.sa_favor_fused_fwd(_ffi.ptr(q), _ffi.ptr(k), _ffi.ptr(v), qs, _ffi.ptr(tiles), _ffi.ptr(ps), _ffi.ptr(offq), _ffi.ptr(offk), _ffi.ptr(gws),
                                            _ffi.ptr(attn), inner, _ffi.ptr(inv), 1e-6, B, N, G, m, _ffi.ptr(state), _ffi.ptr(attn_lp), None, st)
            _ck(rc, "sa_favor_fused_fwd")
            _favor_bracket_end(t0, "favor_prepass+fstates+fout_a" + ("_la" if la_args is not None else ""), B * N * G * FAVOR_FWD_FLOP_PER_HEAD_ROW(m, dh))
            sv.update(fused=True, offq=offq, offk=offk, amq=amq, gws=gws, inv=inv, scan_state=state if tape is not None else None)
        elif G > 0:
            pop = self._proj_op()
            if self._xf:   # fp32 parity mode: exact-fp32 GEMM on contiguous copies of the global-head columns
                qg = q[:, : G * dh].contiguous()
                kg = k[:, : G * dh].contiguous()
                sst = G * dh
                ddq = pop.fprop(qg.view(1, 1, 1, R * G, dh), out_channels_stride=LDF, use_bias=False).view(R * G, LDF)
                ddk = pop.fprop(kg.view(1, 1, 1, R * G, dh), out_channels_stride=LDF, use_bias=False).view(R * G, LDF)
            else:          # throughput mode: HBM-bound split-bf16 projection kernels reading the head blocks of q / k in place
                qg, kg, sst = q, k, qs
                ps = self._pop[2]
                ddq = torch.empty(R * G, LDF, dtype=f32, device=dev)
                ddk = torch.empty(R * G, LDF, dtype=f32, device=dev)
            qf, kf = torch.empty_like(ddq), torch.empty_like(ddk)
            gws = torch.empty(2, dtype=torch.int64, device=dev)
            kmode = 0      # the keys' feature map finds the global maximum itself ...
            if not self._xf:
                # queries: projection and feature map in one launch (the row maximum is local to the block)
                _ck(lib.sa_favor_project_features(_ffi.ptr(q), qs, G, _ffi.ptr(ps), _ffi.ptr(ddq), _ffi.ptr(qf), R * G, m, LDF, dh, st), "sa_favor_project_features(q)")
                _ck(lib.sa_favor_project(_ffi.ptr(k), qs, G, _ffi.ptr(ps), _ffi.ptr(ddk), _ffi.ptr(gws), R * G, m, LDF, dh, st), "sa_favor_project(k)")
                kmode = 2  # ... unless the projection kernel already left it in gws (from its accumulators: no extra pass over ddk)
            else:
                _ck(lib.sa_favor_features_fwd(_ffi.ptr(ddq), _ffi.ptr(qg), sst, 0, G, dh, 1, _ffi.ptr(qf), None, R * G, m, LDF, st), "favor_features(q)")
            _ck(lib.sa_favor_features_fwd(_ffi.ptr(ddk), _ffi.ptr(kg), sst, 0, G, dh, kmode, _ffi.ptr(kf), _ffi.ptr(gws), R * G, m, LDF, st), "favor_features(k)")
            ws = self._scan_ws(B, N, G, dev)
            if tape is not None and (not debug.host("no_fused_sums")):   # training: the chunk states (sum k' (x) v, sum k') are kept for the dq' scan of the backward pass
                ws = torch.empty_like(ws)
            inv = torch.empty(R * G, dtype=f32, device=dev)
            Z = None
            # normaliser fused into the scan (the running key sums ride along as an extra state column): no cumsum / den passes
            rc = lib.sa_favor_scan_a_norm(_ffi.ptr(kf), _ffi.ptr(qf), _ffi.ptr(v), qs, 0, _ffi.ptr(attn), inner, 0, _ffi.ptr(inv), 1e-6, B, N, G, LDF, dh,
                                          _ffi.ptr(ws), self._xf, st) if (not debug.host("no_fused_sums")) else _ffi.SA_EUNSUPPORTED
            if rc == _ffi.SA_EUNSUPPORTED:
                Z = torch.empty_like(kf)
                _ck(lib.sa_cumsum_rows(_ffi.ptr(kf), None, _ffi.ptr(Z), B, N, G, LDF, 0, _ffi.ptr(ws), st), "sa_cumsum_rows")
                _ck(lib.sa_favor_den(_ffi.ptr(qf), _ffi.ptr(Z), 1e-6, _ffi.ptr(inv), R * G, m, LDF, st), "sa_favor_den")
                _ck(lib.sa_favor_scan_a(_ffi.ptr(kf), _ffi.ptr(qf), _ffi.ptr(v), qs, 0, None, _ffi.ptr(attn), inner, 0, _ffi.ptr(inv), B, N, G, LDF, dh, 0, 0,
                                        _ffi.ptr(ws), st), "sa_favor_scan_a")
            else:
                _ck(rc, "sa_favor_scan_a_norm")
            sv.update(qg=qg, kg=kg, ddq=ddq, ddk=ddk, qf=qf, kf=kf, gws=gws, Z=Z, inv=inv, scan_state=ws if (Z is None and tape is not None) else None)
        if L > 0 and "lse" not in sv:          # (not prepared for the co-launch)
            self._local_fwd_prep(q, k, v, qs, attn, attn_lp, B, N, R, dev, sv)
        if L > 0 and la_args is None:
            qr, kr, lse = sv["qr"], sv["kr"], sv["lse"]
            _ck(lib.sa_local_attn_fwd(_ffi.ptr(qr), L * dh, 0, _ffi.ptr(kr), L * dh, 0, _ffi.ptr(v), qs, G * dh, _ffi.ptr(attn), inner, G * dh, _ffi.ptr(lse),
                                      B, N, L, self.W, dh, _ffi.ptr(attn_lp), st), "sa_local_attn_fwd")
        attnT = attn_lp if attn_lp is not None else _cast(attn, T)
        ga = self._gate(self.aw, dev)
        gf = self._gate(self.fw, dev)
        fuse_epi = lp and not debug.host("no_fused_epilogues")
        if fuse_epi:
            # throughput mode: x1 = x + g F leaves the to_out launch itself (fp32 residual stream + its bf16 copy for the next dense layer + the branch
            # output F the ReZero backward needs) -- no sa_rezero_fwd launch, F is never re-read
            x1, Fa, x1T = self.ops["to_out"].fprop(_as5(attnT), out_dtype=f32, alpha=ga, addend=_as5(x), want_pre=True, want_lp=True)
            x1, Fa, x1T = x1.view(R, self.dim), Fa.view(R, self.dim), x1T.view(R, self.dim)
        else:
            Fa = self.ops["to_out"].fprop(_as5(attnT)).view(R, self.dim)
            x1 = torch.empty_like(x)
            x1T = torch.empty(x.shape, dtype=T, device=dev) if lp else None
            _ck(lib.sa_rezero_fwd(_ffi.ptr(x), _ffi.ptr(Fa), _ffi.dtype_id(Fa.dtype), _ffi.ptr(ga), _ffi.ptr(x1), _ffi.ptr(x1T), _ffi.dtype_id(T) if lp else 0, x.numel(), st),
                "sa_rezero_fwd")
        xf, st_f = self._pre(self.fw, x1, R)
        xfT = x1T if lp else _cast(xf, T)
        if fuse_epi:
            # h = gelu(u) and the pre-activation u (for the GELU backward) from ONE launch; x2 = x1 + g F from the w2 launch
            h, u, _ = self.ops["w1"].fprop(_as5(xfT), act=_ffi.ACT_GELU, want_pre=True)
            h, u = h.view(R, -1), u.view(R, -1)
            x2, Ff, x2T = self.ops["w2"].fprop(_as5(h), out_dtype=f32, alpha=gf, addend=_as5(x1), want_pre=True, want_lp=True)
            x2, Ff, x2T = x2.view(R, self.dim), Ff.view(R, self.dim), x2T.view(R, self.dim)
        else:
            u = self.ops["w1"].fprop(_as5(xfT)).view(R, -1)
            h = torch.empty_like(u)
            _ck(lib.sa_gelu(_ffi.ptr(u), _ffi.dtype_id(u.dtype), _ffi.ptr(h), _ffi.dtype_id(h.dtype), u.numel(), st), "sa_gelu")
            Ff = self.ops["w2"].fprop(_as5(h)).view(R, self.dim)
            x2 = torch.empty_like(x)
            x2T = torch.empty(x.shape, dtype=T, device=dev) if lp else None
            _ck(lib.sa_rezero_fwd(_ffi.ptr(x1), _ffi.ptr(Ff), _ffi.dtype_id(Ff.dtype), _ffi.ptr(gf), _ffi.ptr(x2), _ffi.ptr(x2T), _ffi.dtype_id(T) if lp else 0, x.numel(), st),
                "sa_rezero_fwd")
        self.out_lp = x2T
        if tape is not None:
            sv.update(attnT=attnT, Fa=Fa, x1=x1, xf=xf, xfT=xfT, st_f=st_f, u=u, h=h, Ff=Ff)
            tape.append(sv)
        return x2

    # ---------------------------------------------------------------------------------------------- stateful decoding (one position)
    def _local_fwd_prep(self, q, k, v, qs, attn, attn_lp, B, N, R, dev, sv):
        """Rotated q | k of the local heads + the log-sum-exp buffer (kept in `sv` for the backward pass); returns the sa_local_attn_args of the forward launch."""
        lib, st = _ffi.lib(), _ffi.stream()
        G, L, dh = self.G, self.L, self.dh
        inner = self.H * dh
        f32 = torch.float32
        cosb, sinb = self._rot_tables(N, dev)
        qkr = torch.empty(2, R, L * dh, dtype=f32, device=dev)
        qr, kr = qkr[0], qkr[1]
        if k.data_ptr() - q.data_ptr() == inner * 4 and q.stride(0) == k.stride(0):   # q | k are column blocks of one matrix: one launch rotates both
            _ck(lib.sa_rotary_groups(_ffi.ptr(q), qs, G * dh, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(qkr), L * dh, 0, N, R, 0, 0, 2, inner, R * L * dh, None, st),
                "sa_rotary_groups(q|k)")
        else:
            _ck(lib.sa_rotary(_ffi.ptr(q), qs, G * dh, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(qr), L * dh, 0, N, R, 0, 0, st), "sa_rotary(q)")
            _ck(lib.sa_rotary(_ffi.ptr(k), qs, G * dh, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(kr), L * dh, 0, N, R, 0, 0, st), "sa_rotary(k)")
        lse = torch.empty(R * L, dtype=f32, device=dev)
        sv.update(qr=qr, kr=kr, lse=lse)
        a = _ffi.LocalAttnArgs()
        a.q, a.k, a.v = qr.data_ptr(), kr.data_ptr(), v.data_ptr()
        a.q_stride, a.q_off, a.k_stride, a.k_off, a.v_stride, a.v_off, a.o_stride, a.o_off = L * dh, 0, L * dh, 0, qs, G * dh, inner, G * dh
        a.o, a.lse, a.o_lp = attn.data_ptr(), lse.data_ptr(), (attn_lp.data_ptr() if attn_lp is not None else None)
        a.L, a.W = L, self.W
        return a

    def new_state(self, B, N, dev):
        """Per-layer decoding state: FAVOR+ running sums of the global heads (rescalable, see csrc/performer.hip) and the rotated-key /
        value caches of the local heads."""
        G, L, dh, LDF = self.G, self.L, self.dh, self.LDF
        f32 = torch.float32
        stt = dict(smax=torch.empty(2, dtype=f32, device=dev), kmax=torch.empty(2, dtype=torch.int32, device=dev),
                   dd=torch.zeros(2, max(B * G, 1), LDF, dtype=f32, device=dev),
                   E=torch.empty(max(B * G, 1), LDF * dh, dtype=f32, device=dev), Ez=torch.empty(max(B * G, 1), LDF, dtype=f32, device=dev),
                   V1=torch.empty(max(B * G, 1), dh, dtype=f32, device=dev), kc=torch.empty(B, max(L, 1), N, dh, dtype=f32, device=dev),
                   vc=torch.empty(B, max(L, 1), N, dh, dtype=f32, device=dev),
                   lpart=torch.empty(B * max(L, 1) * 4 * 66, dtype=f32, device=dev))      # sa_attn_step: partial softmax of the local heads (4 key segments)
        self.reset_state(stt)
        return stt

    @staticmethod
    def reset_state(stt):
        for k, v in stt.items():
            if k == "smax":
                v.fill_(float("-inf"))
            elif k == "kmax":
                v.fill_(-2139095041)   # order-preserving integer encoding of -inf (0x807fffff)
            else:
                v.zero_()

    def _gemv(self, x, mods, y, act=0, res=None, gate=None, round_out=False):
        """y[b] = epi(x[b] @ cat(W_i)^T + cat(b_i)) for the B rows of a decode step (sa_gemv_rows: streams the fp32 weights once)."""
        lib, st = _ffi.lib(), _ffi.stream()
        B, cin = x.shape
        n = len(mods)
        rnd = 1 if self.dtype == torch.bfloat16 else 0
        rw = rnd
        ws = [m.weight for m in mods]
        if rnd and not debug.host("no_decode_bf16_weights"):
            # bf16 compute: the decode step streams bf16 COPIES of the parameters (the values the kernel would round to anyway): half the bytes per token, and the
            # 200 MB of a 24-layer network stay resident in the 256 MB Infinity Cache between tokens.  Made on first use per parameter version (outside graph capture:
            # the sampler's warm-up step runs first).
            ws = [self._decode_weight(m) for m in mods]
            rw = 2
        wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
        bp = (ctypes.c_void_p * n)(*[(m.bias.data_ptr() if m.bias is not None else None) for m in mods])
        so = (ctypes.c_int32 * n)(*[m.weight.shape[0] for m in mods])
        _ck(lib.sa_gemv_rows(_ffi.ptr(x), x.stride(0), cin, B, n, wp, bp, so, _ffi.ptr(y), y.stride(0), act, _ffi.ptr(res) if res is not None else None,
                             res.stride(0) if res is not None else 0, _ffi.ptr(gate), rnd, rw, 1 if (round_out and rnd) else 0, st), "sa_gemv_rows")
        return y

    def _decode_weight(self, mod):
        cache = self.__dict__.setdefault("_dec_w", {})
        ent = cache.get(id(mod))
        w = mod.weight
        # (the optimizer writes parameters through raw pointers, which `_version` does not see: the sampler drops this cache at the start of every sample())
        if ent is None or ent[0] != w._version or ent[1].device != w.device or ent[2] != w.data_ptr():
            ent = (w._version, w.detach().to(torch.bfloat16).contiguous(), w.data_ptr())
            cache[id(mod)] = ent
        return ent[1]

    def step(self, x, B, N, pos, stt):
        """x [B, dim] fp32 = the block input at position *pos (device int32) -> block output; updates `stt`.  Six launches (seven without sa_attn_step), no host
        synchronisation and no host-side dependence on the position, so a whole token step can be captured in a HIP graph."""
        lib, st, dev = _ffi.lib(), _ffi.stream(), x.device
        sa, ff = self.sa, self.ff
        H, G, L, dh, m, LDF = self.H, self.G, self.L, self.dh, self.m, self.LDF
        inner = H * dh
        f32 = torch.float32
        xa, _ = self._pre(self.aw, x, B)
        qkv = torch.empty(B, 3 * inner, dtype=f32, device=dev)
        self._gemv(xa, [sa.to_q, sa.to_k, sa.to_v], qkv)
        attn = torch.empty(B, inner, dtype=f32, device=dev)
        merged = G > 0 and L > 0 and dh == 64 and (2 * self.W + 3) // 4 <= 256 and not debug.host("no_attn_step_merge")
        if merged:
            # both kinds of heads in TWO launches: [projections | local heads over four key segments], [FAVOR+ update | combine of the segments]
            self._proj_op()
            cosb, sinb = self._rot_tables(N, dev)
            _ck(lib.sa_attn_step(_ffi.ptr(qkv), 3 * inner, inner, _ffi.ptr(self._pop[2]), B, G, L, dh, m, LDF, _ffi.ptr(stt["smax"]), _ffi.ptr(stt["kmax"]),
                                 _ffi.ptr(stt["dd"]), _ffi.ptr(stt["E"]), _ffi.ptr(stt["Ez"]), _ffi.ptr(stt["V1"]), _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(stt["kc"]),
                                 _ffi.ptr(stt["vc"]), N, self.W, _ffi.ptr(stt["lpart"]), _ffi.ptr(pos), _ffi.ptr(attn), inner, st), "sa_attn_step")
        if G > 0 and not merged:
            self._proj_op()
            ps = self._pop[2]    # projection matrix with the data normaliser folded in
            _ck(lib.sa_favor_step(_ffi.ptr(qkv), 3 * inner, 0, _ffi.ptr(qkv), 3 * inner, inner, _ffi.ptr(qkv), 3 * inner, 2 * inner, _ffi.ptr(ps), B, G, dh, m, LDF,
                                  _ffi.ptr(stt["smax"]), _ffi.ptr(stt["kmax"]), _ffi.ptr(stt["dd"]), _ffi.ptr(stt["E"]), _ffi.ptr(stt["Ez"]), _ffi.ptr(stt["V1"]),
                                  _ffi.ptr(pos), _ffi.ptr(attn), inner, 0, st), "sa_favor_step")
        if L > 0 and not merged:
            cosb, sinb = self._rot_tables(N, dev)
            _ck(lib.sa_local_attn_step(_ffi.ptr(qkv), 3 * inner, G * dh, _ffi.ptr(qkv), 3 * inner, inner + G * dh, _ffi.ptr(qkv), 3 * inner, 2 * inner + G * dh,
                                       _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(stt["kc"]), _ffi.ptr(stt["vc"]), _ffi.ptr(pos), B, N, L, self.W, dh, _ffi.ptr(attn),
                                       inner, G * dh, st), "sa_local_attn_step")
        x1 = torch.empty_like(x)
        self._gemv(attn, [sa.to_out], x1, res=x, gate=self._gate(self.aw, dev), round_out=True)
        xf, _ = self._pre(self.fw, x1, B)
        h = torch.empty(B, ff.w1.weight.shape[0], dtype=f32, device=dev)
        self._gemv(xf, [ff.w1], h, act=1, round_out=True)
        x2 = torch.empty_like(x)
        self._gemv(h, [ff.w2], x2, res=x1, gate=self._gate(self.fw, dev), round_out=True)
        return x2

    # ---------------------------------------------------------------------------------------------- backward
    def _post_bwd(self, wrap, dy, Fout, gc, dev):
        """through y = x + g * F: returns dF (compute dtype) and accumulates dg"""
        lib, st = _ffi.lib(), _ffi.stream()
        dF = torch.empty(dy.shape, dtype=self.dtype, device=dev)
        g = self._gate(wrap, dev)
        dg = gc.buf(wrap.g) if self.rezero else torch.zeros((), device=dev)
        if self.rezero and debug.deterministic():
            # --deterministic: d g = <dy, F> in a fixed order (sa_dot_det) instead of one atomic per block; the elementwise half of the kernel runs as usual
            scratch = torch.zeros(1 + 1024, dtype=torch.float32, device=dev)
            _ck(lib.sa_rezero_bwd(_ffi.ptr(dy), _ffi.ptr(Fout), _ffi.dtype_id(Fout.dtype), _ffi.ptr(g), _ffi.ptr(dF), _ffi.dtype_id(dF.dtype), _ffi.ptr(scratch),
                                  dy.numel(), st), "sa_rezero_bwd")
            _ck(lib.sa_dot_det(_ffi.ptr(dy), _ffi.ptr(Fout), _ffi.dtype_id(Fout.dtype), dy.numel(), _ffi.ptr(dg), 1, _ffi.ptr(scratch[1:]), st), "sa_dot_det")
            gc.done(wrap.g)
            return dF
        _ck(lib.sa_rezero_bwd(_ffi.ptr(dy), _ffi.ptr(Fout), _ffi.dtype_id(Fout.dtype), _ffi.ptr(g), _ffi.ptr(dF), _ffi.dtype_id(dF.dtype), _ffi.ptr(dg),
                              dy.numel(), st), "sa_rezero_bwd")
        if self.rezero:
            gc.done(wrap.g)
        return dF

    def _pre_bwd(self, wrap, dxin, dres, xres, stats, R, gc):
        """gradient wrt the block input: residual path dres + path through the (identity | LayerNorm) pre-op"""
        if self.rezero:
            return dxin  # the dgrad epilogues already added dres
        lib, st = _ffi.lib(), _ffi.stream()
        dx = torch.empty_like(dres)
        if debug.deterministic():   # the weight / bias gradients as fixed-order column sums (see _LayerNormFn.backward)
            from ...engine import colsum_det
            junk = torch.zeros(2, self.dim, dtype=torch.float32, device=dres.device)
            _ck(lib.sa_layernorm_bwd(_ffi.ptr(dxin), _ffi.ptr(xres), _ffi.ptr(wrap.norm.weight), _ffi.ptr(stats), _ffi.ptr(dx), _ffi.ptr(junk[0]), _ffi.ptr(junk[1]), R,
                                     self.dim, st), "sa_layernorm_bwd")
            prod = torch.empty_like(dxin)
            _ck(lib.sa_layernorm_dwprod(_ffi.ptr(dxin), _ffi.ptr(xres), _ffi.ptr(stats), _ffi.ptr(prod), R, self.dim, st), "sa_layernorm_dwprod")
            colsum_det(prod.view(R, self.dim), self.dim, gc.buf(wrap.norm.weight))
            colsum_det(dxin.view(R, self.dim), self.dim, gc.buf(wrap.norm.bias))
        else:
            _ck(lib.sa_layernorm_bwd(_ffi.ptr(dxin), _ffi.ptr(xres), _ffi.ptr(wrap.norm.weight), _ffi.ptr(stats), _ffi.ptr(dx), _ffi.ptr(gc.buf(wrap.norm.weight)),
                                     _ffi.ptr(gc.buf(wrap.norm.bias)), R, self.dim, st), "sa_layernorm_bwd")
        gc.done(wrap.norm.weight, wrap.norm.bias)
        _ck(lib.sa_axpy(_ffi.ptr(dx), _ffi.ptr(dres), 1.0, dx.numel(), st), "sa_axpy")
        return dx

    def bwd(self, dx2, sv, B, N, gc: _GradCtx):
        self._sync()
        lib, st, dev, T = _ffi.lib(), _ffi.stream(), dx2.device, self.dtype
        R, H, G, L, dh, m, LDF = B * N, self.H, self.G, self.L, self.dh, self.m, self.LDF
        inner = H * dh
        f32 = torch.float32
        ops, sa, ff = self.ops, self.sa, self.ff
        r5 = (1, 1, R)
        # ---- feed-forward block
        dFf = self._post_bwd(self.fw, dx2, sv["Ff"], gc, dev)
        gc.wgrad(ops["w2"], _as5(sv["h"]), _as5(dFf), gc.buf(ff.w2.weight), gc.buf(ff.w2.bias))
        gc.done(ff.w2.weight, ff.w2.bias)
        du = ops["w2"].dgrad(_as5(dFf), r5, mask=_as5(sv["u"]), mask_mode=MASK_GELU)
        gc.wgrad(ops["w1"], _as5(sv["xfT"]), du, gc.buf(ff.w1.weight), gc.buf(ff.w1.bias))
        gc.done(ff.w1.weight, ff.w1.bias)
        if self.rezero:
            dx1 = ops["w1"].dgrad(du, r5, addend=_as5(dx2), out_dtype=f32).view(R, self.dim)
        else:
            dxf = ops["w1"].dgrad(du, r5, out_dtype=f32).view(R, self.dim)
            dx1 = self._pre_bwd(self.fw, dxf, dx2, sv["x1"], sv["st_f"], R, gc)
        # ---- attention block
        dFa = self._post_bwd(self.aw, dx1, sv["Fa"], gc, dev)
        gc.wgrad(ops["to_out"], _as5(sv["attnT"]), _as5(dFa), gc.buf(sa.to_out.weight), gc.buf(sa.to_out.bias))
        gc.done(sa.to_out.weight, sa.to_out.bias)
        dattn = ops["to_out"].dgrad(_as5(dFa), r5, out_dtype=f32).view(R, inner)
        q, k, v, attn = sv["q"], sv["k"], sv["v"], sv["attn"]
        fused_qkv = "to_qkv" in ops and q.stride(0) == 3 * inner
        dqkv_lp = None
        local_done = False
        if fused_qkv:   # gradients as column blocks of one matrix, like q / k / v themselves: one cast, one wgrad, one dgrad below
            dqkv = torch.empty(R, 3 * inner, dtype=f32, device=dev)
            dq, dk, dv = dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:]
            if T == torch.bfloat16 and (G == 0 or sv.get("fused")) and not debug.host("no_lp_mirrors"):
                dqkv_lp = torch.empty(R, 3 * inner, dtype=T, device=dev)   # bf16 mirror written by the same kernels: operand of the q|k|v weight / data gradient
                dq_lp, dk_lp, dv_lp = dqkv_lp[:, :inner], dqkv_lp[:, inner:2 * inner], dqkv_lp[:, 2 * inner:]
        else:
            dq = torch.empty(R, inner, dtype=f32, device=dev)
            dk = torch.empty(R, inner, dtype=f32, device=dev)
            dv = torch.empty(R, inner, dtype=f32, device=dev)
        if not fused_qkv or dqkv_lp is None or (self.layer_pos is not None and G > 0):    # (rotated global heads: dq / dk are rotated back below, then cast)
            dqkv_lp = dq_lp = dk_lp = dv_lp = None
        qs = dq.stride(0)   # == q.stride(0): the kernels below address v / dv (and dq / dk) with one row stride
        assert qs == q.stride(0) == v.stride(0)
        if G > 0 and sv.get("fused"):
            tiles, ps = self._proj_tiles()
            nst = lib.sa_favor_fused_state_bytes(B, N, G, m) // 4
            if self._ws is None or self._ws.numel() < nst or self._ws.device != dev:
                self._ws = torch.empty(nst, dtype=f32, device=dev)
            dden = torch.empty(R * G, dtype=f32, device=dev)
            tsum = torch.empty(B * G * ((N + 63) // 64), dtype=f32, device=dev)
            la_args = None
            if L > 0 and sv.get("scan_state") is not None and not debug.host("no_attn_colaunch"):
                # the local-window heads' backward blocks ride in the FAVOR+ launches (as in the forward pass); dq / dk of the local heads come out in ROTATED
                # space (dqkr) and are rotated back below
                dqkr = torch.empty(2, R, L * dh, dtype=f32, device=dev)
                Db = torch.empty(R * L, dtype=f32, device=dev)
                la_args = _ffi.LocalAttnArgs()
                la_args.q, la_args.k, la_args.v = sv["qr"].data_ptr(), sv["kr"].data_ptr(), v.data_ptr()
                (la_args.q_stride, la_args.q_off, la_args.k_stride, la_args.k_off, la_args.v_stride, la_args.v_off, la_args.o_stride,
                 la_args.o_off) = L * dh, 0, L * dh, 0, qs, G * dh, inner, G * dh
                la_args.out, la_args.dout, la_args.lse_in = attn.data_ptr(), dattn.data_ptr(), sv["lse"].data_ptr()
                la_args.dq, la_args.dk, la_args.dv, la_args.Dbuf = dqkr[0].data_ptr(), dqkr[1].data_ptr(), dv.data_ptr(), Db.data_ptr()
                la_args.dv_lp = dv_lp.data_ptr() if dv_lp is not None else None
                la_args.L, la_args.W = L, self.W

            def favor_bwd(la):
                return lib.sa_favor_fused_bwd(_ffi.ptr(q), _ffi.ptr(k), _ffi.ptr(v), qs, _ffi.ptr(tiles), _ffi.ptr(ps), _ffi.ptr(sv["offq"]), _ffi.ptr(sv["amq"]),
                                              _ffi.ptr(sv["offk"]), _ffi.ptr(sv["gws"]), _ffi.ptr(dattn), _ffi.ptr(attn), inner, _ffi.ptr(sv["inv"]), _ffi.ptr(dq),
                                              _ffi.ptr(dk), _ffi.ptr(dv), B, N, G, m, _ffi.ptr(sv.get("scan_state")), _ffi.ptr(self._ws), _ffi.ptr(dden), _ffi.ptr(tsum),
                                              _ffi.ptr(dq_lp), _ffi.ptr(dk_lp), _ffi.ptr(dv_lp), ctypes.byref(la) if la is not None else None, st)
            t0 = _favor_bracket_begin()
            rc = favor_bwd(la_args)
            if rc == _ffi.SA_EUNSUPPORTED and la_args is not None:   # exact-fp32 local attention / unpaired launches: separate launches below
                la_args = None
                rc = favor_bwd(None)
            _ck(rc, "sa_favor_fused_bwd")
            _favor_bracket_end(t0, "favor_fdden+fpair_states_b+fpair_b_a+fkey_fix" + ("_la" if la_args is not None else ""),
                               2.0 * B * N * G * FAVOR_FWD_FLOP_PER_HEAD_ROW(m, dh))
            local_done = la_args is not None
            sv["scan_state"] = None
        elif G > 0:
            qf, kf, Z, inv = sv["qf"], sv["kf"], sv["Z"], sv["inv"]
            dden = torch.empty(R * G, dtype=f32, device=dev)
            _ck(lib.sa_favor_dden(_ffi.ptr(dattn), _ffi.ptr(attn), inner, 0, G, dh, _ffi.ptr(inv), _ffi.ptr(dden), R * G, st), "sa_favor_dden")
            ws = self._scan_ws(B, N, G, dev)
            dqf = torch.empty_like(qf)
            dkf = torch.empty_like(kf)
            shared_dv = False
            if Z is None:   # forward ran the fused form: the cumulative terms are rebuilt inside the scans as well
                kept = sv.get("scan_state")   # the forward's chunk states: same (a = k', b = v) -> no state / prefix passes for dq'
                _ck(lib.sa_favor_scan_b_cum(_ffi.ptr(kf), _ffi.ptr(v), qs, 0, None, _ffi.ptr(dattn), inner, 0, _ffi.ptr(inv), _ffi.ptr(dqf), _ffi.ptr(dden), 1, 1e-6,
                                            B, N, G, LDF, dh, 0, _ffi.ptr(kept if kept is not None else ws), (1 if kept is not None else 0) | self._xf, st),
                    "sa_favor_scan_b_cum(dq')")
                sv["scan_state"] = None
                _ck(lib.sa_favor_scan_b_cum(_ffi.ptr(qf), _ffi.ptr(dattn), inner, 0, _ffi.ptr(inv), _ffi.ptr(v), qs, 0, None, _ffi.ptr(dkf), _ffi.ptr(dden), 2, 0.0,
                                            B, N, G, LDF, dh, 1, _ffi.ptr(ws), self._xf, st), "sa_favor_scan_b_cum(dk')")
                shared_dv = True
            else:
                _ck(lib.sa_favor_scan_b(_ffi.ptr(kf), _ffi.ptr(v), qs, 0, None, _ffi.ptr(dattn), inner, 0, _ffi.ptr(inv), _ffi.ptr(dqf), _ffi.ptr(dden), _ffi.ptr(Z),
                                        1e-6, B, N, G, LDF, dh, 0, _ffi.ptr(ws), st), "sa_favor_scan_b(dq')")
                rr = torch.empty_like(qf)
                _ck(lib.sa_cumsum_rows(_ffi.ptr(qf), _ffi.ptr(dden), _ffi.ptr(rr), B, N, G, LDF, 1, _ffi.ptr(ws), st), "sa_cumsum_rows(rev)")
                _ck(lib.sa_favor_scan_b(_ffi.ptr(qf), _ffi.ptr(dattn), inner, 0, _ffi.ptr(inv), _ffi.ptr(v), qs, 0, None, _ffi.ptr(dkf), None, _ffi.ptr(rr), 0.0,
                                        B, N, G, LDF, dh, 1, _ffi.ptr(ws), st), "sa_favor_scan_b(dk')")
            if shared_dv:   # dv runs on the states the dk' scan just built (same a = q', b = d attn * inv, reversed)
                _ck(lib.sa_favor_scan_a_state(_ffi.ptr(qf), _ffi.ptr(kf), _ffi.ptr(dattn), inner, 0, _ffi.ptr(inv), _ffi.ptr(dv), qs, 0, None, B, N, G, LDF, dh, 1, 0,
                                              _ffi.ptr(ws), 3 | self._xf, st), "sa_favor_scan_a_state(dv)")
            else:
                _ck(lib.sa_favor_scan_a(_ffi.ptr(qf), _ffi.ptr(kf), _ffi.ptr(dattn), inner, 0, _ffi.ptr(inv), _ffi.ptr(dv), qs, 0, None, B, N, G, LDF, dh, 1, 0,
                                        _ffi.ptr(ws), st), "sa_favor_scan_a(dv)")
            pop = self._proj_op()
            tsum = torch.empty(R * G, dtype=f32, device=dev)
            if self._xf:   # fp32 parity mode: feature-map backward, then the exact-fp32 dgrad GEMM with the -|x|^2 part as addend
                dddq, dddk = torch.empty_like(qf), torch.empty_like(kf)
                dqg = torch.empty(R, G * dh, dtype=f32, device=dev)
                dkg = torch.empty(R, G * dh, dtype=f32, device=dev)
                _ck(lib.sa_favor_features_bwd(_ffi.ptr(dqf), _ffi.ptr(qf), _ffi.ptr(sv["ddq"]), _ffi.ptr(sv["qg"]), G * dh, 0, G, dh, 1, _ffi.ptr(dddq), _ffi.ptr(dqg),
                                              None, None, R * G, m, LDF, st), "favor_features_bwd(q)")
                _ck(lib.sa_favor_features_bwd(_ffi.ptr(dkf), _ffi.ptr(kf), _ffi.ptr(sv["ddk"]), _ffi.ptr(sv["kg"]), G * dh, 0, G, dh, 0, _ffi.ptr(dddk), _ffi.ptr(dkg),
                                              _ffi.ptr(sv["gws"]), _ffi.ptr(tsum), R * G, m, LDF, st), "favor_features_bwd(k)")
                rg = (1, 1, R * G)
                dqg = pop.dgrad(dddq.view(1, 1, 1, R * G, LDF), rg, addend=dqg.view(1, 1, 1, R * G, dh), fwd_out_stride=LDF).view(R, G * dh)
                dkg = pop.dgrad(dddk.view(1, 1, 1, R * G, LDF), rg, addend=dkg.view(1, 1, 1, R * G, dh), fwd_out_stride=LDF).view(R, G * dh)
                dq[:, : G * dh] = dqg
                dk[:, : G * dh] = dkg
            else:          # throughput mode: one launch per side straight into the global-head columns of dq / dk (d loss / d dd never exists)
                ps = self._pop[2]
                _ck(lib.sa_favor_features_project_bwd(_ffi.ptr(dqf), _ffi.ptr(qf), _ffi.ptr(sv["ddq"]), _ffi.ptr(sv["qg"]), qs, G, _ffi.ptr(ps), 1, _ffi.ptr(dq),
                                                      None, None, R * G, m, LDF, dh, st), "sa_favor_features_project_bwd(q)")
                _ck(lib.sa_favor_features_project_bwd(_ffi.ptr(dkf), _ffi.ptr(kf), _ffi.ptr(sv["ddk"]), _ffi.ptr(sv["kg"]), qs, G, _ffi.ptr(ps), 0, _ffi.ptr(dk),
                                                      _ffi.ptr(sv["gws"]), _ffi.ptr(tsum), R * G, m, LDF, dh, st), "sa_favor_features_project_bwd(k)")
        if L > 0:
            cosb, sinb = self._rot_tables(N, dev)
            if not local_done:
                dqkr = torch.empty(2, R, L * dh, dtype=f32, device=dev)
                Db = torch.empty(R * L, dtype=f32, device=dev)
                _ck(lib.sa_local_attn_bwd(_ffi.ptr(sv["qr"]), L * dh, 0, _ffi.ptr(sv["kr"]), L * dh, 0, _ffi.ptr(v), qs, G * dh, _ffi.ptr(attn), _ffi.ptr(dattn),
                                          inner, G * dh, _ffi.ptr(sv["lse"]), _ffi.ptr(dqkr[0]), _ffi.ptr(dqkr[1]), _ffi.ptr(dv), _ffi.ptr(Db), B, N, L, self.W, dh,
                                          _ffi.ptr(dv_lp), st), "sa_local_attn_bwd")
            dqr, dkr = dqkr[0], dqkr[1]
            if fused_qkv:   # dq | dk are column blocks of one matrix: one launch
                _ck(lib.sa_rotary_groups(_ffi.ptr(dqkr), L * dh, 0, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(dq), qs, G * dh, N, R, 1, 0, 2, R * L * dh, inner, _ffi.ptr(dq_lp), st),
                    "sa_rotary_groups^T(q|k)")
            else:
                _ck(lib.sa_rotary(_ffi.ptr(dqr), L * dh, 0, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(dq), qs, G * dh, N, R, 1, 0, st), "sa_rotary^T(q)")
                _ck(lib.sa_rotary(_ffi.ptr(dkr), L * dh, 0, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(dk), qs, G * dh, N, R, 1, 0, st), "sa_rotary^T(k)")
        self._rotate_global(dq, dk, B, N, 1)
        xaT = _as5(sv["xaT"])
        base = _as5(dx1) if self.rezero else None
        if fused_qkv:
            dqkvT = _as5(dqkv_lp if dqkv_lp is not None else _cast(dqkv, T))
            gbufs = [gc.buf(sa.to_q.weight), gc.buf(sa.to_k.weight), gc.buf(sa.to_v.weight)]
            gw = self._stacked([t.view(inner, self.dim) for t in gbufs])
            if gw is not None:
                gc.wgrad(ops["to_qkv"], xaT, dqkvT, gw.view(3 * inner, self.dim, 1, 1, 1), None)
            else:   # gradient buffers are not adjacent: through a scratch matrix
                tmp = torch.zeros(3 * inner, self.dim, 1, 1, 1, dtype=f32, device=dev)
                ops["to_qkv"].wgrad(xaT, dqkvT, tmp, None)
                for i, t in enumerate(gbufs):
                    t.view(inner, self.dim).add_(tmp.view(3 * inner, self.dim)[i * inner:(i + 1) * inner])
            gc.done(sa.to_q.weight, sa.to_k.weight, sa.to_v.weight)
            dxa = ops["to_qkv"].dgrad(dqkvT, r5, addend=base, out_dtype=f32).view(R, self.dim)
        else:
            dqT, dkT, dvT = (_as5(_cast(t, T)) for t in (dq, dk, dv))
            for nm, g_ in (("to_q", dqT), ("to_k", dkT), ("to_v", dvT)):
                mod = getattr(sa, nm)
                ops[nm].wgrad(xaT, g_, gc.buf(mod.weight), gc.buf(mod.bias))
                gc.done(mod.weight, mod.bias)
            dxa = ops["to_q"].dgrad(dqT, r5, addend=base, out_dtype=f32)
            dxa = ops["to_k"].dgrad(dkT, r5, addend=dxa, out_dtype=f32)
            dxa = ops["to_v"].dgrad(dvT, r5, addend=dxa, out_dtype=f32).view(R, self.dim)
        return self._pre_bwd(self.aw, dxa, dx1, sv["x"], sv["st_a"], R, gc)


class _StackChain:
    def __init__(self, base: BasePerformer, dim, dtype):
        self.base, self.dtype = base, dtype
        self.layers = [_LayerEngine(l[0], l[1], dim, dtype, base.use_rezero) for l in base.net.layers]
        self.grad_sink = None

    def params(self):
        return [p for p in self.base.parameters()]

    def invalidate(self):
        for l in self.layers:
            l.invalidate()

    def forward(self, x, record):
        B, N, D = x.shape
        x = x.reshape(B * N, D).float().contiguous()
        tape = [] if record else None
        x_lp = None
        for l in self.layers:
            x = l.fwd(x, B, N, tape, x_lp)
            x_lp = l.out_lp
        return x.view(B, N, D), tape

    def backward(self, dy, tape):
        B, N, D = dy.shape
        # throughput mode: weight gradients on a second stream (DDP: _GradCtx.done reports a bucket's parameters from that stream)
        side = None
        if (self.dtype != torch.float32 and not debug.host("no_side_wgrad") and not debug.deterministic()
                and not torch.cuda.is_current_stream_capturing()):
            side = _SideWgrad(dy.device)
        gc = _GradCtx(self.grad_sink, side)
        g = dy.reshape(B * N, D).float().contiguous()
        for l, sv in zip(reversed(self.layers), reversed(tape)):
            g = l.bwd(g, sv, B, N, gc)
        if side is not None:
            side.join()
        return g.view(B, N, D), gc.grads


class _StackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, chain: _StackChain, record, x, *params):
        _ffi.require_gpu()
        y, tape = chain.forward(x, record)
        ctx.chain, ctx.tape = chain, tape
        return y

    @staticmethod
    def backward(ctx, gy):
        dx, grads = ctx.chain.backward(gy, ctx.tape)
        ctx.tape = None
        return (None, None, dx, *[grads.get(p) for p in ctx.chain.params()])


class _EmbedFn(torch.autograd.Function):
    """x[b,n,:] = sum of embedding rows (token, spatial x3 with a zero at position 0, absolute position) -- performer.py:241-266"""

    @staticmethod
    def forward(ctx, tables: Sequence[torch.Tensor], idx: Sequence[torch.Tensor], per_pos: Sequence[int], B, N, *params):
        _ffi.require_gpu()
        dim = tables[0].shape[1]
        out = torch.empty(B, N, dim, dtype=torch.float32, device=tables[0].device)
        n = len(tables)
        tp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tables])
        ip = (ctypes.c_void_p * n)(*[i.data_ptr() for i in idx])
        pp = (ctypes.c_int32 * n)(*per_pos)
        _ck(_ffi.lib().sa_embed_sum(n, tp, ip, pp, dim, N, B * N, _ffi.ptr(out), _ffi.stream()), "sa_embed_sum")
        ctx.idx, ctx.per_pos, ctx.shapes, ctx.BN = list(idx), list(per_pos), [t.shape for t in tables], (B, N)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, N = ctx.BN
        dy = dy.contiguous()
        grads = []
        for t, (ix, pp, shp) in enumerate(zip(ctx.idx, ctx.per_pos, ctx.shapes)):
            if not ctx.needs_input_grad[5 + t]:      # fixed (sinusoidal) spatial tables are buffers
                grads.append(None)
                continue
            g = torch.zeros(shp, dtype=torch.float32, device=dy.device)
            if debug.deterministic():
                _ck(_ffi.lib().sa_embed_scatter_det(_ffi.ptr(dy), _ffi.ptr(g), _ffi.ptr(ix), pp, shp[1], N, B * N, shp[0], _ffi.stream()), "sa_embed_scatter_det")
            else:
                _ck(_ffi.lib().sa_embed_scatter(_ffi.ptr(dy), _ffi.ptr(g), _ffi.ptr(ix), pp, shp[1], N, B * N, _ffi.stream()), "sa_embed_scatter")
            grads.append(g)
        return (None, None, None, None, None, *grads)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        _ffi.require_gpu()
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float().contiguous()
        R, C = x2.shape
        y = torch.empty_like(x2)
        stats = torch.empty(2 * R, dtype=torch.float32, device=x.device)
        _ck(_ffi.lib().sa_layernorm_fwd(_ffi.ptr(x2), _ffi.ptr(w), _ffi.ptr(b), _ffi.ptr(y), None, 0, _ffi.ptr(stats), R, C, eps, _ffi.stream()), "sa_layernorm_fwd")
        ctx.save_for_backward(x2, w, stats)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, w, stats = ctx.saved_tensors
        R, C = x2.shape
        d = dy.reshape(R, C).float().contiguous()
        dx, dw, db = torch.empty_like(x2), torch.zeros_like(w), torch.zeros_like(w)
        _ck(_ffi.lib().sa_layernorm_bwd(_ffi.ptr(d), _ffi.ptr(x2), _ffi.ptr(w), _ffi.ptr(stats), _ffi.ptr(dx), _ffi.ptr(dw), _ffi.ptr(db), R, C, _ffi.stream()),
            "sa_layernorm_bwd")
        if debug.deterministic():
            # --deterministic: the weight / bias gradients again as fixed-order column sums (the kernel above accumulated them with one atomic per column and block)
            from ...engine import colsum_det
            prod = torch.empty_like(x2)
            _ck(_ffi.lib().sa_layernorm_dwprod(_ffi.ptr(d), _ffi.ptr(x2), _ffi.ptr(stats), _ffi.ptr(prod), R, C, _ffi.stream()), "sa_layernorm_dwprod")
            dw, db = torch.zeros_like(w), torch.zeros_like(w)
            colsum_det(prod, C, dw)
            colsum_det(d, C, db)
        return dx.view(dy.shape), dw, db, None


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b through the 1-tap implicit-GEMM kernels (fp32 in / fp32 out; GEMM in the op's compute dtype)."""

    @staticmethod
    def forward(ctx, op: ConvOp, x, w, b):
        _ffi.require_gpu()
        op.weight, op.bias = w, b
        shp = x.shape
        x2 = _cast(x.reshape(-1, shp[-1]).float().contiguous(), op.dtype)
        y = op.fprop(_as5(x2), out_dtype=torch.float32).view(*shp[:-1], w.shape[0])
        ctx.op, ctx.x2, ctx.has_b = op, x2, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        op, x2 = ctx.op, ctx.x2
        R = x2.shape[0]
        from ...engine import cast_pad, vec_of
        d2 = dy.reshape(R, -1).float().contiguous()
        g = _as5(cast_pad(d2, op.dtype, _ru(d2.shape[1], vec_of(op.dtype))))  # channel stride padded to a 16-byte multiple
        dw = torch.zeros_like(op.weight)
        db = torch.zeros_like(op.bias) if ctx.has_b else None
        op.wgrad(_as5(x2), g, dw, db)
        dx = op.dgrad(g, (1, 1, R), out_dtype=torch.float32).view(*dy.shape[:-1], x2.shape[1])
        return None, dx, dw, db


class _TiedOut:
    """`tie_embed=True`: the logits are x @ token_emb.weight^T (performer.py:288) -- the token table seen as a bias-free nn.Linear(dim, num_tokens)"""

    def __init__(self, emb: nn.Embedding):
        self._emb = emb
        self.bias = None

    @property
    def weight(self):
        return self._emb.weight

    in_features = property(lambda self: self._emb.weight.shape[1])
    out_features = property(lambda self: self._emb.weight.shape[0])


# ------------------------------------------------------------------------------------------------ the plugin class
class Performer(TransformerBase):
    """NOTE: all tensor logic assumes the ordering [Batch, Length, Channel] (as the reference)."""

    def __init__(
        self,
        *,
        num_tokens: int,
        max_seq_len: int,
        dim: int,
        depth: int,
        heads: int,
        ordering: Ordering,
        dim_head: int = 64,
        local_attn_heads: int = 0,
        local_window_size: int = 256,
        causal: bool = True,
        ff_mult: int = 4,
        nb_features: Optional[int] = None,
        feature_redraw_interval: int = 1000,
        reversible: bool = False,
        ff_chunks: int = 1,
        ff_glu: bool = False,
        emb_dropout: float = 0.0,
        ff_dropout: float = 0.0,
        attn_dropout: float = 0.0,
        generalized_attention: bool = False,
        kernel_fn: torch.nn.Module = nn.ReLU(),
        use_scalenorm: bool = False,
        use_rezero: bool = False,
        cross_attend: bool = False,
        no_projection: bool = False,
        tie_embed: bool = False,
        rotary_position_emb: bool = False,
        fixed_position_emb: bool = False,
        axial_position_emb: bool = False,
        axial_position_shape: Tuple[int, int] = None,
        auto_check_redraw: bool = True,
        qkv_bias: bool = False,
        attn_out_bias: bool = False,
        spatial_position_emb: str = None,
        spatial_shape: Union[Tuple[int, int], Tuple[int, int, int]] = None,
        conditioning_num_tokens: Optional[Tuple[int, ...]] = None,
        conditioning_type: str = TransformerConditioningType.NONE.value,
        compute_dtype: torch.dtype = torch.float32,
    ):
        super().__init__()
        assert 0 <= sum([rotary_position_emb, fixed_position_emb, axial_position_emb]) <= 1, (
            f"rotary_position_emb, fixed_position_emb and axial_position_emb are exclusive, but received "
            f"{rotary_position_emb} {fixed_position_emb} and {axial_position_emb}."
        )
        # accounting for the number of prepended conditionings (performer.py:119-125)
        self.max_seq_len = max_seq_len + (len(conditioning_num_tokens)
                                          if conditioning_num_tokens and conditioning_type == TransformerConditioningType.PREPENDING.value else 0)
        self.token_emb = nn.Embedding(num_tokens, dim)
        # performer.py:134-147: rotary = the sinusoidal table on x AND a dim_head-wide one handed to every attention layer (q / k of the global heads are rotated);
        # fixed = the sinusoidal table (a buffer); axial = two learned axis tables; else the learned absolute table
        self.layer_pos_emb = None
        if rotary_position_emb:
            self.pos_emb = FixedPositionalEmbedding(dim, self.max_seq_len)
            self.layer_pos_emb = FixedPositionalEmbedding(dim_head, self.max_seq_len)
        elif fixed_position_emb:
            self.pos_emb = FixedPositionalEmbedding(dim, self.max_seq_len)
        elif axial_position_emb:
            axial_position_shape = axial_position_shape if axial_position_shape is not None else (math.ceil(self.max_seq_len / 64), 64)
            self.pos_emb = AxialPositionalEmbedding(dim, axial_position_shape)
        else:
            self.pos_emb = AbsolutePositionalEmbedding(dim, self.max_seq_len)
        self.ordering = ordering
        self.spatial_position_emb = nn.ModuleList()
        if spatial_position_emb:
            assert spatial_position_emb in ["fixed", "absolute"], (
                f"spatial_position_emb must be either 'fixed' or  'absolute', but got {spatial_position_emb}")
            coords = np.array(np.meshgrid(*tuple(np.arange(0, s) for s in spatial_shape), indexing="ij"))
            for axis in range(len(spatial_shape)):
                seq = self.ordering(torch.from_numpy(coords[axis, ...].flatten()))
                cls = FixedSpatialPositionalEmbedding if spatial_position_emb == "fixed" else AbsoluteSpatialPositionalEmbedding
                self.spatial_position_emb.append(cls(dim=dim, spatial_indices_sequence=seq))
        self.conditioning_emb = nn.ModuleList()
        self.conditioning_type = conditioning_type
        if conditioning_num_tokens:
            for cnt in conditioning_num_tokens:
                self.conditioning_emb.append(nn.Embedding(cnt, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.performer = BasePerformer(dim, depth, heads, dim_head, local_attn_heads, local_window_size, causal, ff_mult, nb_features,
                                       feature_redraw_interval, reversible, ff_chunks, generalized_attention, kernel_fn, use_scalenorm, use_rezero,
                                       ff_glu, ff_dropout, attn_dropout, cross_attend, no_projection, auto_check_redraw, qkv_bias, attn_out_bias)
        self.norm = nn.LayerNorm(dim)
        self.to_out = nn.Linear(dim, num_tokens) if not tie_embed else None       # performer.py:222, 286-288: tied = x @ token_emb.weight^T, no bias
        self.dim, self.compute_dtype = dim, compute_dtype
        self._chain = _StackChain(self.performer, dim, compute_dtype)
        if self.layer_pos_emb is not None:
            for l in self._chain.layers:
                l.layer_pos = self.layer_pos_emb
        self._out_op = _lin(self._out_mod(), compute_dtype)
        self._idx_cache = {}

    def _out_mod(self):
        """the vocabulary projection as a (weight, bias) holder: ``to_out``, or the token table itself when the embeddings are tied"""
        if self.to_out is not None:
            return self.to_out
        tied = self.__dict__.get("_tied_out")
        if tied is None:
            tied = self.__dict__["_tied_out"] = _TiedOut(self.token_emb)
        return tied

    def check_redraw_projections(self):
        self.performer.check_redraw_projections()

    def fix_projection_matrices_(self):
        self.performer.fix_projection_matrices_()

    def set_grad_sink(self, sink):
        """Route weight gradients into a ``runtime.ddp.GradReducer``.  The layer stack reports each parameter as its gradient kernels are queued;
        the parameters outside it (embeddings, final LayerNorm, vocabulary projection) get their gradients from autograd, which accumulates
        them into the flat buffer, so they report from a post-accumulate hook -- otherwise the bucket holding the LAST parameters (the first
        one backward completes) would only be reduced in ``finish()`` with nothing left to overlap."""
        self._chain.grad_sink = sink
        for h in getattr(self, "_sink_hooks", []):
            h.remove()
        self._sink_hooks = []
        if sink is not None:
            in_chain = {id(p) for p in self._chain.params()}
            for p in self.parameters():
                if p.requires_grad and id(p) not in in_chain:
                    self._sink_hooks.append(p.register_post_accumulate_grad_hook(lambda q, s=sink: s.ready(q)))

    def invalidate_packed_weights(self):
        self._chain.invalidate()
        self._out_op.invalidate()
        # every dense layer's forward / data-gradient operand again, in one launch
        if getattr(self, "_packset", None) is None:
            self._packset = PackSet()
        self._packset.repack([op for l in self._chain.layers for op in l.ops.values()] + [self._out_op])

    def range_repacker(self, flat):
        """For ``FusedAdam(in_backward=reducer)``: ``opt.on_range.append(r); opt.on_step.append(r.finish)`` INSTEAD of ``invalidate_packed_weights`` --
        a layer's operands are re-packed right behind the optimizer slice of its bucket, in the shadow of the backward pass."""
        from ...engine import RangeRepacker
        return RangeRepacker(flat, lambda: [op for l in self._chain.layers for op in l.ops.values()] + [self._out_op])

    # ------------------------------------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample(self, prefix: torch.Tensor, conditioning: torch.Tensor = None, temperature: float = 1.0, sample: bool = True, top_k: Optional[int] = None,
               stateful: Optional[bool] = None, use_graph: bool = True) -> torch.Tensor:
        """TransformerBase.sample (transformer.py:58-101).  ``stateful=False`` is the reference-faithful O(N^2) loop (a full forward over the
        growing prefix per token); ``stateful=True`` (default without conditioning) carries the FAVOR+ running sums and the local-attention
        key/value caches from token to token -- O(N), same logits up to fp32 rounding -- and replays one captured
        HIP graph per token."""
        bos = self.conditioning_type == TransformerConditioningType.BOSREPLACEMENT.value
        if stateful is None:     # O(N) decoding unless the conditioning lengthens the sequence (prepending: the reference-faithful loop)
            stateful = (conditioning is None or bos) and self.layer_pos_emb is None
        if stateful and self.layer_pos_emb is not None:
            raise NotImplementedError("stateful (O(N)) sampling does not rotate the global heads' q / k (rotary_position_emb=True): use stateful=False")
        if not stateful:
            return super().sample(prefix, conditioning=conditioning, temperature=temperature, sample=sample, top_k=top_k)
        assert conditioning is None or bos, "stateful sampling takes BOS-replacement conditionings only (use stateful=False)"
        return self._sample_stateful(prefix, temperature, sample, top_k, use_graph, conditioning)

    def _sample_stateful(self, prefix, temperature, sample, top_k, use_graph, conditioning=None):
        _ffi.require_gpu()
        self.eval()
        dev = self.token_emb.weight.device
        lib = _ffi.lib()
        B, P = prefix.shape
        steps = int(np.prod(self.ordering.dimensions))
        total = P + steps                      # tokens in the final sequence; positions 0 .. total-2 are fed through the network
        npos = total - 1
        assert npos <= self.max_seq_len, f"sequence length {npos} must be less than the max sequence length {self.max_seq_len}"
        seq = torch.zeros(B, total, dtype=torch.int64, device=dev)
        seq[:, :P] = prefix.to(dev).long()
        seq0 = seq.clone()
        posbuf = torch.zeros(2, dtype=torch.int32, device=dev)     # [position, ticket word of sa_sample_step]
        pos, ticket = posbuf[:1], posbuf[1:]
        tok = torch.zeros(B, dtype=torch.int64, device=dev)
        pidx, sp = self._position_indices(npos, dev)
        tok_table = self.token_emb.weight
        if conditioning:
            # BOS replacement (performer.py:252-261): position 0 carries the summed conditioning embeddings instead of its token (+ spatial, which are zero
            # there) embedding -> B extra rows behind the token table, and position 0 of sequence b points at row num_tokens + b
            c = sum(emb(conditioning[i].to(dev))[:, 0, :] for i, emb in enumerate(self.conditioning_emb))
            tok_table = torch.cat((self.token_emb.weight.detach(), c.to(self.token_emb.weight.dtype)), dim=0).contiguous()
            seq[:, 0] = tok_table.shape[0] - B + torch.arange(B, device=dev)
            seq0 = seq.clone()
        ptabs, pidxs = self._pos_tables(pidx)
        tables = [tok_table] + [(m.emb if isinstance(m, FixedSpatialPositionalEmbedding) else m.emb.weight) for m in self.spatial_position_emb] + ptabs
        idx = [tok] + sp + pidxs
        per_pos = [0] + [1] * len(sp) + [1] * len(ptabs)
        n = len(tables)
        tp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tables])
        ip = (ctypes.c_void_p * n)(*[i.data_ptr() for i in idx])
        pp = (ctypes.c_int32 * n)(*per_pos)
        dim = tables[0].shape[1]
        layers = self._chain.layers
        states = [l.new_state(B, npos, dev) for l in layers]
        for l in layers:
            l.__dict__.pop("_dec_w", None)     # bf16 decode copies of the parameters: rebuilt from the current values by the warm-up step
        col = torch.arange(total, device=dev)[None, :]
        tok.copy_(seq0[:, 0])

        fused_tail = not debug.host("no_sample_step")     # one launch for the decision (incl. the top-k cut) + sequence update (sa_sample_step)
        n_vocab = self._out_mod().weight.shape[0]
        if top_k is not None and not 0 < int(top_k) <= n_vocab:   # what torch.topk of the reference (transformer.py:14) raises on
            raise RuntimeError(f"selected index k out of range (top_k = {top_k}, vocabulary {n_vocab})")
        # the uniforms of every step, drawn once (a torch.rand inside the captured step costs three launches per token: the generator's seed / offset fills + the draw)
        u_all = torch.rand(npos + 1, B, device=dev, dtype=torch.float32) if (fused_tail and sample) else None

        def one_step():
            if not fused_tail:
                p64 = pos.to(torch.int64)
                tok.copy_(seq.gather(1, p64.expand(B, 1)).squeeze(1))
            x = torch.empty(B, dim, dtype=torch.float32, device=dev)
            _ck(lib.sa_embed_step(n, tp, ip, pp, dim, _ffi.ptr(pos), B, _ffi.ptr(x), _ffi.stream()), "sa_embed_step")
            for l, stt in zip(layers, states):
                x = l.step(x, B, npos, pos, stt)
            h = _LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps)
            logits = torch.empty(B, n_vocab, dtype=torch.float32, device=dev)
            layers[0]._gemv(h, [self._out_mod()], logits)
            if fused_tail:
                # temperature, softmax, the draw (inverse CDF against the pre-drawn uniforms -- torch.multinomial cannot be captured in a HIP graph) or arg-max,
                # seq[:, pos + 1] (unless it belongs to the given prefix), the next step's token, pos += 1: one launch instead of ~20 small torch kernels
                _ck(lib.sa_sample_step(_ffi.ptr(logits), B, logits.shape[1], float(temperature), _ffi.ptr(u_all), B, int(bool(sample)), int(top_k or 0), _ffi.ptr(seq),
                                       total, P, _ffi.ptr(pos), _ffi.ptr(ticket), _ffi.ptr(tok), _ffi.stream()), "sa_sample_step")
                return
            logits = logits / temperature
            # transformer.py:11-17 (_top_k_logits) without the boolean-mask assignment, which cannot be captured
            if top_k is not None:
                kth = torch.topk(logits, top_k)[0][:, -1:]
                logits = torch.where(logits < kth, torch.full_like(logits, float("-inf")), logits)
            probs = torch.softmax(logits, dim=-1)
            if sample:
                # categorical draw by inverse CDF: same distribution as torch.multinomial(probs, 1) (transformer.py:37), which cannot be
                # captured in a HIP graph (hipErrorStreamCaptureUnsupported)
                u = torch.rand(B, 1, device=dev, dtype=probs.dtype)
                cdf = probs.cumsum(dim=-1)
                ix = (cdf < u * cdf[:, -1:]).sum(dim=-1, keepdim=True).clamp_(max=probs.shape[-1] - 1)
            else:
                ix = torch.topk(probs, k=1, dim=-1)[1]
            # position pos+1 receives the sampled token unless it still belongs to the given prefix
            write = (col == (p64 + 1)) & (col >= P)
            seq.copy_(torch.where(write, ix.expand(B, total), seq))
            pos.add_(1)

        graph = None
        if use_graph:
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):      # warm-up off the capture: packs weights, sizes the caches, sets kernel attributes
                    one_step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                for l, stt in zip(layers, states):
                    l.reset_state(stt)
                posbuf.zero_()
                seq.copy_(seq0)
                tok.copy_(seq0[:, 0])
                graph = torch.cuda.CUDAGraph()
                # thread_local: other threads of the process (the RCCL watchdog of a distributed run) may call HIP during the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    one_step()
            except Exception as exc:               # capture is an optimisation: fall back to eager launches of the same O(N) step
                import warnings
                warnings.warn(f"HIP graph capture of the decode step failed ({type(exc).__name__}: {exc}); running it eagerly")
                graph = None
                torch.cuda.synchronize()
                for l, stt in zip(layers, states):
                    l.reset_state(stt)
                posbuf.zero_()
                seq.copy_(seq0)
                tok.copy_(seq0[:, 0])
        for _ in range(npos):
            if graph is not None:
                graph.replay()
            else:
                one_step()
        return sequence_to_grid(seq, P, self.ordering)

    # ------------------------------------------------------------------------------------------------
    def _pos_table(self):
        return self.pos_emb.emb if isinstance(self.pos_emb, FixedPositionalEmbedding) else self.pos_emb.emb.weight

    def _pos_tables(self, pos):
        """(tables, index rows) of the positional term for the positions `pos`: one table, or the two axis tables of the axial embedding"""
        if isinstance(self.pos_emb, AxialPositionalEmbedding):
            s0, s1 = self.pos_emb.shape
            assert int(pos.numel()) <= s0 * s1, f"sequence length {int(pos.numel())} exceeds the axial grid {s0} x {s1}"
            return ([self.pos_emb.weights_0.view(s0, self.dim), self.pos_emb.weights_1.view(s1, self.dim)],
                    [torch.div(pos, s1, rounding_mode="floor"), pos % s1])
        return [self._pos_table()], [pos]

    def _pos_rows(self, n, dev):
        """the positional term of positions 0 .. n-1 as a dense [n, dim] tensor (autograd-visible): the conditioning paths add single rows of it"""
        tabs, idx = self._pos_tables(torch.arange(n, device=dev, dtype=torch.int64))
        return sum(t[i] for t, i in zip(tabs, idx))

    def _position_indices(self, n, dev):
        key = (n, str(dev))
        if key not in self._idx_cache:
            pos = torch.arange(n, device=dev, dtype=torch.int64)
            sp = []
            for mod in self.spatial_position_emb:
                ix = torch.full((n,), -1, device=dev, dtype=torch.int64)  # position 0 is zero-padded (performer.py:31,38)
                if isinstance(mod, FixedSpatialPositionalEmbedding):      # rows of the buffer are already in sequence order (performer.py:52-57)
                    cnt = min(n - 1, mod.emb.shape[0])
                    if cnt > 0:
                        ix[1:1 + cnt] = torch.arange(cnt, device=dev)
                else:
                    seq = mod.spatial_indices_sequence.to(dev).long()
                    cnt = min(n - 1, seq.numel())
                    if cnt > 0:
                        ix[1:1 + cnt] = seq[:cnt]
                sp.append(ix)
            self._idx_cache[key] = (pos, sp)
        return self._idx_cache[key]

    def forward(self, x: torch.Tensor, conditionings: Sequence[torch.Tensor] = None, return_encodings: bool = False, **kwargs):
        b, n = x.shape
        assert n <= self.max_seq_len, f"sequence length {n} must be less than the max sequence length {self.max_seq_len}"
        dev = self.token_emb.weight.device
        tok = x.to(dev).long().contiguous().view(-1)
        pos, sp = self._position_indices(n, dev)
        sp_tables = [(m.emb if isinstance(m, FixedSpatialPositionalEmbedding) else m.emb.weight) for m in self.spatial_position_emb]
        prepend = bool(conditionings) and self.conditioning_type == TransformerConditioningType.PREPENDING.value
        if prepend:
            # performer.py:262-266: token + spatial embeddings, THEN the conditioning embeddings in front (the last one ends up first), THEN the
            # absolute positional embedding over the longer sequence; the conditioning positions are cut off again after the final norm (:279-281)
            tables, idx, per_pos = [self.token_emb.weight] + sp_tables, [tok] + sp, [0] + [1] * len(sp)
            h = _EmbedFn.apply(tables, idx, per_pos, b, n, *tables)
            for i, emb in enumerate(self.conditioning_emb):
                h = torch.cat((emb(conditionings[i].to(dev)), h), dim=1)
            nt = h.shape[1]
            assert nt <= self.max_seq_len, f"sequence length {nt} must be less than the max sequence length {self.max_seq_len}"
            ptab, pix = self._pos_tables(torch.arange(nt, device=dev, dtype=torch.int64))
            h = h + _EmbedFn.apply(ptab, pix, [1] * len(ptab), b, nt, *ptab)
        else:
            ptab, pix = self._pos_tables(pos)
            tables = [self.token_emb.weight] + sp_tables + ptab
            idx = [tok] + sp + pix
            per_pos = [0] + [1] * len(sp) + [1] * len(ptab)
            h = _EmbedFn.apply(tables, idx, per_pos, b, n, *tables)
        if conditionings and self.conditioning_type == TransformerConditioningType.BOSREPLACEMENT.value:
            # performer.py:252-261: the BOS embedding (incl. its spatial terms) is REPLACED by the summed conditioning embeddings,
            # the absolute positional embedding is added afterwards
            c = sum(emb(conditionings[i].to(dev))[:, 0, :] for i, emb in enumerate(self.conditioning_emb))
            first = c + self._pos_rows(1, dev)[0]
            h = torch.cat((first[:, None, :], h[:, 1:, :]), dim=1)
        if self.dropout.p > 0.0 and self.training:
            # nn.Dropout(emb_dropout) on the summed embeddings (performer.py:201,270): elementwise keep / (1 - p) mask drawn on the device, a broadcast-free multiply
            keep = torch.bernoulli(torch.full_like(h, 1.0 - self.dropout.p)) / (1.0 - self.dropout.p)
            self._last_emb_mask = keep          # (tests replay it through the oracle)
            h = h * keep
        if self.performer.auto_check_redraw:
            self.performer.proj_updater.redraw_projections()
        params = self._chain.params()
        record = torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in params))
        h = _StackFn.apply(self._chain, record, h, *params)
        h = _LayerNormFn.apply(h, self.norm.weight, self.norm.bias, self.norm.eps)
        if prepend:
            h = h[:, len(conditionings):, :]
        if return_encodings:
            return h
        om = self._out_mod()
        return _LinearFn.apply(self._out_op, h, om.weight, om.bias)
