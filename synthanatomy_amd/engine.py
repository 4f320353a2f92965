"""Host-side launch planning for the implicit-GEMM convolution family (csrc/conv_fprop.hip, conv_wgrad.hip).

A ``ConvOp`` owns the packed GEMM operands of one reference layer (``nn.Conv3d`` / ``nn.ConvTranspose3d`` /
``nn.Linear``; reference src/networks/vqvae/baseline.py:153-160,218-244,258-293) and turns forward, data-gradient and
weight-gradient requests into ``sa_conv_geom`` launches.  Activations are channels-last ``[N, D, H, W, C]`` tensors of
the compute dtype (fp32 = exact-f32 MFMA parity mode, bf16 = throughput mode).

Everything here runs on the HIP device through the C ABI; there is no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _ffi, debug
from ._ffi import ACT_NONE, MASK_NONE, ConvGeom, Epilogue


class KernelTimer:
    """Optional live per-kernel timing with HIP events on the launch stream (used by bench.py's roofline leg).

    Keyed by the kernel INSTANCE (same granularity as rocprofv3's kernel names), accumulating launches, algorithmic
    FLOPs and -- after ``collect()`` -- device time."""

    def __init__(self):
        self.pending = []   # (key, flops, ev0, ev1, bytes[, operand bytes])
        self.stats = {}     # key -> [launches, flops, ms, algorithmic bytes of the launches reported against HBM, operand bytes of the MFMA launches]

    def wrap(self, key, flops, fn, nbytes=0.0, abytes=0.0):
        """nbytes: algorithmic bytes of a launch that is REPORTED against the HBM roof (one-channel layers, fused 1x1x1 backward: bench.py's roofline_hbm picks among
        the keys that carry them); abytes: operand bytes of an MFMA launch (input once, packed weights once, every output / addend / mask once) -- what its PMC
        traffic is compared with and what `frac_hbm` of the Performer's dense layers is computed from."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        if key is None or key == "+wgrad_reduce_kernel":  # convolution launches: the dispatcher says which instance it picked
            key = _ffi.lib().sa_last_conv_kernel().decode() + (key or "")
        self.pending.append((key, flops, e0, e1, nbytes, abytes))

    def collect(self):
        torch.cuda.synchronize()
        for ent in self.pending:
            key, flops, e0, e1, nbytes = ent[:5]
            st = self.stats.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0])
            st[0] += 1
            st[1] += flops
            st[2] += e0.elapsed_time(e1)
            st[3] += nbytes
            st[4] += ent[5] if len(ent) > 5 else 0.0
        self.pending = []
        return self.stats


TIMER: Optional[KernelTimer] = None  # set by bench.py around the timed region


def _geom_flops(g) -> float:
    m = g.N * g.Dm * g.Hm * g.Wm
    return 2.0 * m * (g.KT[0] * g.KT[1] * g.KT[2]) * g.cin_valid * g.cout_valid


def _launch(key, flops, fn, nbytes=0.0, abytes=0.0):
    if TIMER is not None:
        TIMER.wrap(key, flops, fn, nbytes, abytes)
    else:
        fn()


def _tbytes(*ts) -> float:
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def vec_of(dtype: torch.dtype) -> int:
    return 4 if dtype == torch.float32 else 8


def _i3(v):
    return (ctypes.c_int32 * 3)(*[int(a) for a in v])


class _Plan:
    """One kernel launch: geometry + how to pack (or scatter, for wgrad) the weight operand."""

    __slots__ = ("geom", "rows", "red", "ntaps", "lut", "s_row", "s_red", "wpk", "lut_c", "packed_ver")

    def __init__(self, geom, rows, red, ntaps, lut, s_row, s_red):
        self.geom, self.rows, self.red, self.ntaps, self.lut, self.s_row, self.s_red = geom, rows, red, ntaps, lut, s_row, s_red
        self.wpk = None
        self.packed_ver = None
        self.lut_c = (ctypes.c_int32 * len(lut))(*lut) if lut is not None else None


def make_geom(dtype, N, grid, idims, cin_s, odims, cout_s, cin_valid, cout_valid, KT, in_mult, tap_step, in_off, out_mult, out_off) -> ConvGeom:
    g = ConvGeom()
    g.N, (g.Dm, g.Hm, g.Wm) = N, grid
    g.Di, g.Hi, g.Wi = idims
    g.Cin = cin_s
    g.Do, g.Ho, g.Wo = odims
    g.Cout = cout_s
    g.cin_valid, g.cout_valid = cin_valid, cout_valid
    g.KT = _i3(KT)
    g.in_mult, g.tap_step, g.in_off = _i3(in_mult), _i3(tap_step), _i3(in_off)
    g.out_mult, g.out_off = _i3(out_mult), _i3(out_off)
    bke = 32 if dtype == torch.float32 else 64
    g.Kpad = _ru(KT[0] * KT[1] * KT[2] * cin_s, bke)
    g.CoutPad = _ru(cout_valid, 128)
    return g


def _parities():
    return [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]


def _parity_lut(par, k=4):
    k0 = [(p + 1) % 2 for p in par]
    return [((k0[0] + 2 * td) * k + (k0[1] + 2 * th)) * k + (k0[2] + 2 * tw) for td in (0, 1) for th in (0, 1) for tw in (0, 1)]


class ConvOp:
    """kind="conv": weight [Cout, Cin, k,k,k];  kind="convT": weight [Cin, Cout, k,k,k] (k4 s2 p1 only)."""

    def __init__(self, kind: str, cin: int, cout: int, k: int, stride: int, pad: int, weight: torch.Tensor, bias: Optional[torch.Tensor],
                 dtype: torch.dtype, w_strides: Optional[Tuple[int, int]] = None, fwd_dtype: Optional[torch.dtype] = None):
        """`w_strides` = (stride of cout, stride of cin) in elements of `weight` for a 1x1x1 "conv" whose weight is stored transposed
        (the 64 taps of the final ConvTranspose3d as output channels): forward and wgrad only.
        `fwd_dtype` (torch.float16 with dtype = torch.bfloat16): the FORWARD launches take f16 activations and an f16-packed operand and write f16
        (+ a bf16 copy on request, `want_lp`); data and weight gradients stay in `dtype` -- the reference's AMP forward (src/engines/trainer.py:161-163)."""
        assert kind in ("conv", "convT")
        assert w_strides is None or (kind == "conv" and k == 1)
        self.w_strides = w_strides
        if kind == "convT" and not (k == 4 and stride == 2 and pad == 1):
            raise NotImplementedError("ConvTranspose3d is implemented for kernel 4 / stride 2 / padding 1 / output_padding 0 (the reference's setting)")
        self.kind, self.cin, self.cout, self.k, self.stride, self.pad = kind, cin, cout, k, stride, pad
        self.weight, self.bias = weight, bias
        self.dtype = dtype
        self.fwd_dtype = fwd_dtype or dtype
        assert self.fwd_dtype == dtype or (self.fwd_dtype == torch.float16 and dtype == torch.bfloat16), (self.fwd_dtype, dtype)
        self.vec = vec_of(dtype)
        self.T = k ** 3
        self._plans = {}
        self._packs = {}
        self._bias_pad = None
        self._epoch = 0

    # ------------------------------------------------------------------ shapes
    def out_dims(self, idims):
        if self.kind == "conv":
            return tuple((d + 2 * self.pad - self.k) // self.stride + 1 for d in idims)
        return tuple(2 * d for d in idims)

    def cs_in(self):
        return _ru(self.cin, self.vec)

    # ------------------------------------------------------------------ plans
    def _get_plans(self, N: int, idims: Tuple[int, int, int], cout_s: int, gout_s: int):
        key = (N, tuple(idims), cout_s, gout_s)
        if key in self._plans:
            return self._plans[key]
        dt, k, s, p, T = self.dtype, self.k, self.stride, self.pad, self.T
        cin, cout = self.cin, self.cout
        cin_s = self.cs_in()
        odims = self.out_dims(idims)
        fwd: List[_Plan] = []
        dgr: List[_Plan] = []
        wgr: List[_Plan] = []
        one, zero = (1, 1, 1), (0, 0, 0)
        if self.kind == "conv":
            sr, sd = self.w_strides if self.w_strides is not None else (cin * T, T)
            g = make_geom(dt, N, odims, idims, cin_s, odims, cout_s, cin, cout, (k,) * 3, (s,) * 3, one, (-p,) * 3, one, zero)
            fwd.append(_Plan(g, cout, cin, T, None, sr, sd))
            gw = make_geom(dt, N, odims, idims, cin_s, odims, gout_s, cin, cout, (k,) * 3, (s,) * 3, one, (-p,) * 3, one, zero)
            wgr.append(_Plan(gw, cout, cin, T, None, sr, sd))
            if self.w_strides is not None:
                dgr = None
            elif s == 1:
                g = make_geom(dt, N, idims, odims, gout_s, idims, cin_s, cout, cin, (k,) * 3, one, (-1,) * 3, (p,) * 3, one, zero)
                dgr.append(_Plan(g, cin, cout, T, None, T, cin * T))
            elif s == 2 and k == 4 and p == 1 and all(d % 2 == 0 for d in idims):
                half = tuple(d // 2 for d in idims)
                for par in _parities():
                    g = make_geom(dt, N, half, odims, gout_s, idims, cin_s, cout, cin, (2, 2, 2), one, (-1,) * 3, par, (2, 2, 2), par)
                    dgr.append(_Plan(g, cin, cout, 8, _parity_lut(par), T, cin * T))
            else:
                dgr = None  # data gradient not available for this geometry
        else:  # convT k4 s2 p1
            for par in _parities():
                lut = _parity_lut(par)
                g = make_geom(dt, N, idims, idims, cin_s, odims, cout_s, cin, cout, (2, 2, 2), one, (-1,) * 3, par, (2, 2, 2), par)
                fwd.append(_Plan(g, cout, cin, 8, lut, T, cout * T))
                gw = make_geom(dt, N, idims, idims, cin_s, odims, gout_s, cin, cout, (2, 2, 2), one, (-1,) * 3, par, (2, 2, 2), par)
                wgr.append(_Plan(gw, cout, cin, 8, lut, T, cout * T))
            g = make_geom(dt, N, idims, odims, gout_s, idims, cin_s, cout, cin, (4,) * 3, (2, 2, 2), one, (-1,) * 3, one, zero)
            dgr.append(_Plan(g, cin, cout, T, None, cout * T, T))
        plans = {"fwd": fwd, "dgrad": dgr, "wgrad": wgr, "odims": odims}
        if len(self._plans) > 64:  # autoregressive sampling visits every prefix length once: keep the cache bounded
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = plans
        return plans

    # ------------------------------------------------------------------ weight packing
    def _pack(self, plan: _Plan, ver, dtype=None):
        # packed operands are shared by every launch geometry that needs the same layout (e.g. a Linear applied to
        # sequences of different lengths), keyed by the pack description (the operand type last: PackSet reads it from there)
        g = plan.geom
        dtype = dtype or self.dtype
        key = (plan.rows, plan.red, plan.ntaps, tuple(plan.lut) if plan.lut is not None else None, plan.s_row, plan.s_red, g.CoutPad, g.Cin, g.Kpad, dtype)
        ent = self._packs.get(key)
        if ent is None or ent[0].device != self.weight.device:
            ent = [torch.empty(g.CoutPad * g.Kpad, dtype=dtype, device=self.weight.device), None]
            self._packs[key] = ent
        plan.wpk = ent[0]
        if ent[1] == ver:
            return
        _ffi.check(_ffi.lib().sa_pack_weights(_ffi.ptr(self.weight), _ffi.ptr(plan.wpk), _ffi.dtype_id(dtype), plan.rows, plan.red,
                                              plan.ntaps, plan.lut_c, plan.s_row, plan.s_red, g.CoutPad, g.Cin, g.Kpad, _ffi.stream()),
                   "sa_pack_weights")
        ent[1] = ver

    def invalidate(self):
        """Call after the weights changed through a raw pointer (our Adam kernel): packed operands are rebuilt lazily."""
        self._epoch += 1

    def _pack_version(self):
        return (self.weight._version, self.weight.data_ptr(), self._epoch)

    def _ensure_packed(self, plans: Sequence[_Plan], dtype=None):
        ver = self._pack_version()
        for pl in plans:
            self._pack(pl, ver, dtype)

    def packed_fwd_operand(self, N: int, idims: Tuple[int, int, int]) -> torch.Tensor:
        """The forward GEMM operand [CoutPad][Kpad] (compute dtype) for this launch geometry, packed and current -- for kernels outside the
        generic launchers that consume the same layout (csrc/conv1.hip)."""
        plans = self._get_plans(N, idims, self.cout, _ru(self.cout, self.vec))
        self._ensure_packed(plans["fwd"], self.fwd_dtype)
        return plans["fwd"][0].wpk

    def _bias_padded(self):
        if self.bias is None:
            return None
        if self.cout % 128 == 0 and self.bias.dtype == torch.float32 and self.bias.is_contiguous():
            return self.bias.detach()   # nothing to pad: the kernels read the parameter itself (no per-step copy)
        ver = (self.bias._version, self.bias.data_ptr(), self._epoch)
        if self._bias_pad is None or self._bias_pad[0] != ver:
            if self._bias_pad is None or self._bias_pad[1].device != self.bias.device:
                bp = torch.zeros(_ru(self.cout, 128), dtype=torch.float32, device=self.bias.device)
            else:
                bp = self._bias_pad[1]      # the padding stays zero: one copy, no fill
            bp[: self.cout] = self.bias.detach()
            self._bias_pad = (ver, bp)
        return self._bias_pad[1]

    # ------------------------------------------------------------------ launches
    @staticmethod
    def _epilogue(bias, addend, mask, alpha, act, mask_mode, add_before_act, out_dtype, slope, out_pre=None, out_lp=None) -> Epilogue:
        ep = Epilogue()
        ep.out_pre = out_pre.data_ptr() if out_pre is not None else None
        ep.out_lp = out_lp.data_ptr() if out_lp is not None else None
        ep.bias = bias.data_ptr() if bias is not None else None
        ep.addend = addend.data_ptr() if addend is not None else None
        ep.mask = mask.data_ptr() if mask is not None else None
        ep.alpha = alpha.data_ptr() if alpha is not None else None
        ep.act, ep.mask_mode, ep.add_before_act = act, (mask_mode if mask is not None else MASK_NONE), int(add_before_act)
        ep.out_dtype = _ffi.dtype_id(out_dtype)
        ep.add_dtype = _ffi.dtype_id(addend.dtype) if addend is not None else 0
        ep.mask_dtype = _ffi.dtype_id(mask.dtype) if mask is not None else 0
        ep.slope = slope
        return ep

    def fprop(self, x: torch.Tensor, *, act=ACT_NONE, addend=None, add_before_act=False, mask=None, mask_mode=MASK_NONE, alpha=None,
              out_dtype=None, out_channels_stride=None, slope=0.2, use_bias=True, want_pre=False, want_lp=False):
        """want_pre / want_lp (dense layers, bf16 extras of the same launch, sa_epilogue.out_pre / out_lp): returns (out, pre, lp) where pre = acc + bias
        before activation / alpha / addend and lp = a bf16 copy of out."""
        N, D, H, W, C = x.shape
        assert x.dtype == self.fwd_dtype and x.is_contiguous() and C == self.cs_in(), (x.dtype, x.shape, self.cs_in())
        out_dtype = out_dtype or self.fwd_dtype
        cout_s = out_channels_stride or self.cout
        plans = self._get_plans(N, (D, H, W), cout_s, _ru(self.cout, self.vec))
        self._ensure_packed(plans["fwd"], self.fwd_dtype)
        od = plans["odims"]
        out = torch.empty((N, *od, cout_s), dtype=out_dtype, device=x.device)
        if cout_s != self.cout:
            out.zero_()
        pre = torch.empty((N, *od, cout_s), dtype=torch.bfloat16, device=x.device) if want_pre else None
        lp = torch.empty((N, *od, cout_s), dtype=torch.bfloat16, device=x.device) if want_lp else None
        ep = self._epilogue(self._bias_padded() if use_bias else None, addend, mask, alpha, act, mask_mode, add_before_act, out_dtype, slope, pre, lp)
        lib, st, did = _ffi.lib(), _ffi.stream(), _ffi.dtype_id(self.fwd_dtype)
        self._run_plans(plans["fwd"], did, x, out, ep, "sa_conv_fprop", _tbytes(addend, mask, pre, lp))
        if want_pre or want_lp:
            return out, pre, lp
        return out

    def _run_plans(self, plans, did, src, dst, ep, what, extra_bytes=0.0):
        """One launch per geometry -- or ONE launch for the eight output-parity classes of a stride-2 layer (sa_conv_fprop_classes: geometries that differ
        only in their offsets, each with its packed operand); the library answers SA_EUNSUPPORTED when the kernel it would pick does not take classes."""
        lib, st = _ffi.lib(), _ffi.stream()
        if len(plans) > 1:
            n = len(plans)
            geoms = (ConvGeom * n)(*[pl.geom for pl in plans])
            wpks = (ctypes.c_void_p * n)(*[pl.wpk.data_ptr() for pl in plans])
            if not getattr(plans[0], "no_classes", False):
                # (a timer bracket is recorded only for a launch that happened: SA_EUNSUPPORTED launches nothing, and its bracket would credit the summed
                #  FLOPs of all classes to whichever kernel ran before)
                rc = []
                pend = len(TIMER.pending) if TIMER is not None else 0
                _launch(None, sum(_geom_flops(pl.geom) for pl in plans),
                        lambda: rc.append(lib.sa_conv_fprop_classes(geoms, n, did, _ffi.ptr(src), wpks, _ffi.ptr(dst), ctypes.byref(ep), st)),
                        abytes=_tbytes(src, dst, *[pl.wpk for pl in plans]) + extra_bytes)
                if rc[0] != _ffi.SA_EUNSUPPORTED:
                    _ffi.check(rc[0], what + " (classes)")
                    return
                if TIMER is not None:
                    del TIMER.pending[pend:]
        for pl in plans:     # (operand bytes: input and output once per LAYER, shared between the launches of its classes)
            _launch(None, _geom_flops(pl.geom),
                    lambda pl=pl: _ffi.check(lib.sa_conv_fprop(ctypes.byref(pl.geom), did, _ffi.ptr(src), _ffi.ptr(pl.wpk), _ffi.ptr(dst), ctypes.byref(ep), st), what),
                    abytes=(_tbytes(src, dst) + extra_bytes) / len(plans) + _tbytes(pl.wpk))

    def dgrad(self, g: torch.Tensor, idims: Tuple[int, int, int], *, addend=None, mask=None, mask_mode=MASK_NONE, out_dtype=None, slope=0.2,
              fwd_out_stride=None) -> torch.Tensor:
        """dx [N, *idims, cs_in] from g [N, *odims, gout_s] (gradient wrt this layer's pre-activation output)."""
        N = g.shape[0]
        gout_s = g.shape[-1]
        assert g.dtype == self.dtype and g.is_contiguous() and gout_s >= self.cout and gout_s % self.vec == 0
        plans = self._get_plans(N, tuple(idims), fwd_out_stride or self.cout, gout_s)
        if plans["dgrad"] is None:
            raise NotImplementedError("data gradient for this conv geometry")
        self._ensure_packed(plans["dgrad"])
        out_dtype = out_dtype or self.dtype
        cin_s = self.cs_in()
        dx = torch.empty((N, *idims, cin_s), dtype=out_dtype, device=g.device)
        if cin_s != self.cin:
            dx.zero_()
        ep = self._epilogue(None, addend, mask, None, ACT_NONE, mask_mode, False, out_dtype, slope)
        lib, st, did = _ffi.lib(), _ffi.stream(), _ffi.dtype_id(self.dtype)
        self._run_plans(plans["dgrad"], did, g, dx, ep, "sa_conv_fprop(dgrad)", _tbytes(addend, mask))
        return dx

    def wgrad(self, x: torch.Tensor, g: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor] = None, fwd_out_stride=None):
        """dw (fp32, reference weight layout, pre-zeroed or accumulating) += x (*) g ; db += colsum(g)."""
        N, D, H, W, C = x.shape
        assert x.dtype == self.dtype and g.dtype == self.dtype and x.is_contiguous() and g.is_contiguous()
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.shape == self.weight.shape
        plans = self._get_plans(N, (D, H, W), fwd_out_stride or self.cout, g.shape[-1])
        lib, st, did = _ffi.lib(), _ffi.stream(), _ffi.dtype_id(self.dtype)
        if db is not None and debug.deterministic():
            # --deterministic: no fused / atomic bias gradient; the launch geometries of one layer partition its output voxels, so the bias gradient is ONE
            # fixed-order column sum over the whole gradient tensor (csrc/deterministic.hip)
            colsum_det(g, self.cout, db)
            db = None
        for pl in plans["wgrad"]:
            nbytes = lib.sa_conv_wgrad_workspace_bytes(ctypes.byref(pl.geom), did)
            if nbytes < 0:
                _ffi.check(int(nbytes), "sa_conv_wgrad_workspace_bytes")
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)  # caching allocator: stream-ordered scratch
            _launch("+wgrad_reduce_kernel", _geom_flops(pl.geom),
                    lambda pl=pl, ws=ws, nbytes=nbytes: _ffi.check(
                        lib.sa_conv_wgrad(ctypes.byref(pl.geom), did, _ffi.ptr(x), _ffi.ptr(g), _ffi.ptr(dw), _ffi.ptr(db), pl.lut_c, pl.s_row, pl.s_red, _ffi.ptr(ws),
                                          nbytes, st),
                        "sa_conv_wgrad"))


class PackSet:
    """Re-packs the GEMM operands of many ConvOps in ONE launch (sa_pack_weights_batch) after the parameters changed.

    The descriptor table lives on the device and is rebuilt only when the set of (parameter, packed operand) addresses changes: parameters sit in
    the flat optimizer buffer and packed operands are allocated once per layout, so in steady state a training step pays one launch instead of
    one per layer and direction.  Operands whose layout first appears later are packed lazily by their op and join the table at the next call."""

    def __init__(self):
        self._sig = None
        self._table = None
        self._first = None
        self._n = 0
        self._blocks = 0

    def repack(self, ops: Sequence["ConvOp"]):
        if debug.host("no_batched_pack"):
            return
        items = [(op, key, ent) for op in ops for key, ent in op._packs.items() if ent[0].device == op.weight.device]
        if not items:
            return
        sig = tuple((op.weight.data_ptr(), ent[0].data_ptr(), key) for op, key, ent in items)
        if sig != self._sig:
            table = (_ffi.PackDesc * len(items))()
            first = [0]
            for d, (op, key, ent) in zip(table, items):
                rows, red, ntaps, lut, s_row, s_red, rows_pad, red_stride, kpad, pdt = key
                d.w, d.wpk = op.weight.data_ptr(), ent[0].data_ptr()
                for t in range(ntaps):
                    d.tap_lut[t] = lut[t] if lut is not None else t
                d.s_row, d.s_red, d.dtype, d.rows, d.red, d.ntaps = s_row, s_red, _ffi.dtype_id(pdt), rows, red, ntaps
                d.rows_pad, d.red_stride, d.Kpad = rows_pad, red_stride, kpad
                first.append(first[-1] + max(1, min(256, ((rows_pad + 63) // 64) * ((kpad + 63) // 64))))   # one block per 64 x 64 tile, capped
            dev = items[0][0].weight.device
            raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8)
            self._table = raw.to(dev)
            self._first = torch.tensor(first, dtype=torch.int32).to(dev)
            self._n, self._blocks, self._sig = len(items), first[-1], sig
        _launch("pack_weights_batch", 0.0,
                lambda: _ffi.check(_ffi.lib().sa_pack_weights_batch(_ffi.ptr(self._table), _ffi.ptr(self._first), self._n, self._blocks, _ffi.stream()),
                                   "sa_pack_weights_batch"))
        for op, key, ent in items:
            ent[1] = op._pack_version()


class RangeRepacker:
    """``FusedAdam(in_backward=...)`` callback pair: ``repacker(lo, hi)`` (``on_range``) re-packs the GEMM operands of every op whose parameter lies wholly
    inside the ranges of the flat buffer stepped so far in this step -- on the stream the optimizer slice ran on, right behind it -- and ``finish()``
    (``on_step``) covers the rest (ops whose parameters are not in the flat buffer) and closes the step.  One PackSet (device descriptor table) per group of
    ops that becomes ready together: the buckets report in the same order every step, so the tables are built once."""

    def __init__(self, flat, ops_fn):
        self.flat, self.ops_fn = flat, ops_fn
        self._ranges, self._done, self._sets = [], set(), {}

    def _span(self, op):
        base = self.flat.data.data_ptr()
        a = (op.weight.data_ptr() - base) // 4
        if a < 0 or a >= self.flat.numel:
            return None
        return a, a + op.weight.numel()

    def _covered(self, span):
        a, b = span
        for lo, hi in sorted(self._ranges):
            if lo > a:
                return False
            if hi > a:
                a = hi
                if a >= b:
                    return True
        return a >= b

    def _repack(self, ops):
        if not ops:
            return
        for op in ops:
            op.invalidate()
            self._done.add(id(op))
        key = tuple(id(op) for op in ops)
        ps = self._sets.get(key)
        if ps is None:
            ps = self._sets[key] = PackSet()
        ps.repack(ops)

    def __call__(self, lo: int, hi: int):
        self._ranges.append((lo, hi))
        ready = []
        for op in self.ops_fn():
            if id(op) in self._done:
                continue
            sp = self._span(op)
            if sp is not None and self._covered(sp):
                ready.append(op)
        self._repack(ready)

    def finish(self):
        self._repack([op for op in self.ops_fn() if id(op) not in self._done])
        self._ranges, self._done = [], set()


def colsum_det(g: torch.Tensor, C: int, db: torch.Tensor):
    """db[c] += sum over all rows of g[..., c] in a fixed order (sa_colsum_det)."""
    lib = _ffi.lib()
    cs = g.shape[-1]
    M = g.numel() // cs
    nb = lib.sa_colsum_det_workspace_bytes(C)
    ws = torch.empty(nb // 4, dtype=torch.float32, device=g.device)
    _ffi.check(lib.sa_colsum_det(_ffi.ptr(g), _ffi.dtype_id(g.dtype), M, C, cs, _ffi.ptr(db), _ffi.ptr(ws), nb, _ffi.stream()), "sa_colsum_det")


def conv1x1_backward(op: "ConvOp", x: torch.Tensor, g: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Weight, bias and (ReLU-masked) data gradient of a 1x1x1 128 -> 128 bf16 convolution in one launch (sa_conv1x1_backward); returns dx,
    or None when the layer is not of that shape (the caller then uses wgrad + dgrad)."""
    if not (op.kind == "conv" and op.k == 1 and op.cin == 128 and op.cout == 128 and op.dtype == torch.bfloat16 and op.w_strides is None
            and not debug.host("no_fused_1x1_bwd") and x.numel() * 2 < 0xffffff00 - (1 << 20)):
        return None
    N, D, H, W, C = x.shape
    assert x.dtype == op.dtype and g.dtype == op.dtype and x.is_contiguous() and g.is_contiguous() and g.shape == x.shape
    plans = op._get_plans(N, (D, H, W), 128, 128)
    op._ensure_packed(plans["dgrad"])
    pw, pd = plans["wgrad"][0], plans["dgrad"][0]
    lib, st, did = _ffi.lib(), _ffi.stream(), _ffi.dtype_id(op.dtype)
    nbytes = lib.sa_conv_wgrad_workspace_bytes(ctypes.byref(pw.geom), did)
    if nbytes <= 0:
        return None
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    if db is not None and debug.deterministic():
        colsum_det(g, op.cout, db)
        db = None
    flops = 2.0 * _geom_flops(pw.geom)
    _launch("+wgrad_reduce_kernel", flops, nbytes=3.0 * x.numel() * x.element_size(),   # algorithmic bytes: read x and g once, write dx once
            fn=lambda: _ffi.check(lib.sa_conv1x1_backward(ctypes.byref(pw.geom), did, _ffi.ptr(x), _ffi.ptr(g), _ffi.ptr(dw), _ffi.ptr(db), pw.s_row, pw.s_red, _ffi.ptr(ws),
                                                        nbytes, _ffi.ptr(pd.wpk), _ffi.ptr(dx), st), "sa_conv1x1_backward"))
    return dx


def cast_pad(src: torch.Tensor, dst_dtype: torch.dtype, dst_stride: int) -> torch.Tensor:
    """[..., C] -> [..., dst_stride] of dst_dtype (zero channel padding) via the HIP cast kernel."""
    src = src.contiguous()
    C = src.shape[-1]
    rows = src.numel() // C
    dst = torch.empty((*src.shape[:-1], dst_stride), dtype=dst_dtype, device=src.device)
    _ffi.check(_ffi.lib().sa_cast_pad(_ffi.ptr(src), _ffi.dtype_id(src.dtype), C, _ffi.ptr(dst), _ffi.dtype_id(dst_dtype), dst_stride, rows, _ffi.stream()),
               "sa_cast_pad")
    return dst
