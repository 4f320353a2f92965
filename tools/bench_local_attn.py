#!/usr/bin/env python3
"""Local-window attention kernels at the README shape (B=6, N=1400, 8 local heads, W=420): time and error against an fp64 band
reference, for the split-bf16 and the exact-fp32 paths (dev tool; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthanatomy_amd import _ffi


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def band_ref(q, k, v, go, W):
    B, L, N, dh = q.shape
    q, k, v = (t.double().requires_grad_(True) for t in (q, k, v))
    i = torch.arange(N, device=q.device)
    lo = ((i // W - 1) * W).clamp_min(0)
    mask = (i[None, :] <= i[:, None]) & (i[None, :] >= lo[:, None])
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    o = torch.softmax(s.masked_fill(~mask, float("-inf")), -1) @ v
    o.backward(go.double())
    return o.detach(), q.grad, k.grad, v.grad


def main():
    lib, st = _ffi.lib(), _ffi.stream()
    B, L, N, W, dh = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (6, 8, 1400, 420, 64)))
    torch.manual_seed(0)
    q, k, v, go = (torch.randn(B, L, N, dh, device="cuda") for _ in range(4))
    ref = band_ref(q, k, v, go, W)
    pack = lambda t: t.permute(0, 2, 1, 3).reshape(B * N, L * dh).contiguous()
    unpack = lambda t: t.view(B, N, L, dh).permute(0, 2, 1, 3)
    qd, kd, vd, god = pack(q), pack(k), pack(v), pack(go)
    for exact in ("0", "1"):
        lib.sa_set_debug_flags((lib.sa_get_debug_flags() & ~(1 << 9)) | (int(exact) << 9))   # SA_DBG_LOCAL_ATTN_EXACT
        o = torch.empty_like(qd); lse = torch.empty(B * N * L, device="cuda")
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        Db = torch.empty(B * N * L, device="cuda")
        fwd = lambda: _ffi.check(lib.sa_local_attn_fwd(_ffi.ptr(qd), L * dh, 0, _ffi.ptr(kd), L * dh, 0, _ffi.ptr(vd), L * dh, 0, _ffi.ptr(o), L * dh, 0,
                                                       _ffi.ptr(lse), B, N, L, W, dh, None, st))
        bwd = lambda: _ffi.check(lib.sa_local_attn_bwd(_ffi.ptr(qd), L * dh, 0, _ffi.ptr(kd), L * dh, 0, _ffi.ptr(vd), L * dh, 0, _ffi.ptr(o), _ffi.ptr(god),
                                                       L * dh, 0, _ffi.ptr(lse), _ffi.ptr(dq), _ffi.ptr(dk), _ffi.ptr(dv), _ffi.ptr(Db), B, N, L, W, dh, None, st))
        tf, tb = timeit(fwd), timeit(bwd)
        rel = lambda a, b: float((a.double() - b).norm() / b.norm())
        mx = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        errs = [(rel(unpack(t), r), mx(unpack(t), r)) for t, r in zip((o, dq, dk, dv), ref)]
        print(f"exact={exact}: fwd {tf:7.1f} us  bwd {tb:7.1f} us   rel/max err  " + "  ".join(f"{n} {e[0]:.1e}/{e[1]:.1e}" for n, e in zip("o dq dk dv".split(), errs)),
              flush=True)


if __name__ == "__main__":
    main()
