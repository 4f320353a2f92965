#!/usr/bin/env python3
"""Dev: the residual-block kernels of the 80x112x80 level (C = 128, bf16) in isolation -- fused forward (sa_resblock_fprop), 3x3x3 data
gradient, 3x3x3 weight gradient, fused 1x1x1 backward -- timed with HIP events.  `--iters 2 --only fwd` is the form to run under rocprofv3 --pmc."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from synthanatomy_amd import _ffi, engine
from synthanatomy_amd.networks.vqvae.baseline import ResidualLayer, _GradCtx, _ResStage


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dims", default="80,112,80")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--save", default="", help="write y / h / dx of the first call to this file")
    ap.add_argument("--compare", default="", help="compare y / h / dx with a file written by --save (another kernel variant)")
    a = ap.parse_args()
    dims = tuple(int(v) for v in a.dims.split(","))
    torch.manual_seed(0)
    mod = ResidualLayer(128, 128, 0.0).cuda()
    st = _ResStage(mod, in_act=True, dtype=torch.bfloat16)
    x = torch.relu(torch.randn(a.batch, *dims, 128, device="cuda")).to(torch.bfloat16)
    G = (torch.randn_like(x, dtype=torch.float32) * 0.01).to(torch.bfloat16)
    M = x.numel() // 128
    f3, f1 = 2.0 * M * 27 * 128 * 128, 2.0 * M * 128 * 128
    tape = []
    y = st.fwd(x, tape)
    h = tape[0][1]
    import hashlib
    dx = st.c3.dgrad(G, dims, addend=G, mask=x, mask_mode=_ffi.MASK_POS)
    torch.cuda.synchronize()
    sig = lambda t: hashlib.sha1(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    print(f"signatures: y {sig(y)} h {sig(h)} dx {sig(dx)}", flush=True)
    if a.save:
        torch.save({"y": y.cpu(), "h": h.cpu(), "dx": dx.cpu()}, a.save)
    if a.compare:
        ref = torch.load(a.compare)
        for k_, t_ in (("y", y), ("h", h), ("dx", dx)):
            d_ = (t_.cpu().float() - ref[k_].float()).abs()
            print(f"    vs saved {k_}: max abs {float(d_.max()):.3e} (ref max {float(ref[k_].float().abs().max()):.3e}), mismatching elements {float((d_ > 0).float().mean()):.4f}", flush=True)
    dw3, db3 = torch.zeros_like(mod[0].weight), torch.zeros_like(mod[0].bias)
    dw1, db1 = torch.zeros_like(mod[3].weight), torch.zeros_like(mod[3].bias)
    st.c3.wgrad(x, G, dw3, db3)
    torch.cuda.synchronize()
    import hashlib as _h
    print("signature dw3", _h.sha1(dw3.cpu().numpy().tobytes()).hexdigest()[:12], "db3", _h.sha1(db3.cpu().numpy().tobytes()).hexdigest()[:12], float(dw3.abs().sum()), flush=True)
    cases = {
        "fwd": (lambda: st.fwd(x, []), f3 + f1),
        "fwd_eval": (lambda: st.fwd(x, None), f3 + f1),
        "dgrad3": (lambda: st.c3.dgrad(G, dims, addend=G, mask=x, mask_mode=_ffi.MASK_POS), f3),
        "wgrad3": (lambda: st.c3.wgrad(x, G, dw3, db3), f3),
        "bwd1x1": (lambda: engine.conv1x1_backward(st.c1, h, G, dw1, db1), 2 * f1),
    }
    for name, (fn, fl) in cases.items():
        if a.only and name not in a.only.split(","):
            continue
        import ctypes
        lib = _ffi.lib()
        timing = hasattr(lib, "sa_debug_timing")
        if timing:
            torch.cuda.synchronize()
            lib.sa_debug_timing(None, 1)
        t = timeit(fn, a.iters)
        print(f"{name:10s} {t * 1e3:9.1f} us  {fl / t / 1e9:8.1f} TFLOP/s   [{_ffi.lib().sa_last_conv_kernel().decode()}]", flush=True)
        if timing:
            buf = (ctypes.c_ulonglong * 8)()
            lib.sa_debug_timing(buf, 1)
            nb = max(1, buf[7])
            names = ["compute", "dma_wait", "barrier", "halo", "prologue", "epilogue", "total"]
            print("    per block (cycles of the 100 MHz? s_memtime clock): " + "  ".join(f"{n}={buf[i] / nb:.0f}" for i, n in enumerate(names)) + f"  blocks={nb}")


if __name__ == "__main__":
    main()
