#!/usr/bin/env python3
"""The dense layers of the Performer step (1x1x1 'convolutions' over R = batch * N rows) one by one: fprop / dgrad / wgrad time and TFLOP/s
(dev tool; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthanatomy_amd import engine, _ffi


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 8400
    cases = [("to_q/k/v/out 512->512", R, 512, 512, torch.bfloat16), ("w1 512->2048", R, 512, 2048, torch.bfloat16), ("w2 2048->512", R, 2048, 512, torch.bfloat16),
             ("proj 64->266 fp32", R * 8, 64, 266, torch.float32)]
    for name, rows, cin, cout, dt in cases:
        w = torch.randn(cout, cin, 1, 1, 1, device="cuda") * cin ** -0.5
        b = torch.zeros(cout, device="cuda")
        op = engine.ConvOp("conv", cin, cout, 1, 1, 0, w, b, dt)
        x = torch.randn(1, 1, 1, rows, cin, device="cuda").to(dt)
        vec = engine.vec_of(dt)
        cs = (cout + 15) // 16 * 16 if dt == torch.float32 else (cout + vec - 1) // vec * vec
        g = torch.randn(1, 1, 1, rows, cs, device="cuda").to(dt)
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        fl = 2.0 * rows * cin * cout
        kw = dict(out_channels_stride=cs) if dt == torch.float32 else {}
        tf = timeit(lambda: op.fprop(x, out_dtype=torch.float32, **kw)); kf = _ffi.last_conv_kernel() if hasattr(_ffi, "last_conv_kernel") else ""
        td = timeit(lambda: op.dgrad(g, (1, 1, rows), out_dtype=torch.float32, **({"fwd_out_stride": cs} if dt == torch.float32 else {})))
        tw = timeit(lambda: op.wgrad(x, g, dw, db))
        print(f"{name:26s} R={rows:6d}  fprop {tf:7.1f} us {fl/tf/1e6:6.1f} TF | dgrad {td:7.1f} us {fl/td/1e6:6.1f} TF | wgrad {tw:7.1f} us {fl/tw/1e6:6.1f} TF", flush=True)
    # reference: hipBLASLt through torch
    for cin, cout in ((512, 512), (512, 2048), (2048, 512)):
        x = torch.randn(R, cin, device="cuda", dtype=torch.bfloat16); w = torch.randn(cout, cin, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: torch.nn.functional.linear(x, w))
        print(f"torch linear {cin}->{cout}: {t:7.1f} us {2.0*R*cin*cout/t/1e6:6.1f} TF")


if __name__ == "__main__":
    main()
