#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernels at the config-2 shapes (dev tool; run on the GPU box)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthanatomy_amd import engine

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    B = a.batch
    L1, L2, L3, L4 = (80, 112, 80), (40, 56, 40), (20, 28, 20), (10, 14, 10)
    cases = [
        ("res3x3 C128 @L1", "conv", 128, 128, 3, 1, 1, L1),
        ("res1x1 C128 @L1", "conv", 128, 128, 1, 1, 0, L1),
        ("down k4s2 128->128 L1->L2", "conv", 128, 128, 4, 2, 1, L1),
        ("up convT 128->128 L2->L1", "convT", 128, 128, 4, 2, 1, L2),
        ("res3x3 C128 @L2", "conv", 128, 128, 3, 1, 1, L2),
        ("first k4s2 1->128 V0->L1", "conv", 1, 128, 4, 2, 1, (160, 224, 160)),
        ("last convT 128->1 L1->V0", "convT", 128, 1, 4, 2, 1, L1),
        ("res3x3 C256 @L4", "conv", 256, 256, 3, 1, 1, L4),
    ]
    for name, kind, cin, cout, k, s, p, dims in cases:
        if a.only and a.only not in name: continue
        T = k ** 3
        w = torch.randn((cout, cin, k, k, k) if kind == "conv" else (cin, cout, k, k, k), device="cuda") * 0.05
        b = torch.zeros(cout, device="cuda")
        op = engine.ConvOp(kind, cin, cout, k, s, p, w, b, dt)
        x = torch.randn(B, *dims, op.cs_in(), device="cuda").to(dt)
        od = op.out_dims(dims)
        M_out = B * od[0] * od[1] * od[2]
        flops = 2.0 * M_out * T * cin * cout if kind == "conv" else 2.0 * B * dims[0] * dims[1] * dims[2] * T * cin * cout
        vec = engine.vec_of(dt)
        gs = (cout + vec - 1) // vec * vec
        g = torch.randn(B, *od, gs, device="cuda").to(dt)
        dw = torch.zeros_like(w); db = torch.zeros_like(b)
        t_f = timeit(lambda: op.fprop(x, out_channels_stride=cout if cout % vec == 0 else cout, out_dtype=dt if cout % vec == 0 else torch.float32))
        t_w = timeit(lambda: op.wgrad(x, g, dw, None))
        try:
            t_d = timeit(lambda: op.dgrad(g, dims))
        except NotImplementedError:
            t_d = float("nan")
        print(f"{name:28s} fprop {t_f:8.3f} ms {flops/t_f/1e9:7.1f} TF | dgrad {t_d:8.3f} ms {flops/t_d/1e9:7.1f} TF | wgrad {t_w:8.3f} ms {flops/t_w/1e9:7.1f} TF", flush=True)

if __name__ == "__main__":
    main()
