#!/usr/bin/env python3
"""Dev: weight gradients of the four dense layers of a Performer layer (M = 8 400 rows) -- kernel + split reduction, timed with events.
SA_WGRAD_MIN_BLOCKS=<n> changes the number of row-range splits (blocks = tiles x splits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthanatomy_amd import engine

def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8400
for name, K, N in (("qkv", 512, 3072), ("to_out", 1024, 512), ("w1", 512, 2048), ("w2", 2048, 512)):
    w = torch.randn(N, K, 1, 1, 1, device="cuda") * K ** -0.5
    op = engine.ConvOp("conv", K, N, 1, 1, 0, w, torch.zeros(N, device="cuda"), torch.bfloat16)
    x = torch.randn(1, 1, 1, R, K, device="cuda").to(torch.bfloat16)
    g = torch.randn(1, 1, 1, R, N, device="cuda").to(torch.bfloat16)
    dw, db = torch.zeros_like(w), torch.zeros(N, device="cuda")
    t = timeit(lambda: op.wgrad(x, g, dw, db))
    print(f"{name:7s} K={K:5d} N={N:5d}  {t:7.1f} us  {2.0 * R * K * N / t / 1e6:6.1f} TF", flush=True)
