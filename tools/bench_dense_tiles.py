#!/usr/bin/env python3
"""Dev: the eight dense forward / data-gradient GEMM shapes of one Performer layer (M = batch * N rows) on the im2col-order kernel with the narrow
(128 x 64, three blocks per CU) and the wide (128 x 128, two blocks per CU) tiles, next to hipBLASLt through torch as an outside reference point."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from synthanatomy_amd import _ffi, debug, engine


def timeit(fn, n=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 8400
    shapes = [("qkv fwd", 512, 3072), ("to_out fwd", 1024, 512), ("w1 fwd", 512, 2048), ("w2 fwd", 2048, 512), ("w2 dgrad", 512, 2048), ("w1 dgrad", 2048, 512),
              ("to_out dgrad", 512, 1024), ("qkv dgrad", 3072, 512)]
    only = os.environ.get("DENSE_ONLY")
    for name, K, N in shapes:
        if only and only != name:
            continue
        w = torch.randn(N, K, 1, 1, 1, device="cuda") * K ** -0.5
        op = engine.ConvOp("conv", K, N, 1, 1, 0, w, None, torch.bfloat16)
        x = torch.randn(1, 1, 1, R, K, device="cuda").to(torch.bfloat16)
        fl = 2.0 * R * K * N
        ref = None
        cols = []
        for label, kw in (("2buf narrow", dict(no_small_tiles=False, no_kgroups=True)), ("2buf wide", dict(no_small_tiles=True, no_kgroups=True)),
                          ("product", dict())):
            with debug.override(**kw):
                t = timeit(lambda: op.fprop(x, out_dtype=torch.float32))
                y = op.fprop(x, out_dtype=torch.float32)
                kk = _ffi.lib().sa_last_conv_kernel().decode()
            if ref is None:
                ref = y
            err = float((y - ref).abs().max() / ref.abs().max())
            cols.append(f"{label} {t:6.1f} us {fl / t / 1e6:6.1f} TF [{kk.replace('conv_fprop_dma_kernel', 'dma').replace('dense_gemm_kernel', 'dense').replace('unsigned short', 'bf16')}] d={err:.1e}")
        xt, wt = x.view(R, K), w.view(N, K).to(torch.bfloat16)
        tb = timeit(lambda: torch.nn.functional.linear(xt, wt))
        print(f"{name:13s} K={K:5d} N={N:5d} | " + " | ".join(cols) + f" | hipBLASLt {tb:6.1f} us {fl / tb / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
