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
    for name, K, N in shapes:
        w = torch.randn(N, K, 1, 1, 1, device="cuda") * K ** -0.5
        op = engine.ConvOp("conv", K, N, 1, 1, 0, w, None, torch.bfloat16)
        x = torch.randn(1, 1, 1, R, K, device="cuda").to(torch.bfloat16)
        fl = 2.0 * R * K * N
        res = []
        for wide in (False, True):
            with debug.override(no_small_tiles=wide, no_kgroups=True):
                t = timeit(lambda: op.fprop(x, out_dtype=torch.float32))
                res.append((t, _ffi.lib().sa_last_conv_kernel().decode()))
        tk = timeit(lambda: op.fprop(x, out_dtype=torch.float32))      # the product's dispatch (two K groups where the rule selects them)
        kk = _ffi.lib().sa_last_conv_kernel().decode()
        xt, wt = x.view(R, K), w.view(N, K).to(torch.bfloat16)
        tb = timeit(lambda: torch.nn.functional.linear(xt, wt))
        print(f"{name:13s} K={K:5d} N={N:5d}  narrow {res[0][0]:6.1f} us {fl / res[0][0] / 1e6:6.1f} TF | wide {res[1][0]:6.1f} us {fl / res[1][0] / 1e6:6.1f} TF | "
              f"product {tk:6.1f} us {fl / tk / 1e6:6.1f} TF | hipBLASLt {tb:6.1f} us {fl / tb / 1e6:6.1f} TF   [{res[0][1]} / {res[1][1]} / {kk}]", flush=True)


if __name__ == "__main__":
    main()
