#!/usr/bin/env python3
"""Chunked FAVOR+ scans at the README shape (B=6, N=1400, 8 global heads, 266 features padded to 272): time per entry point for the split-bf16
kernels and the exact-fp32 ones (SA_SCAN_EXACT bits), and the relative difference between the two (dev tool; run on the GPU box)."""
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


def main():
    lib, st = _ffi.lib(), _ffi.stream()
    B, N, G, m, LDF, dv = 6, 1400, 8, 266, 272, 64
    if len(sys.argv) > 2:
        B, N = int(sys.argv[1]), int(sys.argv[2])
    torch.manual_seed(0)
    a = torch.zeros(B, N, G, LDF, device="cuda"); c = torch.zeros(B, N, G, LDF, device="cuda")
    a[..., :m] = torch.rand(B, N, G, m, device="cuda") * 0.1 + 1e-3
    c[..., :m] = torch.rand(B, N, G, m, device="cuda") * 0.1 + 1e-3
    bb = torch.randn(B * N, G * dv, device="cuda"); cc = torch.randn(B * N, G * dv, device="cuda")
    bs = torch.rand(B, N, G, device="cuda") + 0.5
    ws = torch.empty(lib.sa_favor_scan_workspace_bytes(B, N, G, LDF, dv) // 4, device="cuda")
    res = {}
    for exact in ("0", "7"):
        lib.sa_set_debug_flags((lib.sa_get_debug_flags() & ~(7 << 10)) | (int(exact) << 10))   # SA_DBG_SCAN_EXACT bits
        yn = torch.zeros(B * N, G * dv, device="cuda"); inv = torch.zeros(B * N * G, device="cuda")
        fa = lambda: _ffi.check(lib.sa_favor_scan_a_norm(_ffi.ptr(a), _ffi.ptr(c), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(yn), G * dv, 0, _ffi.ptr(inv), 1e-6,
                                                         B, N, G, LDF, dv, _ffi.ptr(ws), 0, st))
        y1 = torch.zeros(B, N, G, LDF, device="cuda")
        fb = lambda: _ffi.check(lib.sa_favor_scan_b_cum(_ffi.ptr(a), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(bs), _ffi.ptr(cc), G * dv, 0, None, _ffi.ptr(y1), _ffi.ptr(bs),
                                                        1, 0.25, B, N, G, LDF, dv, 1, _ffi.ptr(ws), 0, st))
        y2 = torch.zeros(B * N, G * dv, device="cuda")
        fs = lambda: _ffi.check(lib.sa_favor_scan_a_state(_ffi.ptr(a), _ffi.ptr(c), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(bs), _ffi.ptr(y2), G * dv, 0, None, B, N, G, LDF, dv,
                                                          1, 0, _ffi.ptr(ws), 3, st))
        ta, tb = timeit(fa), timeit(fb)
        fb(); ts = timeit(fs)   # scan_a_state reuses the states fb left in ws
        res[exact] = (yn.clone(), inv.clone(), y1.clone(), y2.clone())
        print(f"SA_SCAN_EXACT={exact}: scan_a_norm {ta:7.1f} us   scan_b_cum(rev) {tb:7.1f} us   scan_a_state(shared) {ts:7.1f} us", flush=True)
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    print("split vs exact:  " + "  ".join(f"{n} {rel(p, q):.1e}" for n, p, q in zip(("y_norm", "inv", "y_b", "y_a_state"), res["0"], res["7"])))


if __name__ == "__main__":
    main()
