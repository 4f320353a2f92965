#!/usr/bin/env python3
"""FAVOR+ projection kernels at the README shape (67 200 rows x 64 -> 266 features padded to 272): time and effective HBM rate (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthanatomy_amd import _ffi
from bench_scan import timeit
lib, st = _ffi.lib(), _ffi.stream()
rows, m, LDF = 67200, 266, 272
x = torch.randn(rows, 64, device="cuda"); P_ = torch.randn(m, 64, device="cuda") * 0.35
dd = torch.empty(rows, LDF, device="cuda"); dx = torch.empty(rows, 64, device="cuda"); add = torch.randn(rows, 64, device="cuda")
tf = timeit(lambda: _ffi.check(lib.sa_favor_project(_ffi.ptr(x), 64, 1, _ffi.ptr(P_), _ffi.ptr(dd), None, rows, m, LDF, 64, st)))
tb = timeit(lambda: _ffi.check(lib.sa_favor_project_bwd(_ffi.ptr(dd), _ffi.ptr(P_), _ffi.ptr(add), _ffi.ptr(dx), 64, 1, rows, m, LDF, 64, st)))
print(f"project fwd {tf:6.1f} us ({rows*(64+LDF)*4/tf/1e6:5.2f} TB/s)   bwd {tb:6.1f} us ({rows*(128+LDF)*4/tb/1e6:5.2f} TB/s)")
