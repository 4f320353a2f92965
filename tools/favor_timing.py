"""dev: s_memtime phase sums of the FAVOR+ scan-B body (library built with SA_EXTRA_HIPCC_FLAGS=-DSA_TIMING_FAVOR into a separate .so)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthanatomy_amd import _ffi
import bench, argparse
lib = ctypes.CDLL(_ffi.LIB_PATH)
buf = (ctypes.c_ulonglong * 24)()
args = argparse.Namespace(dtype="bf16", performer_shape="10,14,10", performer_batch=6, warmup=2, steps=3, sampling=False, ddp_mode=None, grad_transport=None, no_kernel_timer=True, opt_in_backward=False)
torch.zeros(1, device="cuda")
lib.sa_debug_timing_favor(None, 1)
res = bench.bench_performer(args, 0, 1, torch.device("cuda", 0))
lib.sa_debug_timing_favor(buf, 0)
v = list(buf)
n = max(v[8], 1)
names = ["prologue", "features", "tile writes (+wait prev)", "barrier 2", "two GEMMs", "VALU + split", "dx GEMM", "epilogue", "blocks", "total"]
print(res["value"], "tokens/s")
for i, nm in enumerate(names):
    print(f"{nm:28s} {v[i] / n:12.0f} cycles per block" + (f"  ({v[i] / n / 5:8.0f} per slab)" if 1 <= i <= 6 else ""))
