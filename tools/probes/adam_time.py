import sys, torch
sys.path.insert(0, "/root/repo")
from synthanatomy_amd import _ffi
lib, st = _ffi.lib(), _ffi.stream()
n = 79_000_000
p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4))
v.abs_()
def run(): _ffi.check(lib.sa_adam(_ffi.ptr(p), _ffi.ptr(g), _ffi.ptr(m), _ffi.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 3, 1.0, st))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
print(f"sa_adam n={n}: {t:.1f} us, {n * 28 / t / 1e6:.2f} TB/s")
