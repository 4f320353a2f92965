import os, sys
sys.path.insert(0, "/root/repo")
import torch
from synthanatomy_amd import _ffi
lib, st = _ffi.lib(), _ffi.stream()
B, N, G, m, LDF, dv = 2, 200, 2, 266, 272, 64
torch.manual_seed(0)
a = torch.zeros(B, N, G, LDF, device="cuda"); c = torch.zeros(B, N, G, LDF, device="cuda")
a[..., :m] = torch.rand(B, N, G, m, device="cuda") + 0.01
c[..., :m] = torch.rand(B, N, G, m, device="cuda") + 0.01
bb = torch.randn(B * N, G * dv, device="cuda")
bs = torch.rand(B, N, G, device="cuda") + 0.5
out = {}
for rev in (0, 1):
    for exact in ("7", "6"):
        os.environ["SA_SCAN_EXACT"] = exact
        ws = torch.zeros(lib.sa_favor_scan_workspace_bytes(B, N, G, LDF, dv) // 4, device="cuda")
        y = torch.zeros(B * N, G * dv, device="cuda")
        _ffi.check(lib.sa_favor_scan_a(_ffi.ptr(a), _ffi.ptr(c), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(bs), _ffi.ptr(y), G * dv, 0, None, B, N, G, LDF, dv, rev, 0, _ffi.ptr(ws), st))
        torch.cuda.synchronize()
        out[(rev, exact)] = (ws.clone(), y.clone())
    w0, w1 = out[(rev, "7")][0], out[(rev, "6")][0]
    S = (N + 63) // 64
    d = (w0 - w1)[:B*G*S*LDF*dv].view(B, G, S, LDF, dv).abs().amax(dim=(3, 4))
    print("rev", rev, "state max diff per (b,g,chunk):", d.flatten().tolist(), "ref max", float(w0.abs().max()))
    print("  y rel", float((out[(rev,'7')][1]-out[(rev,'6')][1]).norm()/out[(rev,'7')][1].norm()))
