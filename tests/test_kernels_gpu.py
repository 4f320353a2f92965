"""GPU: each HIP kernel through the C ABI against a plain PyTorch fp32 CPU reference of the same op."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402


def _ops():
    from synthanatomy_amd import _ffi, engine
    return _ffi, engine


def _tol(dtype):
    # bf16 operands are rounded once (inputs AND weights are pre-rounded in the reference), so only the output rounding
    # and fp32 summation order differ
    return (2e-5, 2e-5) if dtype == torch.float32 else (1.2e-2, 1e-2)


def _close(got, ref, dtype, what=""):
    rtol, atol = _tol(dtype)
    scale = ref.abs().max().item() + 1e-12
    err = (got.double() - ref.double()).abs().max().item()
    assert err <= atol * scale + rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _cl(x):  # NCDHW -> channels-last
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _rt(x, dtype):  # round-trip through the compute dtype (what the kernel sees)
    return x.to(dtype).float()


CASES = [
    # kind, cin, cout, k, s, p, dims
    ("conv", 16, 24, 3, 1, 1, (6, 7, 9)),
    ("conv", 8, 136, 3, 1, 1, (5, 6, 7)),
    ("conv", 32, 40, 4, 2, 1, (8, 10, 12)),
    ("conv", 128, 128, 1, 1, 0, (4, 5, 6)),
    ("conv", 24, 16, 4, 1, 1, (6, 6, 7)),
    ("conv", 1, 16, 4, 2, 1, (12, 8, 10)),
    ("convT", 32, 24, 4, 2, 1, (4, 5, 6)),
    ("convT", 16, 1, 4, 2, 1, (5, 4, 6)),
    ("convT", 136, 64, 4, 2, 1, (3, 4, 3)),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}-k{c[3]}s{c[4]}" for c in CASES])
def test_conv_fprop_dgrad_wgrad(case, dtype):
    _ffi, engine = _ops()
    kind, cin, cout, k, s, p, dims = case
    torch.manual_seed(hash(case) % 1000)
    N = 2
    dev = "cuda"
    vec = engine.vec_of(dtype)
    wshape = (cout, cin, k, k, k) if kind == "conv" else (cin, cout, k, k, k)
    w = _rt(torch.randn(wshape) * 0.1, dtype)
    b = torch.randn(cout) * 0.1
    x = _rt(torch.randn(N, cin, *dims), dtype)
    wd, bd = w.to(dev), b.to(dev)
    op = engine.ConvOp(kind, cin, cout, k, s, p, wd, bd, dtype)
    xin = engine.cast_pad(_cl(x).to(dev), dtype, op.cs_in())

    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    if kind == "conv":
        yr = F.conv3d(xr, wr, br, stride=s, padding=p)
    else:
        yr = F.conv_transpose3d(xr, wr, br, stride=s, padding=p)

    # ---- forward, plain and with fused epilogues
    cout_s = cout if cout % vec == 0 else cout
    y = op.fprop(xin, out_dtype=torch.float32, out_channels_stride=cout_s)
    torch.cuda.synchronize()
    _close(y.cpu()[..., :cout], _cl(yr.detach()), dtype, "fprop")

    y2 = op.fprop(xin, act=_ffi.ACT_RELU, out_dtype=torch.float32, out_channels_stride=cout_s)
    _close(y2.cpu()[..., :cout], _cl(F.relu(yr.detach())), dtype, "fprop+relu")

    if cout % vec == 0:
        add = _rt(torch.randn(N, *y.shape[1:4], cout), dtype)
        y3 = op.fprop(xin, act=_ffi.ACT_RELU, addend=add.to(dev).to(dtype), add_before_act=True)
        assert y3.dtype == dtype
        _close(y3.float().cpu(), F.relu(_cl(yr.detach()) + add), torch.bfloat16 if dtype == torch.bfloat16 else dtype, "fprop+add+relu")
        mk = torch.randn_like(add)
        y4 = op.fprop(xin, addend=add.to(dev).to(dtype), mask=mk.to(dev).to(dtype), mask_mode=_ffi.MASK_POS, out_dtype=torch.float32)
        _close(y4.cpu(), (_cl(yr.detach()) + add) * (_rt(mk, dtype) > 0), dtype, "fprop+add+mask")

    # ---- backward
    g = _rt(torch.randn_like(yr), dtype)
    yr.backward(g)
    gin = engine.cast_pad(_cl(g).to(dev), dtype, (cout + vec - 1) // vec * vec)
    dw = torch.zeros_like(wd)
    db = torch.zeros_like(bd)
    op.wgrad(xin, gin, dw, db)
    torch.cuda.synchronize()
    _close(dw.cpu(), wr.grad, dtype, "wgrad")
    _close(db.cpu(), br.grad, dtype, "bgrad")
    if not (kind == "conv" and s == 2 and any(d % 2 for d in dims)):
        dx = op.dgrad(gin, dims, out_dtype=torch.float32)
        torch.cuda.synchronize()
        _close(dx.cpu()[..., :cin], _cl(xr.grad), dtype, "dgrad")


@pytest.mark.parametrize("fwd_dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind,cin,cout,dims", [("conv", 128, 128, (16, 112, 80)), ("conv", 128, 256, (14, 112, 80)), ("convT", 128, 128, (8, 56, 40)),
                                                 ("convT", 256, 128, (8, 52, 38))])
def test_stride2_layers_on_the_cell_mainloop(kind, cin, cout, dims, fwd_dtype):
    """Forward and data gradient of Conv3d k4 s2 p1 / ConvTranspose3d k4 s2 p1 at production widths: conv_fprop_cells256_kernel (4 x 8 x 8-cell tiles, eight taps per
    staged halo image) against the im2col-order kernel (same products, another summation order) and against torch on the same rounded operands; the second shapes
    have ragged tiles (an odd number of cell planes; 52 x 38 cells per plane) and four channel chunks / two output tiles."""
    _ffi, engine = _ops()
    from synthanatomy_amd import debug
    torch.manual_seed(cin + cout + dims[2])
    dtype, N, dev = torch.bfloat16, 4, "cuda"
    wshape = (cout, cin, 4, 4, 4) if kind == "conv" else (cin, cout, 4, 4, 4)
    w = (torch.randn(wshape, device=dev) * 0.05).to(dtype).float()
    b = torch.randn(cout, device=dev) * 0.1
    op = engine.ConvOp(kind, cin, cout, 4, 2, 1, w, b, dtype, fwd_dtype=fwd_dtype)
    x = torch.randn(N, *dims, cin, device=dev).to(fwd_dtype)                  # channels-last
    odims = op.out_dims(dims)
    g = (torch.randn(N, *odims, cout, device=dev) * 0.1).to(dtype)
    res = {}
    for name, flags in (("im2col", dict(no_cells256=True)), ("cells256", {}), ("cells256_separate", dict(no_class_launch=True)),
                        ("im2col_separate", dict(no_cells256=True, no_class_launch=True))):
        with debug.override(**flags):
            y = op.fprop(x, act=_ffi.ACT_RELU, out_dtype=torch.float32)
            kf = _ffi.lib().sa_last_conv_kernel().decode()
            dx = op.dgrad(g, dims, out_dtype=torch.float32)
            kd = _ffi.lib().sa_last_conv_kernel().decode()
            torch.cuda.synchronize()
            res[name] = (y, dx, kf, kd)
    assert res["im2col"][2].startswith("conv_fprop_dma_kernel") and res["im2col"][3].startswith("conv_fprop_dma_kernel"), res["im2col"][2:]
    # round 5: the 256-voxel cell mainloop (4 x 8 x 8-cell tiles, four plane slots) is what the dispatcher picks for these layers
    assert res["cells256"][2].startswith("conv_fprop_cells256_kernel") and res["cells256"][3].startswith("conv_fprop_cells256_kernel"), res["cells256"][2:]
    # the eight output-parity classes of a layer in ONE launch (sa_conv_fprop_classes) against eight launches: the same blocks, bit for bit
    for fam in ("cells256", "im2col"):
        assert torch.equal(res[fam][0], res[fam + "_separate"][0]) and torch.equal(res[fam][1], res[fam + "_separate"][1]), fam
    for i, what in ((0, "forward"), (1, "data gradient")):
        scale = float(res["im2col"][i].abs().max())
        assert float((res["cells256"][i] - res["im2col"][i]).abs().max()) <= 2e-5 * scale + 1e-6, what + " (256-voxel tiles)"
    xr = x.float().permute(0, 4, 1, 2, 3).requires_grad_(True)
    yr = F.conv3d(xr, w, b, stride=2, padding=1) if kind == "conv" else F.conv_transpose3d(xr, w, b, stride=2, padding=1)
    yr.backward(g.float().permute(0, 4, 1, 2, 3))
    _close(res["cells256"][0].cpu(), F.relu(yr.detach()).permute(0, 2, 3, 4, 1).cpu(), torch.float32, "forward vs torch fp32 on the rounded operands")
    _close(res["cells256"][1].cpu(), xr.grad.permute(0, 2, 3, 4, 1).cpu(), torch.float32, "data gradient vs torch")


def test_conv_large_m_tail():
    """M not a multiple of 128, several N tiles, cout tail, fp32."""
    _ffi, engine = _ops()
    torch.manual_seed(1)
    cin, cout, dims = 16, 264, (7, 9, 11)
    w = torch.randn(cout, cin, 3, 3, 3) * 0.05
    x = torch.randn(3, cin, *dims)
    op = engine.ConvOp("conv", cin, cout, 3, 1, 1, w.cuda(), None, torch.float32)
    y = op.fprop(_cl(x).cuda(), use_bias=False)
    _close(y.cpu(), _cl(F.conv3d(x, w, None, padding=1)), torch.float32, "fprop tail")


@pytest.mark.parametrize("dtype,kind,cin,cout,k,st,dims", [
    (torch.bfloat16, "conv", 128, 128, 3, 1, (33, 45, 47)),
    (torch.bfloat16, "conv", 128, 256, 3, 1, (35, 40, 47)),      # two-plane tiles with an odd number of planes, ragged in W, two channel tiles
    (torch.bfloat16, "conv", 64, 136, 3, 1, (29, 48, 50)),
    (torch.float32, "conv", 32, 128, 3, 1, (33, 45, 47)),
    (torch.bfloat16, "conv", 128, 128, 4, 2, (82, 84, 86)),
    (torch.bfloat16, "convT", 128, 128, 4, 2, (21, 40, 42)),
    (torch.bfloat16, "conv", 128, 128, 1, 1, (33, 45, 47)),
])
def test_conv_big_m_mainloop(dtype, kind, cin, cout, k, st, dims):
    """M >= 65536 rows with Cin*sizeof a multiple of 128 B selects the 8-wave ping-pong LDS-DMA mainloop (256-voxel tiles, M tail, padding
    rows, stride-2 and transposed geometry, fused epilogue) and the dgrad that runs on it with the roles swapped."""
    _ffi, engine = _ops()
    torch.manual_seed(7)
    pad = 0 if k == 1 else 1
    wshape = (cout, cin, k, k, k) if kind == "conv" else (cin, cout, k, k, k)
    w = _rt(torch.randn(wshape) * 0.05, dtype)
    b = torch.randn(cout) * 0.1
    x = _rt(torch.randn(1, cin, *dims), dtype)
    op = engine.ConvOp(kind, cin, cout, k, st, pad, w.cuda(), b.cuda(), dtype)
    xin = engine.cast_pad(_cl(x).cuda(), dtype, op.cs_in())
    xr = x.clone().requires_grad_(True)
    yr = F.conv3d(xr, w, b, stride=st, padding=pad) if kind == "conv" else F.conv_transpose3d(xr, w, b, stride=st, padding=pad)
    y = op.fprop(xin, out_dtype=torch.float32, out_channels_stride=cout)
    torch.cuda.synchronize()
    _close(y.cpu()[..., :cout], _cl(yr.detach()), dtype, "fprop")
    add = _rt(torch.randn(1, *y.shape[1:4], cout), dtype)
    y3 = op.fprop(xin, act=_ffi.ACT_RELU, addend=add.cuda().to(dtype), add_before_act=True)
    _close(y3.float().cpu(), F.relu(_cl(yr.detach()) + add), torch.bfloat16 if dtype == torch.bfloat16 else dtype, "fprop+add+relu")
    if cout % engine.vec_of(dtype) == 0 and not (kind == "conv" and st == 2 and any(d % 2 for d in dims)):
        g = _rt(torch.randn_like(yr), dtype)
        yr.backward(g)
        dx = op.dgrad(_cl(g).cuda().to(dtype), dims, out_dtype=torch.float32)
        torch.cuda.synchronize()
        _close(dx.cpu()[..., :cin], _cl(xr.grad), dtype, "dgrad")
        if k == 3:  # weight gradient: the 4 x 16-voxel halo steps (bf16, Cin = 128) or the im2col-order kernel
            wr = w.clone().requires_grad_(True)
            F.conv3d(x, wr, None, stride=st, padding=pad).backward(g)
            dw = torch.zeros(wshape, device="cuda")
            db = torch.zeros(cout, device="cuda")
            op.wgrad(xin, _cl(g).cuda().to(dtype), dw, db)
            torch.cuda.synchronize()
            _close(dw.cpu(), wr.grad, dtype, "wgrad")
            _close(db.cpu(), g.sum(dim=(0, 2, 3, 4)), dtype, "bgrad")


def test_linear_as_one_tap_conv():
    _ffi, engine = _ops()
    torch.manual_seed(2)
    M, K, Nn = 1000, 512, 2049
    x = torch.randn(M, K)
    w = torch.randn(Nn, K) * 0.05
    b = torch.randn(Nn)
    for dtype in (torch.float32, torch.bfloat16):
        op = engine.ConvOp("conv", K, Nn, 1, 1, 0, _rt(w, dtype).view(Nn, K, 1, 1, 1).cuda(), b.cuda(), dtype)
        y = op.fprop(_rt(x, dtype).view(1, 1, 1, M, K).cuda().to(dtype), out_dtype=torch.float32, act=_ffi.ACT_GELU)
        _close(y.cpu().view(M, Nn), F.gelu(_rt(x, dtype) @ _rt(w, dtype).t() + b), dtype, "linear+gelu")


def test_batched_weight_repack_matches_single_launches():
    """engine.PackSet (sa_pack_weights_batch: all operands of a network in one launch after the optimizer step) writes exactly what the per-operand
    sa_pack_weights launches write: conv / transposed conv / linear, forward and data-gradient layouts, bf16 and fp32."""
    _ffi, engine = _ops()
    torch.manual_seed(6)
    mk = lambda *shp: (torch.randn(*shp) * 0.1).cuda()
    specs = [("conv", 16, 24, 3, 1, 1, torch.bfloat16), ("conv", 8, 136, 4, 2, 1, torch.float32), ("convT", 24, 16, 4, 2, 1, torch.bfloat16),
             ("conv", 512, 96, 1, 1, 0, torch.bfloat16), ("conv", 64, 40, 1, 1, 0, torch.float32), ("conv", 6, 16, 1, 1, 0, torch.bfloat16)]
    # (the 1x1x1 / Linear operands with multiple-of-four sides take the 16-byte-vector walk of the batched kernel; 6 -> 16 channels does not)
    ops = []
    for kind, cin, cout, k, st, pad, dt in specs:
        w = mk(cout, cin, k, k, k) if kind == "conv" else mk(cin, cout, k, k, k)
        op = engine.ConvOp(kind, cin, cout, k, st, pad, w, mk(cout), dt)
        dims = (1, 1, 40) if k == 1 else (8, 8, 8)
        x = torch.randn(1, *dims, op.cs_in(), device="cuda").to(dt)
        y = op.fprop(x)                                     # lazy single-launch packs: forward ...
        op.dgrad(torch.randn_like(y), dims)                 # ... and data-gradient operands
        ops.append(op)
    ref = []
    for op in ops:
        op.weight = op.weight * 1.5 + 0.01                  # "optimizer step"
        op.invalidate()
    for op in ops:                                          # reference: what the lazy path packs for the new weights
        for plans in op._plans.values():
            op._ensure_packed(plans["fwd"]); op._ensure_packed(plans["dgrad"])
        ref.append({k: e[0].clone() for k, e in op._packs.items()})
        for e in op._packs.values():
            e[0].zero_(); e[1] = None
    ps = engine.PackSet()
    ps.repack(ops)
    torch.cuda.synchronize()
    n = 0
    for op, r in zip(ops, ref):
        for k, e in op._packs.items():
            assert e[1] == op._pack_version()
            assert torch.equal(e[0].view(torch.uint8), r[k].view(torch.uint8)), (op.kind, k[:3])
            n += 1
    assert n >= 8
    ps.repack(ops)   # second call reuses the device table
    torch.cuda.synchronize()
    assert all(torch.equal(e[0].view(torch.uint8), r[k].view(torch.uint8)) for op, r in zip(ops, ref) for k, e in op._packs.items())


@pytest.mark.parametrize("N,dims", [(1, (8, 8, 8)), (2, (6, 10, 12)), (1, (10, 14, 20)), (3, (4, 6, 50))])
def test_first_layer_kernels(N, dims):
    """sa_conv1_fwd / sa_conv1_wgrad (Conv3d(1 -> 128, k4 s2 p1) + ReLU with the taps gathered from the fp32 volume) against torch on bf16-rounded
    operands: ragged tile tails (cell counts that are not multiples of 128 / 256), every border."""
    _ffi, engine = _ops()
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(N + dims[2])
    D, H, W = dims
    x = torch.randn(N, 1, 2 * D, 2 * H, 2 * W)
    w = torch.randn(128, 1, 4, 4, 4) * 0.2
    b = torch.randn(128) * 0.1
    xr, wr = _rt(x, torch.bfloat16), _rt(w, torch.bfloat16)
    ref = F.relu(F.conv3d(xr.double(), wr.double(), b.double(), stride=2, padding=1)).permute(0, 2, 3, 4, 1)
    wpk = wr.view(128, 64).bfloat16().cuda().contiguous()
    y = torch.empty(N, D, H, W, 128, dtype=torch.bfloat16, device="cuda")
    xd, bd = x.cuda().contiguous(), b.cuda()
    _ffi.check(lib.sa_conv1_fwd(_ffi.ptr(xd), _ffi.ptr(wpk), _ffi.ptr(bd), _ffi.ptr(y), N, D, H, W, 128, _ffi.ACT_RELU, st))
    _close(y.float().cpu(), ref.float(), torch.bfloat16, "conv1 fwd")
    y0 = torch.empty_like(y)
    _ffi.check(lib.sa_conv1_fwd(_ffi.ptr(xd), _ffi.ptr(wpk), None, _ffi.ptr(y0), N, D, H, W, 128, _ffi.ACT_NONE, st))
    _close(y0.float().cpu(), F.conv3d(xr.double(), wr.double(), None, stride=2, padding=1).permute(0, 2, 3, 4, 1).float(), torch.bfloat16, "conv1 fwd (no bias / act)")
    # weight / bias gradient
    g = torch.randn(N, D, H, W, 128)
    gr = _rt(g, torch.bfloat16)
    wl = wr.double().clone().requires_grad_(True)
    bl = b.double().clone().requires_grad_(True)
    (F.conv3d(xr.double(), wl, bl, stride=2, padding=1).permute(0, 2, 3, 4, 1) * gr.double()).sum().backward()
    dw = torch.zeros(128, 64, device="cuda")
    db = torch.zeros(128, device="cuda")
    gd = gr.bfloat16().cuda().contiguous()
    _ffi.check(lib.sa_conv1_wgrad(_ffi.ptr(xd), _ffi.ptr(gd), _ffi.ptr(dw), _ffi.ptr(db), N, D, H, W, 128, st))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dw.cpu().numpy(), wl.grad.view(128, 64).float().numpy(), rtol=2e-3, atol=2e-3 * float(wl.grad.abs().max()))
    np.testing.assert_allclose(db.cpu().numpy(), bl.grad.float().numpy(), rtol=2e-3, atol=2e-3 * float(bl.grad.abs().max()))
    assert lib.sa_conv1_fwd(_ffi.ptr(xd), _ffi.ptr(wpk), None, _ffi.ptr(y0), N, D, H, W, 64, _ffi.ACT_NONE, st) == _ffi.SA_EUNSUPPORTED


# ----------------------------------------------------------------------------------------------- quantizer
def _vq_run(x_rows, cb, decay=None, N=None, avg=None):
    _ffi, _ = _ops()
    lib, st = _ffi.lib(), _ffi.stream()
    M, D = x_rows.shape
    K = cb.shape[0]
    dev = "cuda"
    rows = x_rows.to(dev).contiguous()
    cbd = cb.to(dev).contiguous()
    idx = torch.empty(M, dtype=torch.int64, device=dev)
    zq = torch.empty_like(rows)
    counts = torch.zeros(K, device=dev)
    dw = torch.zeros(K, D, device=dev)
    sq = torch.zeros(1, device=dev)
    wn = torch.zeros(K, device=dev)
    _ffi.check(lib.sa_vq_assign(_ffi.ptr(rows), _ffi.ptr(cbd), M, K, D, _ffi.ptr(idx), _ffi.ptr(zq), None, _ffi.ptr(counts), _ffi.ptr(dw), _ffi.ptr(sq),
                                _ffi.ptr(wn), st))
    out = dict(idx=idx, zq=zq, counts=counts, dw=dw, sq=sq)
    if decay is not None:
        Nd, avgd = N.to(dev), avg.to(dev)
        _ffi.check(lib.sa_vq_ema_update(_ffi.ptr(Nd), _ffi.ptr(avgd), _ffi.ptr(cbd), _ffi.ptr(counts), _ffi.ptr(dw), K, D, decay, 1e-5, st))
        out.update(N=Nd, avg=avgd, cb=cbd)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}


def test_vq_against_reference_golden():
    g = load_golden("quantizer")
    cb = torch.from_numpy(g["W0"].copy())
    K, D = cb.shape
    N = torch.zeros(K)
    avg = cb.clone()
    x = torch.from_numpy(np.transpose(g["eval/x"], (0, 2, 3, 4, 1)).reshape(-1, D).copy())
    r = _vq_run(x, cb)
    assert np.array_equal(r["idx"].numpy().reshape(g["eval/idx"].shape), g["eval/idx"])  # bit-exact code indices
    np.testing.assert_allclose(0.25 * r["sq"].item() / x.numel(), g["eval/loss"], rtol=1e-5)
    for s in range(3):
        x = torch.from_numpy(np.transpose(g[f"train{s}/x"], (0, 2, 3, 4, 1)).reshape(-1, D).copy())
        r = _vq_run(x, cb, 0.5, N, avg)
        assert np.array_equal(r["idx"].numpy().reshape(g[f"train{s}/idx"].shape), g[f"train{s}/idx"]), s
        np.testing.assert_allclose(r["zq"].numpy(), np.transpose(g[f"train{s}/zq"], (0, 2, 3, 4, 1)).reshape(-1, D), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(r["N"].numpy(), g[f"train{s}/N"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r["avg"].numpy(), g[f"train{s}/embed_avg"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r["cb"].numpy(), g[f"train{s}/weight"], rtol=2e-5, atol=1e-6)
        cb, N, avg = r["cb"], r["N"], r["avg"]


@pytest.mark.parametrize("M,K,D", [(1, 16, 4), (17, 40, 8), (1400, 2048, 32), (333, 256, 256), (11200, 2048, 32),
                                   (100, 72, 64), (50, 48, 128), (40, 33, 320)])     # (round 6: every register-prefetch width of vq_assign_kernel<NV>: 1, 1, 2, 16, 2, 4, 8, 32)
def test_vq_against_c_oracle(M, K, D):
    from test_oracle_vs_golden import c_vq_assign
    torch.manual_seed(M + K)
    x = torch.randn(M, D)
    cb = torch.randn(K, D)
    cb[K // 3] = cb[K // 3 + 1] if K > 4 else cb[K // 3]  # exact duplicate code: first index must win
    r = _vq_run(x, cb)
    idx, counts, dw, se, gap = c_vq_assign(x.numpy(), cb.numpy())
    safe = gap > 1e-4  # rows whose top-2 margin is above fp32 rounding noise must agree exactly
    assert np.array_equal(r["idx"].numpy()[safe], idx[safe])
    assert (r["idx"].numpy() != idx).mean() < 1e-3
    assert not np.any(r["idx"].numpy() == K // 3 + 1) or K <= 4
    if np.array_equal(r["idx"].numpy(), idx):
        np.testing.assert_allclose(r["counts"].numpy(), counts)
        np.testing.assert_allclose(r["dw"].numpy(), dw, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r["sq"].item(), se, rtol=1e-4)


def test_vq_statistics_of_a_collapsed_codebook():
    """Round 6: rows of a block that choose the same code are summed in the block and leave as ONE atomic per (code, dimension).  A codebook with three codes in use
    (what synthetic volumes train to) puts whole blocks on one code; ragged last block; counts / dw / commitment error against the C oracle."""
    from test_oracle_vs_golden import c_vq_assign
    torch.manual_seed(7)
    M, K, D = 2003, 256, 32
    cb = torch.randn(K, D) * 10.0
    centers = cb[torch.tensor([5, 77, 200])]
    x = centers[torch.randint(0, 3, (M,))] + 0.01 * torch.randn(M, D)
    x[:64] = centers[0] + 0.01 * torch.randn(64, D)          # four whole blocks on one code
    r = _vq_run(x, cb)
    idx, counts, dw, se, _gap = c_vq_assign(x.numpy(), cb.numpy())
    assert np.array_equal(r["idx"].numpy(), idx) and set(np.unique(idx)) == {5, 77, 200}
    np.testing.assert_allclose(r["counts"].numpy(), counts)
    np.testing.assert_allclose(r["dw"].numpy(), dw, rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(r["sq"].item(), se, rtol=1e-4)


def test_vq_misc_kernels():
    _ffi, _ = _ops()
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(0)
    K, D, M = 64, 8, 100
    counts = torch.randint(0, 5, (K,)).float()
    Mtot = int(counts.sum().item())
    out = torch.zeros(1, device="cuda")
    counts_d = counts.cuda()
    _ffi.check(lib.sa_vq_perplexity(_ffi.ptr(counts_d), K, Mtot, _ffi.ptr(out), st))
    p = counts / Mtot
    np.testing.assert_allclose(out.item(), torch.exp(-(p * torch.log(p + 1e-10)).sum()).item(), rtol=1e-5)
    cb = torch.randn(K, D)
    idx = torch.randint(0, K, (M,))
    rows = torch.randn(M, D)
    gz = torch.randn(M, D)
    gl = torch.tensor([0.7])
    dz = torch.empty(M, D, device="cuda")
    rows_d, cb_d, idx_d, gz_d, gl_d = rows.cuda(), cb.cuda(), idx.cuda(), gz.cuda(), gl.cuda()  # keep device buffers alive across the launch
    _ffi.check(lib.sa_vq_backward(_ffi.ptr(rows_d), _ffi.ptr(cb_d), _ffi.ptr(idx_d), _ffi.ptr(gz_d), 0, _ffi.ptr(gl_d), 0.25, M, D,
                                  _ffi.ptr(dz), 0, st))
    ref = gz + 0.7 * 0.25 * 2 * (rows - cb[idx]) / (M * D)
    np.testing.assert_allclose(dz.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    e = torch.empty(M, D, device="cuda")
    _ffi.check(lib.sa_vq_embed(_ffi.ptr(cb_d), _ffi.ptr(idx_d), M, K, D, _ffi.ptr(e), 0, st))
    assert torch.equal(e.cpu(), cb[idx])


def test_elementwise_kernels():
    _ffi, engine = _ops()
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(3)
    a, b = torch.randn(10007), torch.randn(10007)
    ls = torch.zeros(1, device="cuda")
    gr = torch.empty(10007, device="cuda")
    a_d, b_d = a.cuda(), b.cuda()
    _ffi.check(lib.sa_mse(_ffi.ptr(a_d), _ffi.ptr(b_d), a.numel(), _ffi.ptr(ls), _ffi.ptr(gr), 1.0, st))
    np.testing.assert_allclose(ls.item() / a.numel(), F.mse_loss(a, b).item(), rtol=1e-5)
    np.testing.assert_allclose(gr.cpu().numpy(), (2 * (a - b) / a.numel()).numpy(), rtol=1e-5, atol=1e-9)
    # Adam against torch.optim.Adam, 3 steps
    p = torch.randn(5000)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1.65e-4)
    pd, m, v = p.cuda(), torch.zeros(5000, device="cuda"), torch.zeros(5000, device="cuda")
    for step in range(1, 4):
        g = torch.randn(5000)
        pr.grad = g.clone()
        opt.step()
        g_d = g.cuda()
        _ffi.check(lib.sa_adam(_ffi.ptr(pd), _ffi.ptr(g_d), _ffi.ptr(m), _ffi.ptr(v), 5000, 1.65e-4, 0.9, 0.999, 1e-8, 0.0, step, 1.0, st))
        torch.cuda.synchronize()
    np.testing.assert_allclose(pd.cpu().numpy(), pr.detach().numpy(), rtol=1e-5, atol=1e-7)
    # the 16-byte-vector form: slices at every misalignment (optimizer-in-backward ranges of the flat buffers) and odd lengths give the bits of one whole call
    n = 40013
    p0, g0 = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    whole = [p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    _ffi.check(lib.sa_adam(_ffi.ptr(whole[0]), _ffi.ptr(g0), _ffi.ptr(whole[1]), _ffi.ptr(whole[2]), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 0.5, st))
    parts = [p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    cuts = [0, 1, 3, 6, 10, 4099, 4100, 20001, 20002, 20003, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        _ffi.check(lib.sa_adam(_ffi.ptr(parts[0][lo:]), _ffi.ptr(g0[lo:]), _ffi.ptr(parts[1][lo:]), _ffi.ptr(parts[2][lo:]), hi - lo, 1e-3, 0.9, 0.999, 1e-8, 0.01,
                               1, 0.5, st))
    torch.cuda.synchronize()
    for a_, b_ in zip(whole, parts):
        assert torch.equal(a_, b_)
    ref = torch.optim.Adam([torch.nn.Parameter(p0.clone())], lr=1e-3, weight_decay=0.01)
    ref.param_groups[0]["params"][0].grad = g0 * 0.5
    ref.step()
    np.testing.assert_allclose(whole[0].cpu().numpy(), ref.param_groups[0]["params"][0].detach().cpu().numpy(), rtol=1e-5, atol=1e-7)
    # cast + channel pad
    x = torch.randn(37, 3)
    y = engine.cast_pad(x.cuda(), torch.bfloat16, 8)
    assert y.shape == (37, 8) and torch.equal(y[:, :3].float().cpu(), x.bfloat16().float()) and float(y[:, 3:].abs().max()) == 0.0
    # no padding: the flat vectorised conversion (round to nearest even, like torch), including specials
    x = torch.randn(1001, 24)
    x[0, :4] = torch.tensor([float("inf"), float("-inf"), 0.0, -0.0])
    x[1, 0] = float("nan")
    y = engine.cast_pad(x.cuda(), torch.bfloat16, 24).cpu()
    ref = x.bfloat16()
    ok = torch.isnan(ref) & torch.isnan(y) | (y.view(torch.int16) == ref.view(torch.int16))
    assert bool(ok.all())


def test_abi_rejects_bad_arguments():
    _ffi, engine = _ops()
    lib = _ffi.lib()
    assert lib.sa_conv_fprop(None, 0, None, None, None, None, None) == -1
    assert lib.sa_vq_assign(None, None, 0, 0, 0, None, None, None, None, None, None, None, None) == -1
    g = _ffi.ConvGeom()
    ep = _ffi.Epilogue()
    x = torch.zeros(8, device="cuda")
    assert lib.sa_conv_fprop(ctypes.byref(g), 7, _ffi.ptr(x), _ffi.ptr(x), _ffi.ptr(x), ctypes.byref(ep), None) == -2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_convt1_direct_kernels(dtype):
    """ConvTranspose3d(128 -> 1, k4 s2 p1) direct kernels against torch CPU (forward, dgrad with relu mask, wgrad, bias grad)."""
    _ffi, engine = _ops()
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(7)
    N, dims = 2, (3, 5, 4)
    x = _rt(torch.randn(N, 128, *dims), dtype)
    w = torch.randn(128, 1, 4, 4, 4) * 0.1
    b = torch.tensor([0.3])
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2, padding=1)
    g = torch.randn_like(yr)
    yr.backward(g)
    xd = _cl(x).cuda().to(dtype)
    wd, bd = w.cuda(), b.cuda()
    out = torch.empty(N, *[2 * d for d in dims], device="cuda")
    _ffi.check(lib.sa_convt1_fwd(_ffi.ptr(xd), _ffi.dtype_id(dtype), _ffi.ptr(wd), _ffi.ptr(bd), _ffi.ptr(out), N, *dims, 128, st))
    _close(out.cpu(), yr.detach()[:, 0], dtype, "convt1 fwd")
    gd = g[:, 0].contiguous().cuda()
    dx = torch.empty_like(xd)
    dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
    _ffi.check(lib.sa_convt1_bwd(_ffi.ptr(xd), _ffi.dtype_id(dtype), _ffi.ptr(wd), _ffi.ptr(gd), _ffi.ptr(xd), _ffi.ptr(dx), _ffi.ptr(dw), _ffi.ptr(db), N, *dims, 128, st))
    torch.cuda.synchronize()
    _close(dx.float().cpu(), _cl(xr.grad * (x > 0)), torch.bfloat16 if dtype == torch.bfloat16 else dtype, "convt1 dgrad")
    _close(dw.cpu(), wr.grad, dtype, "convt1 wgrad")
    _close(db.cpu(), br.grad, dtype, "convt1 bgrad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_convt1_gemm_route(dtype):
    """The same layer at a size that takes the GEMM route: taps-as-channels 1x1x1 convolution on the MFMA kernels + sa_convt1_gather, and
    sa_convt1_im2col + the 1x1x1 dgrad / wgrad, through the stage object the network uses (odd sizes: every boundary case of the taps)."""
    import torch.nn as nn
    from synthanatomy_amd.networks.vqvae.baseline import _ConvT1Stage
    torch.manual_seed(11)
    N, dims = 2, (7, 18, 17)      # 4284 cells >= GEMM_MIN_CELLS
    mod = nn.ConvTranspose3d(128, 1, 4, 2, 1)
    x = _rt(torch.relu(torch.randn(N, 128, *dims)), dtype)
    wq = _rt(mod.weight.detach(), dtype)          # the MFMA route rounds the weights to the compute dtype like every other layer
    xr, wr, br = x.clone().requires_grad_(True), wq.clone().requires_grad_(True), mod.bias.detach().clone().requires_grad_(True)
    yr = F.conv_transpose3d(xr, wr, br, stride=2, padding=1)
    g = torch.randn_like(yr)
    yr.backward(g)
    mod = mod.cuda()
    st = _ConvT1Stage(mod, in_act=True, dtype=dtype)
    assert st._gemm(torch.empty(N, *dims, 128))
    xd = _cl(x).cuda().to(dtype)
    tape = []
    out = st.fwd(xd, tape)
    _close(out[..., 0].cpu(), yr.detach()[:, 0], dtype, "convt1 fwd (gemm)")

    class _G:
        def __init__(self):
            self.b = {id(mod.weight): torch.zeros_like(mod.weight), id(mod.bias): torch.zeros_like(mod.bias)}
        def buf(self, p):
            return self.b[id(p)]
        def done(self, *ps):
            pass
    gr = _G()
    dx = st.bwd(_cl(g).cuda(), tape[0], gr)
    torch.cuda.synchronize()
    _close(dx.float().cpu(), _cl(xr.grad * (x > 0)), torch.bfloat16 if dtype == torch.bfloat16 else dtype, "convt1 dgrad (gemm)")
    _close(gr.buf(mod.weight).cpu(), wr.grad, dtype, "convt1 wgrad (gemm)")
    _close(gr.buf(mod.bias).cpu(), br.grad, dtype, "convt1 bgrad (gemm)")


@pytest.mark.parametrize("dims", [(2, 9, 10, 11), (1, 33, 45, 47)])
def test_conv1x1_backward_fused(dims):
    """sa_conv1x1_backward: dw, db and the ReLU-masked data gradient of a 1x1x1 128 -> 128 bf16 convolution from one pass over the tiles,
    against torch autograd (row counts that are not multiples of the 64-row chunk)."""
    _ffi, engine = _ops()
    torch.manual_seed(dims[1])
    dt = torch.bfloat16
    N, sp = dims[0], dims[1:]
    w = _rt(torch.randn(128, 128, 1, 1, 1) * 0.08, dt)
    b = torch.randn(128) * 0.1
    x = _rt(torch.relu(torch.randn(N, 128, *sp)), dt)          # post-ReLU input: its zeros are the mask
    g = _rt(torch.randn(N, 128, *sp), dt)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.conv3d(xr, wr, br).backward(g)
    op = engine.ConvOp("conv", 128, 128, 1, 1, 0, w.cuda(), b.cuda(), dt)
    dw, db = torch.zeros_like(w, device="cuda"), torch.zeros(128, device="cuda")
    dx = engine.conv1x1_backward(op, _cl(x).cuda().to(dt), _cl(g).cuda().to(dt), dw, db)
    assert dx is not None and dx.dtype == dt
    torch.cuda.synchronize()
    _close(dx.float().cpu(), _cl(xr.grad * (x > 0)), dt, "masked dgrad")
    _close(dw.cpu(), wr.grad, dt, "wgrad")
    _close(db.cpu(), br.grad, dt, "bgrad")


def _rel_t(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize("shape,in_act", [((2, 8, 10, 12), True), ((1, 20, 28, 20), True), ((3, 5, 7, 9), False)])
def test_last_layer_backward_on_the_first_layer_kernels(shape, in_act):
    """sa_convt1_backward (data gradient = conv1 forward on the gradient volume, weight gradient = conv1 weight gradient with the layer input as
    `g`, bias gradient = sum of the volume) against the im2col + GEMM route it replaces and against torch autograd on the same bf16-rounded operands."""
    from synthanatomy_amd import debug
    from synthanatomy_amd.networks.vqvae.baseline import _ConvT1Stage, _GradCtx
    torch.manual_seed(sum(shape))
    mod = torch.nn.ConvTranspose3d(128, 1, 4, 2, 1).cuda()
    st = _ConvT1Stage(mod, in_act=in_act, dtype=torch.bfloat16)
    st.GEMM_MIN_CELLS = 1
    N, D, H, W = shape
    x = torch.randn(N, D, H, W, 128, device="cuda")
    if in_act:
        x = torch.relu(x)
    x = x.to(torch.bfloat16)
    G = torch.randn(N, 2 * D, 2 * H, 2 * W, 1, device="cuda")
    res = []
    for fused in (True, False):
        with debug.override(no_convt1_fused_bwd=not fused):
            gc = _GradCtx(None)
            dx = st.bwd(G, (x,), gc)
            torch.cuda.synchronize()
            res.append((dx.float().clone(), gc.grads[mod.weight].clone(), gc.grads[mod.bias].clone()))
    # torch reference on the bf16-rounded operands (the gathered gradient taps are rounded to bf16 by both routes)
    xr = x.float().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    wr = mod.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = mod.bias.detach().clone().requires_grad_(True)
    Gr = G.to(torch.bfloat16).float().permute(0, 4, 1, 2, 3)
    y = torch.nn.functional.conv_transpose3d(xr, wr, br, stride=2, padding=1)
    (y * Gr).sum().backward()
    dx_ref = xr.grad.permute(0, 2, 3, 4, 1)
    if in_act:
        dx_ref = dx_ref * (x.float() > 0)
    for dx, dw, db in res:
        assert _rel_t(dx, dx_ref) < 1e-2 and _rel_t(dw, wr.grad) < 5e-3
        assert abs(float(db) - float(G.sum())) < 1e-3 * float(G.abs().sum()) ** 0.5 + 1e-2
    assert _rel_t(res[0][0], res[1][0]) < 8e-3 and _rel_t(res[0][1], res[1][1]) < 2e-3
    # the wgrad-only form used by the adaptive adversarial weight
    gc = _GradCtx(None)
    assert st.bwd(G, (x,), gc, wgrad_only=True) is None
    assert _rel_t(gc.grads[mod.weight], res[0][1]) < 1e-5


@pytest.mark.parametrize("K,N,R", [(2048, 512, 8192), (1024, 512, 8100), (3072, 512, 8192), (1088, 512, 4100)])
def test_dense_layer_two_k_groups_equal_one_group(K, N, R):
    """conv_fprop_dma_kernel<..., 2> (round 4: about one 128 x 128 tile per CU and a long reduction -> two groups of eight waves reduce the two halves of K and
    group 1 hands its accumulators over through LDS) against the one-group kernel on the same operands -- same products, one fp32 addition in a different
    place -- and against torch; with the dense-layer epilogue extras (bias, GELU, pre-activation copy, bf16 copy, fp32 residual, ReZero gate)."""
    from synthanatomy_amd import _ffi, debug, engine
    torch.manual_seed(K + N)
    w = (torch.randn(N, K, 1, 1, 1) * K ** -0.5).to(torch.bfloat16).float().cuda()
    b = (torch.randn(N) * 0.1).cuda()
    op = engine.ConvOp("conv", K, N, 1, 1, 0, w, b, torch.bfloat16)
    x = torch.randn(1, 1, 1, R, K).to(torch.bfloat16).cuda()
    res = torch.randn(1, 1, 1, R, N).cuda()
    gate = torch.tensor([0.37], device="cuda")
    outs = {}
    for one_group in (False, True):
        with debug.override(no_kgroups=one_group):
            y = op.fprop(x, out_dtype=torch.float32)
            kern = _ffi.lib().sa_last_conv_kernel().decode()
            y2, pre, lp = op.fprop(x, act=_ffi.ACT_GELU, alpha=gate, addend=res, out_dtype=torch.float32, want_pre=True, want_lp=True)
        outs[one_group] = (y, y2, pre, lp, kern)
    assert outs[False][4].endswith("true, false, 2>") and outs[True][4].endswith("true, false, 1>"), (outs[False][4], outs[True][4])
    ref = torch.nn.functional.linear(x.view(R, K).float().cpu(), w.view(N, K).cpu(), b.cpu())
    for a_, b_ in zip(outs[False][:4], outs[True][:4]):
        assert float((a_.float() - b_.float()).abs().max()) <= 2e-5 * float(b_.float().abs().max()) + (8e-3 * float(b_.float().abs().max()) if a_.dtype == torch.bfloat16 else 0.0)
    _close(outs[False][0].view(R, N).cpu(), ref, torch.bfloat16, "two K groups vs torch")
    ref2 = torch.nn.functional.gelu(ref) * 0.37 + res.view(R, N).cpu()
    _close(outs[False][1].view(R, N).cpu(), ref2, torch.bfloat16, "two K groups + epilogue vs torch")
    _close(outs[False][2].float().view(R, N).cpu(), ref, torch.bfloat16, "pre-activation copy")
