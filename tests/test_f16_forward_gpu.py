"""GPU: the float16 FORWARD operand type of the VQ-VAE encoder chain (round 4) -- the reference trains with fp16 autocast (src/engines/trainer.py:161-163,
run_vqvae.py --amp=True); here the encoder's forward launches take IEEE-half activations and weights (csrc/conv_fprop_f16.hip, conv1_fwd_f16_kernel) and
write, next to their f16 output, the bf16 copy the bf16 backward pass reads (sa_epilogue.out_lp).  Every kernel against torch fp32 on operands rounded
to the kernel's operand type; the encoder against the fp32 product path (indices) and against the bf16-forward encoder (gradients)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

H16, BF = torch.float16, torch.bfloat16


def _cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))


@pytest.mark.parametrize("kind,cin,cout,k,s,p,dims", [
    ("conv", 32, 40, 4, 2, 1, (8, 10, 12)),        # strided, im2col-order kernel, partial channel tile (LDS epilogue)
    ("conv", 128, 128, 4, 2, 1, (34, 36, 38)),     # the production down-sampling layer, eight-wave tiles
    ("conv", 128, 128, 3, 1, 1, (17, 32, 48)),     # 3x3x3: 16 x 16-patch halo kernel, register epilogue with the bf16 copy
    ("conv", 64, 136, 3, 1, 1, (9, 24, 32)),       # 8 x 16-patch halo kernel, channel tail
    ("conv", 256, 32, 3, 1, 1, (10, 14, 10)),      # the pre-quantizer layer (fp32 output)
    ("conv", 256, 256, 1, 1, 0, (10, 14, 10)),
])
def test_f16_forward_launches_match_torch(kind, cin, cout, k, s, p, dims):
    from synthanatomy_amd import _ffi, engine
    torch.manual_seed(cin + cout + k)
    N = 2
    w = (torch.randn(cout, cin, k, k, k) * (1.5 / (cin * k ** 3) ** 0.5)).to(H16).float()
    b = torch.randn(cout) * 0.1
    x = torch.randn(N, cin, *dims).to(H16).float()
    op = engine.ConvOp(kind, cin, cout, k, s, p, w.cuda(), b.cuda(), BF, fwd_dtype=H16)
    xin = _cl(x).cuda().to(H16)
    ref = _cl(F.conv3d(x, w, b, stride=s, padding=p))
    # f16 output + bf16 copy, ReLU
    y, _, lp = op.fprop(xin, act=_ffi.ACT_RELU, want_lp=True)
    assert y.dtype == H16 and lp.dtype == BF and y.shape == lp.shape
    r = F.relu(ref)
    assert _rel(y.float().cpu(), r) < 1.5e-3, _rel(y.float().cpu(), r)            # half a ulp of f16 on the largest values + summation order
    assert _rel(lp.float().cpu(), r) < 6e-3
    assert torch.equal(lp.float(), y.float().to(BF).float()) or _rel(lp.float(), y.float()) < 4.5e-3   # the copy is the same fp32 value rounded to bf16
    # fp32 output (what feeds the quantizer)
    y32 = op.fprop(xin, out_dtype=torch.float32)
    assert _rel(y32.cpu(), ref) < 2e-5, _rel(y32.cpu(), ref)
    # residual addend in f16 (add before the ReLU), no copy
    if cout % 8 == 0 and s == 1:
        add = torch.randn(N, *dims, cout).to(H16)
        y3 = op.fprop(xin, act=_ffi.ACT_RELU, addend=add.cuda(), add_before_act=True)
        assert _rel(y3.float().cpu(), F.relu(ref + add.float())) < 1.5e-3
    # the backward operands of the same op stay bf16
    g = torch.randn(N, *ref.shape[1:4], cout).to(BF)
    if not (s == 2 and any(d % 2 for d in dims)):
        dx = op.dgrad(g.cuda(), dims, out_dtype=torch.float32)
        xr = x.clone().requires_grad_(True)
        F.conv3d(xr, w.to(BF).float(), None, stride=s, padding=p).backward(g.float().permute(0, 4, 1, 2, 3))
        assert _rel(dx.cpu(), _cl(xr.grad)) < 1.2e-2


@pytest.mark.parametrize("dims", [(17, 32, 48), (9, 24, 32), (5, 12, 14)])   # 16 x 16-patch kernel / 8 x 16-patch kernel / im2col-order fused kernel
def test_f16_fused_residual_block(dims):
    """relu(x + conv1(relu(conv3(x) + b1)) + b2) in one launch on f16 operands: y (f16), its bf16 copy and the bf16 hidden tensor of the backward pass."""
    from synthanatomy_amd.networks.vqvae.baseline import ResidualLayer, _Act, _ResStage
    torch.manual_seed(dims[0])
    C, N = 128, 2
    mod = ResidualLayer(C, C, 0.0)
    with torch.no_grad():
        for q in mod.parameters():
            q.copy_(q.to(H16).float())
    x = F.relu(torch.randn(N, C, *dims)).to(H16)
    xc = _cl(x)
    st = _ResStage(mod.cuda(), in_act=True, dtype=BF, fwd_dtype=H16)
    tape = []
    out = st.fwd(_Act(xc.cuda(), xc.cuda().to(BF)), tape)
    assert isinstance(out, _Act) and out.f.dtype == H16 and out.s.dtype == BF
    xs, h = tape[0]
    assert xs.dtype == BF and h.dtype == BF
    xf = x.float()
    hr = F.relu(F.conv3d(xf, mod[0].weight.cpu(), mod[0].bias.cpu(), padding=1))
    yr = F.relu(xf + F.conv3d(hr.to(H16).float(), mod[3].weight.cpu(), mod[3].bias.cpu()))
    assert _rel(h.float().cpu(), _cl(hr)) < 6e-3
    assert _rel(out.f.float().cpu(), _cl(yr)) < 2e-3, _rel(out.f.float().cpu(), _cl(yr))
    assert _rel(out.s.float().cpu(), _cl(yr)) < 6e-3
    # eval: no copies
    out2 = st.fwd(_Act(xc.cuda(), None), None)
    assert out2.s is None and torch.equal(out2.f, out.f)


def test_f16_first_layer():
    from synthanatomy_amd import _ffi, engine
    torch.manual_seed(5)
    N, D, H, W = 2, 10, 12, 16
    w = (torch.randn(128, 1, 4, 4, 4) * 0.2).to(H16).float()
    b = torch.randn(128) * 0.1
    x = torch.rand(N, 1, 2 * D, 2 * H, 2 * W)
    op = engine.ConvOp("conv", 64, 128, 1, 1, 0, w.view(128, 64, 1, 1, 1).cuda(), b.cuda(), BF, fwd_dtype=H16)
    wpk = op.packed_fwd_operand(N, (D, H, W))
    assert wpk.dtype == H16
    y = torch.empty(N, D, H, W, 128, dtype=H16, device="cuda")
    ys = torch.empty(N, D, H, W, 128, dtype=BF, device="cuda")
    xd = x.cuda().view(N, 2 * D, 2 * H, 2 * W).contiguous()
    _ffi.check(_ffi.lib().sa_conv1_fwd_f16(_ffi.ptr(xd), _ffi.ptr(wpk), _ffi.ptr(b.cuda()), _ffi.ptr(y), _ffi.ptr(ys), N, D, H, W, 128, _ffi.ACT_RELU, _ffi.stream()))
    ref = _cl(F.relu(F.conv3d(x.to(H16).float(), w, b, stride=2, padding=1)))
    assert _rel(y.float().cpu(), ref) < 1.5e-3
    assert _rel(ys.float().cpu(), ref) < 6e-3


NET = dict(n_levels=4, downsample_parameters=((4, 2, 1, 1),) * 4, upsample_parameters=((4, 2, 1, 0, 1),) * 4, n_embed=2048, embed_dim=32, n_channels=256,
           n_res_channels=256, n_res_layers=3)


def test_encoder_chain_f16_forward_tracks_fp32_and_trains_like_bf16():
    """Config-2 widths on a 64 x 96 x 64 crop.  (i) z of the f16-forward encoder is several times closer to the fp32 encoder than the bf16-forward one;
    (ii) one training step: same kernels in the backward pass, gradients agree with the bf16-forward network to bf16 noise; (iii) eval forward ==
    training forward (the bf16 copies do not change the f16 stream)."""
    from synthanatomy_amd import _ffi
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    torch.manual_seed(4)
    ref = BaselineVQVAE(**NET, compute_dtype=torch.float32).cuda().eval()
    sd = ref.state_dict()
    n16 = BaselineVQVAE(**NET, compute_dtype=BF).cuda()
    nbf = BaselineVQVAE(**NET, compute_dtype=BF, encoder_forward_dtype=BF).cuda()
    assert n16.encoder_forward_dtype == H16 and nbf.encoder_forward_dtype == BF
    n16.load_state_dict(sd)
    nbf.load_state_dict(sd)
    x = torch.rand(2, 1, 64, 96, 64, generator=torch.Generator().manual_seed(9)).cuda()
    with torch.no_grad():
        z32 = ref.encode(x)[0].float()
        _ffi.lib().sa_kernel_log_begin()
        z16 = n16.eval().encode(x)[0].float()
        buf = ctypes.create_string_buffer(1 << 14)
        _ffi.lib().sa_kernel_log_read(buf, len(buf), 1)
        log = buf.value.decode()
        zbf = nbf.eval().encode(x)[0].float()
    assert "conv1_fwd_f16_kernel" in log and "conv_fprop_halo256_kernel<f16_t, true, 8, " in log and "unsigned short" not in log, log
    e16, ebf = _rel(z16, z32), _rel(zbf, z32)
    print(f"[f16 forward] z vs fp32 encoder: f16 {e16:.2e}, bf16 {ebf:.2e}")
    assert e16 < 1.5e-3 and e16 < 0.35 * ebf, (e16, ebf)
    # encoder backward under a FIXED random upstream gradient (no quantizer in between), against the fp32 product path: thirty layers of ReLU masks make this a
    # harsh comparison for any 16-bit forward (measured: bf16 forward operands 0.13-0.17 Frobenius-relative on the first level, f16 forward operands 0.05) --
    # the f16-forward chain saves bf16 copies and runs the SAME bf16 backward kernels, and must be the closer of the two on every parameter group
    gz = torch.randn(z16.shape, generator=torch.Generator().manual_seed(3)).cuda()
    grads = {}
    for name, net in (("fp32", ref), ("f16", n16), ("bf16", nbf)):
        net.train()
        z = net.encode(x)[0]
        if name == "f16":
            assert _rel(z.float().detach(), z16) == 0.0       # recording forward == eval forward (the bf16 copies do not touch the f16 stream)
        (z.float() * gz).sum().backward()
        torch.cuda.synchronize()
        grads[name] = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
        net.zero_grad(set_to_none=True)
    assert grads["f16"].keys() == grads["bf16"].keys() == grads["fp32"].keys() and len(grads["f16"]) >= 55

    def fro(a, b):
        return float((a - b).norm() / (b.norm() + 1e-20))
    d16 = {k: fro(grads["f16"][k], grads["fp32"][k]) for k in grads["fp32"]}
    dbf = {k: fro(grads["bf16"][k], grads["fp32"][k]) for k in grads["fp32"]}
    w16, wbf = max(d16.values()), max(dbf.values())
    print(f"[f16 forward] worst encoder-gradient deviation from the fp32 path: f16 forward {w16:.2e}, bf16 forward {wbf:.2e}")
    assert w16 < 8e-2 and w16 < 0.6 * wbf, (w16, wbf)
    assert sum(d16[k] <= dbf[k] * 1.05 + 1e-3 for k in d16) >= len(d16) - 2
    # and one whole training step runs (quantizer + decoder behind the f16 encoder): finite loss, every parameter receives a gradient
    out = n16(x)
    loss = F.mse_loss(out["reconstruction"][0].float(), x) + out["quantization_losses"][0]
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in n16.parameters() if p.requires_grad)


def test_f16_forward_network_falls_back_to_bf16_chain_when_dma_is_unavailable():
    """The f16 forward instances are LDS-DMA kernels (32-bit buffer offsets).  With operands the DMA cannot address (>= 4 GiB: first-level activations at
    batch ~24; here forced with the no_dma switch) the default throughput-mode network must keep training on the all-bf16 encoder chain over the SAME
    parameters -- identical to a network built with encoder_forward_dtype=bfloat16 under the same switch -- instead of raising SA_EUNSUPPORTED."""
    from synthanatomy_amd import debug
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    small = dict(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=64, embed_dim=16, n_channels=32,
                 n_res_channels=32, n_res_layers=1)
    torch.manual_seed(2)
    n16 = BaselineVQVAE(**small, compute_dtype=BF).cuda().train()
    nbf = BaselineVQVAE(**small, compute_dtype=BF, encoder_forward_dtype=BF).cuda().train()
    nbf.load_state_dict(n16.state_dict())
    x = torch.rand(2, 1, 16, 24, 16, generator=torch.Generator().manual_seed(1)).cuda()
    res = {}
    with debug.override(no_dma=True):
        for name, net in (("f16", n16), ("bf16", nbf)):
            out = net(x)
            loss = F.mse_loss(out["reconstruction"][0].float(), x) + out["quantization_losses"][0]
            loss.backward()
            torch.cuda.synchronize()
            res[name] = (loss.item(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    assert getattr(n16, "_enc_chain_lp", None) is not None
    assert res["f16"][0] == res["bf16"][0]
    for k, g in res["bf16"][1].items():      # (same kernels on the same operands; the weight gradients accumulate with fp32 atomics, so not bit for bit)
        assert torch.allclose(res["f16"][1][k], g, rtol=1e-4, atol=1e-6), k
    # without the switch the f16 chain is back (and differs from the bf16 one)
    with torch.no_grad():
        z16 = n16.eval().encode(x)[0].float()
        zbf = nbf.eval().encode(x)[0].float()
    assert not torch.equal(z16, zbf)
    # the re-pack after an optimizer step covers both chains' operands
    n16.invalidate_packed_weights()


def test_f16_conversion_saturates_finite_values_and_keeps_nan_visible():
    """f32 -> f16 in every epilogue / cast: finite overflow saturates to +-65504 (no loss scaling needed), NaN and +-inf come out as NaN -- a diverged run must
    show in z and the loss (v_med3_f32 alone turned a NaN into -65504, i.e. zero behind the next ReLU)."""
    from synthanatomy_amd import engine
    vals = torch.tensor([[1.0, -2.5, 7e4, -7e4, 65504.0, 3e38, float("nan"), float("inf"), float("-inf"), 0.0, -0.0, 1e-8, 65519.9, 6.1e-5, -1e6, 0.333251953125]])
    out = engine.cast_pad(vals.cuda(), H16, 16).cpu()[0]
    exp = vals[0].clamp(-65504, 65504).to(H16)
    fin = torch.isfinite(vals[0])
    assert torch.equal(out[fin], exp[fin]), (out, exp)
    assert bool(torch.isnan(out[~fin]).all()), out
    # through a launch's epilogue: one NaN input voxel poisons its 3x3x3 neighbourhood in the f16 output AND in the bf16 copy
    op = engine.ConvOp("conv", 16, 16, 3, 1, 1, (torch.randn(16, 16, 3, 3, 3) * 0.05).cuda(), torch.zeros(16).cuda(), BF, fwd_dtype=H16)
    x = torch.randn(1, 6, 8, 8, 16)
    x[0, 3, 4, 4, 5] = float("nan")
    y, _, lp = op.fprop(x.cuda().to(H16), want_lp=True)      # (no activation: the kernels' ReLU is fmaxf(x, 0), which maps NaN to 0 in every dtype)
    assert bool(torch.isnan(y[0, 3, 4, 4].float()).all()) and bool(torch.isnan(lp[0, 3, 4, 4].float()).all())
    assert bool(torch.isfinite(y[0, 0, 0, 0].float()).all())


def test_side_stream_weight_gradients_equal_the_one_stream_backward():
    """VQ-VAE backward with the weight gradients on a second HIP stream (round-5 default in throughput mode) against the same pass on one stream
    (SA_NO_SIDE_WGRAD): same kernels on the same operands, so the gradients agree to the fp32 atomics' ordering noise; twice in a row (the second pass must
    not see stale workspaces or un-joined streams)."""
    from synthanatomy_amd import debug
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    small = dict(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=64, embed_dim=16, n_channels=64,
                 n_res_channels=64, n_res_layers=2)
    torch.manual_seed(3)
    net = BaselineVQVAE(**small, compute_dtype=BF).cuda().train()
    x = torch.rand(2, 1, 32, 48, 32, generator=torch.Generator().manual_seed(2)).cuda()

    def grads():
        net.zero_grad(set_to_none=True)
        out = net(x)
        (F.mse_loss(out["reconstruction"][0].float(), x) + out["quantization_losses"][0]).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    net.eval()      # (no EMA update between the passes: identical forward)
    for p in net.parameters():
        p.requires_grad_(True)
    with debug.override(no_side_wgrad=True):
        ref = grads()
    for _ in range(2):
        got = grads()
        assert got.keys() == ref.keys()
        for k in ref:
            assert torch.allclose(got[k], ref[k], rtol=2e-4, atol=1e-6), (k, float((got[k] - ref[k]).abs().max()))
