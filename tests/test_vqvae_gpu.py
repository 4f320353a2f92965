"""GPU parity: the product ``baseline_vqvae`` (HIP kernels behind the reference plugin surface) against
(a) fixtures computed by the reference itself and (b) the CPU oracle on fresh seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import load_golden, meta_of, state_from_golden  # noqa: E402

REL = 1e-3  # north_star: fp32 recon within 1e-3 relative


def _relerr(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)


def _make(meta, dtype):
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    kw = dict(meta["net_kwargs"])
    kw["downsample_parameters"] = tuple(map(tuple, kw["downsample_parameters"]))
    kw["upsample_parameters"] = tuple(map(tuple, kw["upsample_parameters"]))
    return BaselineVQVAE(**kw, compute_dtype=dtype)


def _load(net, g):
    sd = state_from_golden(g)
    net.load_state_dict(sd)
    return net.cuda()


@pytest.mark.parametrize("name", ["vqvae_cfg1", "vqvae_tiny4"])
def test_eval_paths_match_reference_fixture_fp32(name):
    g = load_golden(name)
    net = _load(_make(meta_of(g), torch.float32), g).eval()
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        z = net.encode(x)[0]
        assert z.shape == tuple(g["eval/z"].shape)
        assert _relerr(z.cpu().numpy(), g["eval/z"]) < REL
        idx = net.index_quantize(x)[0]
        assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), g["eval/idx"])  # bit-exact code indices
        rec = net.decode_samples([torch.from_numpy(g["eval/idx"]).cuda()])
        assert _relerr(rec.cpu().numpy(), g["eval/decode_samples"]) < REL
        out = net(x)
        assert set(out) == {"reconstruction", "quantization_losses"} and len(out["reconstruction"]) == 1
        assert _relerr(out["reconstruction"][0].cpu().numpy(), g["eval/recon"]) < REL
        np.testing.assert_allclose(out["quantization_losses"][0].item(), g["eval/qloss"], rtol=1e-4)
    # eval mode must leave the EMA state untouched (Quantizer_impl.forward only updates when training)
    sd = net.state_dict()
    for k in ("quantizer.0.impl.N", "quantizer.0.impl.embed_avg", "quantizer.0.impl.weight"):
        assert np.array_equal(sd[k].cpu().numpy(), g["sd0/" + k])


@pytest.mark.parametrize("name", ["vqvae_cfg1", "vqvae_tiny4"])
def test_train_steps_match_reference_fixture_fp32(name):
    g = load_golden(name)
    net = _load(_make(meta_of(g), torch.float32), g).train()
    x = torch.from_numpy(g["x"]).cuda()
    for step in (1, 2):
        net.zero_grad()
        out = net(x)
        loss = torch.nn.functional.mse_loss(out["reconstruction"][0].float(), x) + out["quantization_losses"][0]
        loss.backward()
        torch.cuda.synchronize()
        assert _relerr(out["reconstruction"][0].detach().cpu().numpy(), g[f"train{step}/recon"]) < REL
        np.testing.assert_allclose(loss.item(), g[f"train{step}/loss"], rtol=1e-4)
        np.testing.assert_allclose(net.get_perplexity()[0].item(), g[f"train{step}/perplexity"], rtol=1e-4)
        sd = net.state_dict()
        for nm in ("N", "embed_avg", "weight"):
            assert _relerr(sd["quantizer.0.impl." + nm].cpu().numpy(), g[f"train{step}/{nm}"]) < 1e-4, (step, nm)
        if step == 1:
            n = 0
            params = dict(net.named_parameters())
            for k in g.files:
                if k.startswith("train1/grad/"):
                    pk = k[len("train1/grad/"):]
                    assert params[pk].grad is not None, pk
                    assert _relerr(params[pk].grad.cpu().numpy(), g[k]) < 2e-3, pk
                    n += 1
            assert n > 5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_against_oracle_fresh_inputs(dtype):
    """A shape the fixtures do not hold (odd-ish grid, batch 3), weights by seed; oracle emulates bf16 storage rounding."""
    from oracle import vqvae_ref
    cfg = vqvae_ref.VQVAEConfig(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=64,
                                embed_dim=16, n_channels=32, n_res_channels=32, n_res_layers=2)
    st = vqvae_ref.init_state(cfg, seed=9)
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    net = BaselineVQVAE(n_levels=2, downsample_parameters=cfg.downsample_parameters, upsample_parameters=cfg.upsample_parameters, n_embed=64, embed_dim=16,
                        n_channels=32, n_res_channels=32, n_res_layers=2, compute_dtype=dtype)
    net.load_state_dict({k: v.clone() for k, v in st.items()})
    net = net.cuda().eval()
    torch.manual_seed(10)
    x = torch.rand(3, 1, 24, 40, 16)
    rd = None if dtype == torch.float32 else torch.bfloat16
    with torch.no_grad():
        ref = vqvae_ref.forward({k: v.clone() for k, v in st.items()}, cfg, x, training=False, round_dtype=rd,
                                enc_round_dtype=torch.float16 if rd is not None else None)   # throughput mode: encoder forward on float16 operands
        z = net.encode(x.cuda())[0]
        idx = net.index_quantize(x.cuda())[0]
        rec = net.decode_samples([ref["indices"].cuda()])
    tol = REL if dtype == torch.float32 else 2e-2
    assert _relerr(z.cpu().numpy(), ref["z"].numpy()) < tol
    agree = (idx.cpu() == ref["indices"]).float().mean().item()
    assert agree == 1.0 if dtype == torch.float32 else agree > 0.97, agree
    ref_rec = vqvae_ref.decode(st, cfg, vqvae_ref.embed(st, ref["indices"]), round_dtype=rd)
    assert _relerr(rec.cpu().numpy(), ref_rec.numpy()) < tol


def test_batch_independence_and_determinism_bf16():
    """Size-independent properties at a larger grid: samples do not interact; repeated runs are bit-identical."""
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    torch.manual_seed(4)
    net = BaselineVQVAE(n_levels=3, downsample_parameters=((4, 2, 1, 1),) * 3, upsample_parameters=((4, 2, 1, 0, 1),) * 3, n_embed=256, embed_dim=32,
                        n_channels=64, n_res_channels=64, n_res_layers=1).cuda().eval()
    x = torch.rand(2, 1, 48, 64, 32, device="cuda")
    with torch.no_grad():
        a = net(x)["reconstruction"][0]
        b = net(x)["reconstruction"][0]
        s0 = net(x[:1])["reconstruction"][0]
        i_all = net.index_quantize(x)[0]
        i0 = net.index_quantize(x[1:])[0]
    assert a.shape == x.shape and torch.equal(a, b)
    assert torch.equal(a[:1], s0)
    assert torch.equal(i_all[1:], i0)
    assert int(i_all.min()) >= 0 and int(i_all.max()) < 256


def test_training_step_bf16_decreases_loss_and_matches_oracle_grads():
    from oracle import vqvae_ref
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    cfg = vqvae_ref.VQVAEConfig(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=32,
                                embed_dim=16, n_channels=32, n_res_channels=32, n_res_layers=1)
    st = vqvae_ref.init_state(cfg, seed=3)
    net = BaselineVQVAE(n_levels=2, downsample_parameters=cfg.downsample_parameters, upsample_parameters=cfg.upsample_parameters, n_embed=32, embed_dim=16,
                        n_channels=32, n_res_channels=32, n_res_layers=1, compute_dtype=torch.bfloat16)
    net.load_state_dict({k: v.clone() for k, v in st.items()})
    net = net.cuda().train()
    torch.manual_seed(1)
    x = torch.rand(2, 1, 16, 16, 16)
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "quantizer" not in k}
    stt = {k: v.clone() for k, v in st.items()}
    stt.update(leaf)
    ref = vqvae_ref.forward(stt, cfg, x, training=True, round_dtype=torch.bfloat16, enc_round_dtype=torch.float16)
    vqvae_ref.mse_loss(ref, x).backward()
    out = net(x.cuda())
    loss = torch.nn.functional.mse_loss(out["reconstruction"][0], x.cuda()) + out["quantization_losses"][0]
    loss.backward()
    params = dict(net.named_parameters())
    worst = 0.0
    for k, p in leaf.items():
        gr = params[k].grad
        assert gr is not None and torch.isfinite(gr).all(), k
        worst = max(worst, _relerr(gr.cpu().numpy(), p.grad.numpy()))
    assert worst < 8e-2, worst  # bf16 activations/gradients vs the fp32-accumulated oracle
    # ... and the gradients point downhill: a few Adam steps on the same batch lower the loss
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-3)
    first = last = None
    for _ in range(12):
        opt.zero_grad()
        out = net(x.cuda())
        l = torch.nn.functional.mse_loss(out["reconstruction"][0], x.cuda()) + out["quantization_losses"][0]
        l.backward()
        opt.step()
        net.invalidate_packed_weights()
        last = float(l)
        first = last if first is None else first
    assert last < 0.8 * first, (first, last)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_discriminator_matches_reference_fixture(dtype):
    from synthanatomy_amd.networks.discriminator.baseline import BaselineDiscriminator
    g = load_golden("discriminator")
    net = BaselineDiscriminator(input_nc=1, ndf=8, n_layers=3, compute_dtype=dtype)
    sd = state_from_golden(g)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().train()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = net(x)
    tol = REL if dtype == torch.float32 else 3e-2
    assert y.shape == tuple(g["train/logits"].shape) and _relerr(y.detach().cpu().numpy(), g["train/logits"]) < tol
    sd2 = net.state_dict()
    for k in g.files:
        if k.startswith("train/sd/"):
            assert _relerr(sd2[k[len("train/sd/"):]].cpu().numpy(), g[k]) < tol, k
    # gradients against the CPU oracle (training-mode BatchNorm)
    from oracle import vqvae_ref
    st = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in state_from_golden(g).items()}
    xr = torch.from_numpy(g["x"]).requires_grad_(True)
    yr = vqvae_ref.discriminator_forward(st, xr, training=True)
    w = torch.randn(yr.shape, generator=torch.Generator().manual_seed(0))
    (yr * w).sum().backward()
    (y * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    gtol = 3e-3 if dtype == torch.float32 else 0.25  # bf16 through 3 tiny-batch BatchNorms: sanity only, fp32 is the parity gate
    for k, p in net.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        if dtype == torch.float32:  # bf16 through three tiny-batch BatchNorms is a smoke check only; fp32 is the parity gate
            assert _relerr(p.grad.cpu().numpy(), st[k].grad.numpy()) < gtol, k
    if dtype == torch.float32:
        assert _relerr(x.grad.cpu().numpy(), xr.grad.numpy()) < gtol
    net.eval()
    with torch.no_grad():
        ye = net(torch.from_numpy(g["x"]).cuda())
    if dtype == torch.float32:
        # eval uses the running statistics updated by the single training forward above, as in the fixture
        assert _relerr(ye.cpu().numpy(), g["eval/logits"]) < 5e-3


@pytest.mark.parametrize("shape", [(2, 9, 10, 11), (2, 33, 32, 44), (2, 35, 40, 44)])
def test_fused_residual_block_matches_two_launch_path(shape):
    """sa_resblock_fprop (one launch) against the two-launch path of the same stage: y and the stored hidden activation.  The larger
    shapes run on the halo mainloops (16 x 16 patches; two-plane 2 x 8 x 16 tiles with an odd number of planes, ragged in W); their reference path is forced
    onto the im2col-order kernels."""
    from synthanatomy_amd.networks.vqvae.baseline import ResidualLayer, _ResStage
    torch.manual_seed(3)
    mod = ResidualLayer(128, 128, 0.0).cuda()
    st = _ResStage(mod, in_act=True, dtype=torch.bfloat16)
    x = torch.relu(torch.randn(*shape, 128, device="cuda")).to(torch.bfloat16)
    tape = []
    y_f = st.fwd(x, tape)
    h_f = tape[0][1]
    from synthanatomy_amd import _ffi, debug
    if shape[1] == 35:
        assert _ffi.lib().sa_last_conv_kernel().decode() == "conv_fprop_halo256_kernel<unsigned short, true, 8, true>"
    elif shape[1] == 33:
        assert _ffi.lib().sa_last_conv_kernel().decode() == "conv_fprop_halo256_kernel<unsigned short, true, 8, false>"
    with debug.override(no_fused_res=True, no_halo=True):
        tape2 = []
        y_r = st.fwd(x, tape2)
    # same products, fp32 accumulation in a different K order on the halo path -> bf16 rounding may differ in the last bit
    assert _relerr(h_f.float().cpu().numpy(), tape2[0][1].float().cpu().numpy()) < 1e-2
    if shape[1] < 16:
        assert torch.equal(h_f, tape2[0][1])             # same 3x3x3 accumulation + rounding
    assert _relerr(y_f.float().cpu().numpy(), y_r.float().cpu().numpy()) < 1e-2  # h enters the 1x1 GEMM as the same bf16 values
    y_e = st.fwd(x, None)                                # eval: h is not written
    assert torch.equal(y_e, y_f)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_residual_layer_with_dropout_matches_torch(dtype):
    """ResidualLayer with p_dropout > 0 (baseline.py:150-160: Conv3d -> ReLU -> Dropout3d -> Conv3d 1x1x1, + x, ReLU) in TRAINING mode: forward, data gradient and
    all four parameter gradients against torch on the CPU with the SAME channel mask; eval mode ignores the dropout (the fused block)."""
    from synthanatomy_amd.networks.vqvae.baseline import ResidualLayer, _GradCtx, _ResStage
    torch.manual_seed(11)
    C, p = 128, 0.3
    mod = ResidualLayer(C, C, p).cuda().train()
    st = _ResStage(mod, in_act=True, dtype=dtype)
    masks = []

    def fixed_mask(N, Cc, dev, stride=None):
        g = torch.Generator().manual_seed(100 + len(masks))
        m = (torch.bernoulli(torch.full((N, 1, 1, 1, Cc), 1.0 - p), generator=g) / (1.0 - p)).to(dev)
        masks.append(m)
        return m
    st._dropout_mask = fixed_mask
    x = torch.relu(torch.randn(2, 6, 7, 9, C)).to(dtype)
    G = (torch.randn(2, 6, 7, 9, C) * 0.1).to(dtype)
    tape = []
    y = st.fwd(x.cuda(), tape)
    gc = _GradCtx(None)
    Gm = (G.cuda().float() * (y.float() > 0)).to(dtype)      # (the ReLU mask of a stage's OUTPUT is applied by the stage behind it: hand bwd the masked gradient)
    dx = st.bwd(Gm, tape[0], gc)
    torch.cuda.synchronize()
    assert len(masks) == 1 and 0 < int((masks[0] == 0).sum()) < 2 * C
    # torch reference on the same (rounded) operands
    xr = x.float().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    w3, b3 = mod[0].weight.detach().cpu().float().requires_grad_(True), mod[0].bias.detach().cpu().float().requires_grad_(True)
    w1, b1 = mod[3].weight.detach().cpu().float().requires_grad_(True), mod[3].bias.detach().cpu().float().requires_grad_(True)
    if dtype == torch.bfloat16:
        w3e, w1e = w3.to(dtype).float(), w1.to(dtype).float()
    else:
        w3e, w1e = w3, w1
    mref = masks[0].cpu().permute(0, 4, 1, 2, 3)
    h = F.relu(F.conv3d(xr, w3e, b3, padding=1)) * mref
    if dtype == torch.bfloat16:
        h = h + (h.to(dtype).float() - h).detach()          # the hidden activation is stored in bf16
    yr = F.relu(xr + F.conv3d(h, w1e, b1))
    yr.backward(G.float().permute(0, 4, 1, 2, 3))
    tol = 2e-5 if dtype == torch.float32 else 5e-2      # (bf16: the gate of the throughput-mode gradients in tests/test_width_parity_gpu.py is 4e-2)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert rel(y.float().cpu().permute(0, 4, 1, 2, 3), yr.detach()) < tol
    assert rel(dx.float().cpu().permute(0, 4, 1, 2, 3), xr.grad * (x.float().permute(0, 4, 1, 2, 3) > 0)) < tol      # (in_act: the stage applies the ReLU mask of its input)
    errs = [rel(gc.grads[prm].cpu(), ref) for prm, ref in ((mod[0].weight, w3.grad), (mod[0].bias, b3.grad), (mod[3].weight, w1.grad), (mod[3].bias, b1.grad))]
    assert max(errs) < tol, errs
    # eval: no dropout, no mask drawn
    mod.eval()
    y_eval = st.fwd(x.cuda(), None)
    h0 = F.relu(F.conv3d(x.float().permute(0, 4, 1, 2, 3), w3e.detach(), b3.detach(), padding=1))
    if dtype == torch.bfloat16:
        h0 = h0.to(dtype).float()
    y0 = F.relu(x.float().permute(0, 4, 1, 2, 3) + F.conv3d(h0, w1e.detach(), b1.detach()))
    assert len(masks) == 1 and rel(y_eval.float().cpu().permute(0, 4, 1, 2, 3), y0) < tol


def test_network_with_dropout_trains_and_evaluates_deterministically():
    """baseline_vqvae with dropout > 0 through the plugin surface (configure.py:27-37 passes `dropout` as p_dropout): a training step runs on the dropout form of
    every residual block (two steps draw different masks), eval is deterministic and equals the p = 0 network on the same weights."""
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    cfg = dict(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=64, embed_dim=16, n_channels=64,
               n_res_channels=64, n_res_layers=2)
    torch.manual_seed(8)
    net = BaselineVQVAE(**cfg, p_dropout=0.25, compute_dtype=torch.bfloat16).cuda()
    ref = BaselineVQVAE(**cfg, p_dropout=0.0, compute_dtype=torch.bfloat16).cuda()
    ref.load_state_dict(net.state_dict())
    x = torch.rand(2, 1, 16, 24, 16, generator=torch.Generator().manual_seed(1)).cuda()
    net.eval()
    ref.eval()
    with torch.no_grad():
        a, b = net(x)["reconstruction"][0], net(x)["reconstruction"][0]
        c = ref(x)["reconstruction"][0]
    assert torch.equal(a, b) and torch.equal(a, c)
    net.train()
    losses = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        out = net(x)
        loss = F.mse_loss(out["reconstruction"][0].float(), x) + out["quantization_losses"][0]
        loss.backward()
        losses.append(float(loss))
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.requires_grad)
    assert all(np.isfinite(losses)) and losses[0] != losses[1]
