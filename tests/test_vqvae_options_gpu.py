"""``use_subpixel_conv=True`` (reference src/networks/vqvae/baseline.py:274-282: the last decoder layer is MONAI's SubpixelUpsample): the conv_block on the
implicit-GEMM kernels + the pixelshuffle / pad / pool gather (sa_subpixel_pool_fwd / _bwd) against oracle/vqvae_ref.py -- reconstruction, loss and every gradient."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vqvae_ref as V  # noqa: E402


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _cfg():
    return V.VQVAEConfig(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=64, embed_dim=16,
                         n_channels=32, n_res_channels=32, n_res_layers=1, use_subpixel_conv=True)


def _net(cfg, st, dtype):
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    net = BaselineVQVAE(n_levels=2, downsample_parameters=cfg.downsample_parameters, upsample_parameters=cfg.upsample_parameters, n_embed=64, embed_dim=16,
                        n_channels=32, n_res_channels=32, n_res_layers=1, use_subpixel_conv=True, compute_dtype=dtype)
    assert set(net.state_dict().keys()) == set(st.keys()), set(net.state_dict().keys()) ^ set(st.keys())
    net.load_state_dict({k: v.clone() for k, v in st.items()})
    return net.cuda().train()


def test_pool_kernels_against_torch():
    """the gather and its adjoint alone: pixelshuffle -> pad -> average pool of torch on the same tensor, odd sizes"""
    import torch.nn.functional as F
    from synthanatomy_amd import _ffi
    N, D, H, W = 2, 3, 5, 4
    torch.manual_seed(0)
    c = torch.randn(N, D, H, W, 8, device="cuda")
    out = torch.empty(N, 2 * D, 2 * H, 2 * W, device="cuda")
    _ffi.check(_ffi.lib().sa_subpixel_pool_fwd(_ffi.ptr(c), _ffi.ptr(out), N, D, H, W, _ffi.stream()), "fwd")
    cr = c.permute(0, 4, 1, 2, 3).cpu().requires_grad_(True)
    ref = F.avg_pool3d(F.pad(V.pixelshuffle3d(cr, 2), (1, 0) * 3), 2, stride=1)
    assert torch.allclose(out.cpu(), ref[:, 0].detach(), atol=1e-6)
    g = torch.randn(N, 2 * D, 2 * H, 2 * W, device="cuda")
    ref.backward(g.cpu()[:, None])
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        dc = torch.empty(N, D, H, W, 8, dtype=dt, device="cuda")
        _ffi.check(_ffi.lib().sa_subpixel_pool_bwd(_ffi.ptr(g), _ffi.ptr(dc), _ffi.dtype_id(dt), N, D, H, W, _ffi.stream()), "bwd")
        assert torch.allclose(dc.float().cpu(), cr.grad.permute(0, 2, 3, 4, 1), atol=tol, rtol=tol)


def test_subpixel_decoder_matches_oracle_fp32():
    cfg = _cfg()
    st = V.init_state(cfg, seed=3)
    net = _net(cfg, st, torch.float32)
    torch.manual_seed(1)
    x = torch.rand(2, 1, 16, 24, 16)
    out = net(x.cuda())
    loss = torch.nn.functional.mse_loss(out["reconstruction"][0], x.cuda()) + out["quantization_losses"][0]
    loss.backward()
    leaf = {k: v.clone().requires_grad_(v.is_floating_point() and "quantizer" not in k) for k, v in st.items()}
    ref = V.forward(leaf, cfg, x, training=True)
    rl = V.mse_loss(ref, x)
    rl.backward()
    assert out["reconstruction"][0].shape == ref["reconstruction"][0].shape == (2, 1, 16, 24, 16)
    assert _rel(out["reconstruction"][0], ref["reconstruction"][0]) < 1e-3
    assert abs(loss.item() - rl.item()) <= 1e-3 * abs(rl.item())
    params = dict(net.named_parameters())
    n = 0
    for k, v in leaf.items():
        if v.requires_grad and v.grad is not None and k in params:
            assert _rel(params[k].grad, v.grad) < 2e-3, (k, _rel(params[k].grad, v.grad))
            n += 1
    assert n >= 14 and "decoder.0.5.conv_block.weight" in params
    # the adaptive adversarial weight's last-layer gradient goes through the same stage (wgrad only)
    assert net.get_last_layer() is params["decoder.0.5.conv_block.weight"]


def test_subpixel_decoder_bf16_against_the_rounded_oracle_and_inference():
    cfg = _cfg()
    st = V.init_state(cfg, seed=4)
    net = _net(cfg, st, torch.bfloat16)
    torch.manual_seed(2)
    x = torch.rand(2, 1, 16, 16, 16)
    out = net(x.cuda())
    (torch.nn.functional.mse_loss(out["reconstruction"][0], x.cuda()) + out["quantization_losses"][0]).backward()
    ref = V.forward({k: v.clone() for k, v in st.items()}, cfg, x, training=True, round_dtype=torch.bfloat16, enc_round_dtype=torch.float16)
    assert _rel(out["reconstruction"][0], ref["reconstruction"][0]) < 3e-2
    assert float(net.decoder[0][5].conv_block.weight.grad.abs().max()) > 0
    net.eval()
    with torch.no_grad():
        rec = net.decode_samples(net.index_quantize(x.cuda()))
    assert rec.shape == (2, 1, 16, 16, 16) and bool(torch.isfinite(rec).all())
