"""CPU: the non-README options of the kept surface (round 6) -- constructor surface and state_dict keys of the product modules, and the oracle's restatements of the
third-party pieces behind them (MONAI pixelshuffle / pad-pool, performer_pytorch rotate-every-two, axial_positional_embedding) against explicit element-wise forms."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import performer_ref as P  # noqa: E402
from oracle import vqvae_ref as V  # noqa: E402


def _ordering(shape=(2, 3, 4)):
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    return Ordering("raster_scan", 3, (1,) + shape, (False,) * 3, (), ())


def _performer(**kw):
    from synthanatomy_amd.networks.transformers.performer import Performer
    return Performer(num_tokens=33, max_seq_len=24, dim=32, depth=1, heads=4, ordering=_ordering(), dim_head=64, local_attn_heads=2, local_window_size=6,
                     use_rezero=True, spatial_position_emb="absolute", spatial_shape=(2, 3, 4), feature_redraw_interval=None, **kw)


def test_performer_option_modules_have_the_reference_keys():
    sd = _performer(rotary_position_emb=True).state_dict()
    assert sd["pos_emb.emb"].shape == (24, 32) and sd["layer_pos_emb.emb"].shape == (24, 64)          # performer.py:134-137
    assert torch.allclose(sd["layer_pos_emb.emb"], P.fixed_position_table(64, 24), atol=1e-6)
    sd = _performer(axial_position_emb=True).state_dict()                                              # default grid (ceil(24 / 64), 64), performer.py:142-144
    assert sd["pos_emb.weights_0"].shape == (1, 1, 1, 32) and sd["pos_emb.weights_1"].shape == (1, 1, 64, 32)
    sd = _performer(axial_position_emb=True, axial_position_shape=(4, 6)).state_dict()
    assert sd["pos_emb.weights_0"].shape == (1, 4, 1, 32) and sd["pos_emb.weights_1"].shape == (1, 1, 6, 32)
    net = _performer(tie_embed=True)
    assert net.to_out is None and not any(k.startswith("to_out") for k in net.state_dict())          # performer.py:222
    assert _performer(emb_dropout=0.3).dropout.p == 0.3
    with pytest.raises(AssertionError):                                                                # performer.py:127-132: exclusive
        _performer(rotary_position_emb=True, axial_position_emb=True)


def test_subpixel_module_keys_and_icnr_initialisation():
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    kw = dict(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=64, embed_dim=16, n_channels=32,
              n_res_channels=32, n_res_layers=1)
    net = BaselineVQVAE(use_subpixel_conv=True, **kw)
    cfg = V.VQVAEConfig(use_subpixel_conv=True, **{k: v for k, v in kw.items()})
    assert set(net.state_dict()) == set(V.init_state(cfg, 0))
    w = net.decoder[0][5].conv_block.weight                       # [8, 16, 3, 3, 3]: one Kaiming kernel repeated over the eight sub-pixel channels (ICNR)
    assert w.shape == (8, 16, 3, 3, 3) and all(torch.equal(w[0], w[i]) for i in range(1, 8)) and float(w.detach().std()) > 0
    assert net.get_last_layer() is w
    with pytest.raises(NotImplementedError):                      # one level: the reference's own SubpixelUpsample(in_channels=n_channels // 2) cannot follow its stack
        BaselineVQVAE(use_subpixel_conv=True, **{**kw, "n_levels": 1, "downsample_parameters": ((4, 2, 1, 1),), "upsample_parameters": ((4, 2, 1, 0, 1),)})


def test_oracle_pixelshuffle_and_pad_pool_against_explicit_loops():
    torch.manual_seed(0)
    x = torch.randn(2, 16, 2, 3, 2)
    y = V.pixelshuffle3d(x, 2)
    assert y.shape == (2, 2, 4, 6, 4)
    for o in range(2):
        for fd in range(2):
            for fh in range(2):
                for fw in range(2):
                    assert torch.equal(y[:, o, fd::2, fh::2, fw::2], x[:, o * 8 + (fd * 2 + fh) * 2 + fw])
    w, b = torch.randn(8, 4, 3, 3, 3), torch.randn(8)
    xin = torch.randn(1, 4, 2, 3, 2)
    out = V.subpixel_upsample(xin, w, b, 2)
    s = V.pixelshuffle3d(F.conv3d(xin, w, b, padding=1), 2)[0, 0]
    ref = torch.zeros_like(s)
    for z in range(s.shape[0]):
        for yy in range(s.shape[1]):
            for xx in range(s.shape[2]):
                acc = 0.0
                for dz in (0, 1):
                    for dy in (0, 1):
                        for dx in (0, 1):
                            a, bb, c = z - 1 + dz, yy - 1 + dy, xx - 1 + dx
                            if a >= 0 and bb >= 0 and c >= 0:
                                acc += float(s[a, bb, c])
                ref[z, yy, xx] = acc / 8
    assert out.shape == (1, 1, 4, 6, 4) and torch.allclose(out[0, 0], ref, atol=1e-5)


def test_oracle_rotary_pairs_and_axial_table():
    torch.manual_seed(1)
    q, k = torch.randn(2, 3, 7, 64), torch.randn(2, 3, 7, 64)
    tab = P.fixed_position_table(64, 9)[:7]
    qr, kr = P.apply_rotary_pos_emb(q, k, tab)
    for n in range(7):              # element-wise: pair (2i, 2i + 1) rotated by the angle n * inv_freq_i
        for i in (0, 5, 31):
            ang = n / (10000 ** (2 * i / 64))
            c, s_ = math.cos(ang), math.sin(ang)
            assert torch.allclose(qr[..., n, 2 * i], q[..., n, 2 * i] * c - q[..., n, 2 * i + 1] * s_, atol=1e-4)
            assert torch.allclose(qr[..., n, 2 * i + 1], q[..., n, 2 * i + 1] * c + q[..., n, 2 * i] * s_, atol=1e-4)
    assert torch.allclose(qr.norm(dim=-1), q.norm(dim=-1), atol=1e-4)
    d = (qr[0, 0] @ kr[0, 0].t())
    q1 = q[:, :, :1].expand(-1, -1, 7, -1)
    k1 = k[:, :, :1].expand(-1, -1, 7, -1)
    qq, kk = P.apply_rotary_pos_emb(q1, k1, tab)
    dd = qq[0, 0] @ kk[0, 0].t()    # the same vectors at every position: scores depend on the position DIFFERENCE only
    assert torch.allclose(dd[2, 1], dd[6, 5], atol=1e-3) and torch.allclose(dd[3, 0], dd[6, 3], atol=1e-3) and d.shape == (7, 7)
    st = {"pos_emb.weights_0": torch.randn(1, 3, 1, 8), "pos_emb.weights_1": torch.randn(1, 1, 4, 8)}
    t = P.axial_position_table(st, (3, 4))
    assert t.shape == (12, 8)
    for pos in range(12):
        assert torch.allclose(t[pos], st["pos_emb.weights_0"][0, pos // 4, 0] + st["pos_emb.weights_1"][0, 0, pos % 4])
