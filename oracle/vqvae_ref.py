"""ORACLE (test infrastructure only) -- CPU restatement of the reference VQ-VAE hot path.

This file is a checker. Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``synthanatomy_amd`` never does.

It restates, as plain functions over a ``state`` dict (reference state_dict key
names), the arithmetic of ``/root/reference/src/networks/vqvae/baseline.py``:

* encoder   ``construct_encoder`` baseline.py:213-246, ``encode`` :329-330
* residual  ``ResidualLayer``     baseline.py:150-160
* quantizer ``Quantizer_impl.forward`` baseline.py:38-87, ``Quantizer.forward`` :105-122
* decoder   ``construct_decoder`` baseline.py:257-299, ``decode`` :338-340
* discriminator ``BaselineDiscriminator`` discriminator/baseline.py:21-88
* MSE loss  ``MSELoss.forward`` losses/vqvae/vqvae.py:14-71

Parity status: PINNED -- ``tests/test_oracle_vs_golden.py`` checks every function
here against fixtures produced by importing the reference itself
(``tests/golden/make_goldens.py``).

A ``round_dtype`` knob emulates the product's reduced-precision storage (activations
and weights rounded to bf16 -- or float16, the encoder's forward operand type in the
product's throughput mode -- between layers, fp32 accumulation) so the 16-bit HIP paths
can be compared at tight tolerance; with ``round_dtype=None`` it is the fp32 reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VQVAEConfig:
    n_levels: int = 3
    downsample_parameters: Tuple[Tuple[int, int, int, int], ...] = ((4, 2, 1, 1),) * 3
    upsample_parameters: Tuple[Tuple[int, int, int, int, int], ...] = ((4, 2, 1, 0, 1),) * 3
    n_embed: int = 256
    embed_dim: int = 256
    n_channels: int = 144
    n_res_channels: int = 144
    n_res_layers: int = 3
    p_dropout: float = 0.0
    commitment_cost: float = 0.25
    vq_decay: float = 0.5
    eps: float = 1e-5
    use_subpixel_conv: bool = False     # baseline.py:274-282: the LAST up-sampling layer is MONAI's SubpixelUpsample(3, n_channels // 2, 1, scale_factor, apply_pad_pool=True)

    def enc_channels(self, level: int) -> Tuple[int, int]:
        cin = 1 if level == 0 else self.n_channels // 2
        cout = self.n_channels // (1 if level == self.n_levels - 1 else 2)
        return cin, cout

    def dec_channels(self, level: int) -> Tuple[int, int]:
        cin = self.n_channels // (1 if level == 0 else 2)
        cout = 1 if level == self.n_levels - 1 else self.n_channels // 2
        return cin, cout


def _uniform(shape, bound, gen):
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def init_state(cfg: VQVAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random state with the reference's key names / shapes (values are NOT the
    reference's RNG stream; parity tests load reference weights from fixtures)."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}

    def conv(prefix, cout, cin, k):
        fan_in = cin * k ** 3
        b = 1.0 / math.sqrt(fan_in)
        st[prefix + ".weight"] = _uniform((cout, cin, k, k, k), b, g)
        st[prefix + ".bias"] = _uniform((cout,), b, g)

    def convT(prefix, cin, cout, k):
        fan_in = cout * k ** 3
        b = 1.0 / math.sqrt(fan_in)
        st[prefix + ".weight"] = _uniform((cin, cout, k, k, k), b, g)
        st[prefix + ".bias"] = _uniform((cout,), b, g)

    for i in range(cfg.n_levels):
        cin, cout = cfg.enc_channels(i)
        conv(f"encoder.0.{3 * i}", cout, cin, cfg.downsample_parameters[i][0])
        for r in range(cfg.n_res_layers):
            conv(f"encoder.0.{3 * i + 2}.{r}.0", cout, cout, 3)
            conv(f"encoder.0.{3 * i + 2}.{r}.3", cout, cout, 1)
    conv(f"encoder.0.{3 * cfg.n_levels}", cfg.embed_dim, cfg.n_channels, 3)

    w = torch.randn(cfg.n_embed, cfg.embed_dim, generator=g)
    st["quantizer.0.impl.weight"] = w
    st["quantizer.0.impl.embedding.weight"] = w  # same storage in the reference (baseline.py:33)
    st["quantizer.0.impl.N"] = torch.zeros(cfg.n_embed)
    st["quantizer.0.impl.embed_avg"] = w.clone()

    conv("decoder.0.0", cfg.n_channels, cfg.embed_dim, 3)
    for i in range(cfg.n_levels):
        cin, cout = cfg.dec_channels(i)
        for r in range(cfg.n_res_layers):
            conv(f"decoder.0.{1 + 3 * i}.{r}.0", cin, cin, 3)
            conv(f"decoder.0.{1 + 3 * i}.{r}.3", cin, cin, 1)
        if cfg.use_subpixel_conv and i == cfg.n_levels - 1:
            # MONAI SubpixelUpsample: `conv_block` = Conv3d(n_channels // 2 -> out_channels * scale^3, k 3, p 1); (its ICNR initialisation is not restated: parity
            # tests load explicit weights)
            conv(f"decoder.0.{2 + 3 * i}.conv_block", cout * cfg.upsample_parameters[i][1] ** 3, cfg.n_channels // 2, 3)
        else:
            convT(f"decoder.0.{2 + 3 * i}", cin, cout, cfg.upsample_parameters[i][0])
    return st


def _rd(t: torch.Tensor, round_dtype: Optional[torch.dtype]) -> torch.Tensor:
    if round_dtype is None:
        return t
    if round_dtype == torch.float16:
        # float16 is a FORWARD storage type only (the product keeps every gradient in bf16 / fp32): round the value, pass the gradient straight through --
        # autograd through .to(float16) would push the (tiny, un-scaled) gradients through half precision and flush them
        return t + (t.detach().to(round_dtype).to(torch.float32) - t.detach())
    return t.to(round_dtype).to(torch.float32)


def residual_layer(x, w3, b3, w1, b1, round_dtype=None):
    """relu(x + conv1x1(relu(conv3x3(x))))  -- baseline.py:150-160 (dropout p=0)."""
    h = _rd(F.relu(F.conv3d(x, _rd(w3, round_dtype), b3, padding=1)), round_dtype)
    y = F.relu(x + F.conv3d(h, _rd(w1, round_dtype), b1))
    return _rd(y, round_dtype)


def encode(st, cfg: VQVAEConfig, images: torch.Tensor, round_dtype=None) -> torch.Tensor:
    """baseline.py:213-246 / :329-330.  images [B,1,D,H,W] fp32 -> z [B,embed_dim,d,h,w]."""
    x = _rd(images if images.dtype == torch.float64 else images.float(), round_dtype)   # fp64 in = fp64 throughout (error-floor checks)
    for i in range(cfg.n_levels):
        k, s, p, dil = cfg.downsample_parameters[i]
        p_ = f"encoder.0.{3 * i}"
        x = F.conv3d(x, _rd(st[p_ + ".weight"], round_dtype), st[p_ + ".bias"], stride=s, padding=p, dilation=dil)
        x = _rd(F.relu(x), round_dtype)
        for r in range(cfg.n_res_layers):
            q = f"encoder.0.{3 * i + 2}.{r}"
            x = residual_layer(x, st[q + ".0.weight"], st[q + ".0.bias"], st[q + ".3.weight"], st[q + ".3.bias"], round_dtype)
    p_ = f"encoder.0.{3 * cfg.n_levels}"
    # the pre-VQ conv output feeds the fp32 quantizer: it is never rounded
    z = F.conv3d(x, _rd(st[p_ + ".weight"], round_dtype), st[p_ + ".bias"], padding=1)
    return z


def vq_distances(flat: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """baseline.py:49-53 -- the expanded form, fp32."""
    return (flat ** 2).sum(dim=1, keepdim=True) - 2 * torch.mm(flat, weight.t()) + (weight ** 2).sum(dim=1, keepdim=True).t()


def quantize(st, cfg: VQVAEConfig, z: torch.Tensor, training: bool, world_stats=None):
    """baseline.py:38-87.  Returns (zq_st, loss, idx, aux) and, when ``training``,
    updates ``st`` (N, embed_avg, weight) in place exactly like the reference.

    ``world_stats`` optionally supplies (counts, dw) already summed over ranks
    (the two all_reduce calls at baseline.py:70-72)."""
    b, c, h, w, d = z.shape
    x = z.float()
    flat = x.permute(0, 2, 3, 4, 1).contiguous().view(-1, cfg.embed_dim)
    W = st["quantizer.0.impl.weight"]
    dist = vq_distances(flat, W)
    idx = torch.max(-dist, dim=1)[1]
    onehot = F.one_hot(idx, cfg.n_embed).type_as(flat)
    idx = idx.view(b, h, w, d)
    quantized = F.embedding(idx, W).permute(0, 4, 1, 2, 3).contiguous()  # pre-update codebook (baseline.py:63)
    counts = onehot.sum(0)
    dw = torch.mm(onehot.t(), flat)
    if training:
        with torch.no_grad():
            cs, ds = (counts, dw) if world_stats is None else world_stats
            g = cfg.vq_decay
            st["quantizer.0.impl.N"].mul_(g).add_(torch.mul(cs, 1 - g))
            st["quantizer.0.impl.embed_avg"].mul_(g).add_(torch.mul(ds, 1 - g))
            n = st["quantizer.0.impl.N"].sum()
            Wn = (st["quantizer.0.impl.N"] + cfg.eps) / (n + cfg.n_embed * cfg.eps) * n
            st["quantizer.0.impl.weight"].copy_(st["quantizer.0.impl.embed_avg"] / Wn.unsqueeze(1))
    loss = cfg.commitment_cost * F.mse_loss(quantized.detach(), x)
    zq_st = (quantized - x).detach() + x
    return zq_st, loss, idx, {"counts": counts, "dw": dw, "dist": dist}


def perplexity(idx: torch.Tensor, n_embed: int) -> torch.Tensor:
    """baseline.py:110-120."""
    p = torch.histc(idx.float(), bins=n_embed, max=n_embed).float().div(idx.numel())
    return torch.exp(-torch.sum(p * torch.log(p + 1e-10)))


def embed(st, idx: torch.Tensor) -> torch.Tensor:
    """baseline.py:89-91."""
    return F.embedding(idx, st["quantizer.0.impl.weight"]).permute(0, 4, 1, 2, 3).contiguous()


def pixelshuffle3d(x: torch.Tensor, factor: int) -> torch.Tensor:
    """monai.networks.utils.pixelshuffle(x, 3, factor): channel c = o * factor^3 + (fd * factor + fh) * factor + fw of voxel (d, h, w) becomes channel o of
    voxel (d * factor + fd, h * factor + fh, w * factor + fw).  (MONAI is absent offline: restated from the published source.)"""
    b, c, d, h, w = x.shape
    o = c // factor ** 3
    x = x.reshape(b, o, factor, factor, factor, d, h, w)
    x = x.permute(0, 1, 5, 2, 6, 3, 7, 4)
    return x.reshape(b, o, d * factor, h * factor, w * factor)


def subpixel_upsample(x, weight, bias, factor: int):
    """monai.networks.blocks.SubpixelUpsample(dimensions=3, ..., apply_pad_pool=True): conv_block -> pixelshuffle -> ConstantPad3d((factor - 1, 0) * 3, 0) ->
    AvgPool3d(kernel_size=factor, stride=1)."""
    x = pixelshuffle3d(F.conv3d(x, weight, bias, padding=1), factor)
    x = F.pad(x, (factor - 1, 0) * 3, value=0.0)
    return F.avg_pool3d(x, kernel_size=factor, stride=1)


def decode(st, cfg: VQVAEConfig, zq: torch.Tensor, round_dtype=None) -> torch.Tensor:
    """baseline.py:257-299 / :338-340."""
    x = _rd(zq, round_dtype)
    x = F.conv3d(x, _rd(st["decoder.0.0.weight"], round_dtype), st["decoder.0.0.bias"], padding=1)
    x = _rd(x, round_dtype)  # no ReLU after the post-VQ conv
    for i in range(cfg.n_levels):
        for r in range(cfg.n_res_layers):
            q = f"decoder.0.{1 + 3 * i}.{r}"
            x = residual_layer(x, st[q + ".0.weight"], st[q + ".0.bias"], st[q + ".3.weight"], st[q + ".3.bias"], round_dtype)
        k, s, p, op, dil = cfg.upsample_parameters[i]
        q = f"decoder.0.{2 + 3 * i}"
        if cfg.use_subpixel_conv and i == cfg.n_levels - 1:
            x = subpixel_upsample(x, _rd(st[q + ".conv_block.weight"], round_dtype), st[q + ".conv_block.bias"], s)
        else:
            x = F.conv_transpose3d(x, _rd(st[q + ".weight"], round_dtype), st[q + ".bias"], stride=s, padding=p, output_padding=op, dilation=dil)
        if i != cfg.n_levels - 1:
            x = _rd(F.relu(x), round_dtype)
    return x


def forward(st, cfg: VQVAEConfig, images, training: bool, round_dtype=None, world_stats=None, enc_round_dtype="same"):
    """baseline.py:354-362.  ``enc_round_dtype`` (default: ``round_dtype``): storage rounding of the ENCODER half alone -- the product's throughput
    mode runs the encoder's forward on float16 operands (the reference's AMP dtype) and everything else on bf16."""
    z = encode(st, cfg, images, round_dtype if isinstance(enc_round_dtype, str) else enc_round_dtype)
    zq, qloss, idx, aux = quantize(st, cfg, z, training, world_stats)
    recon = decode(st, cfg, zq, round_dtype)
    return {"reconstruction": [recon], "quantization_losses": [qloss], "indices": idx, "z": z, "aux": aux}


def mse_loss(out, target):
    """losses/vqvae/vqvae.py:14-71 (MSELoss): mse(recon, y) + sum(q_losses)."""
    return F.mse_loss(out["reconstruction"][0].float(), target.float()) + sum(out["quantization_losses"])


# ----------------------------------------------------------------------------- discriminator
def init_discriminator_state(seed=0, input_nc=1, ndf=64, n_layers=3):
    g = torch.Generator().manual_seed(seed)
    st = {}
    chans = [(input_nc, ndf)]
    m = 1
    for n in range(1, n_layers):
        mp, m = m, min(2 ** n, 8)
        chans.append((ndf * mp, ndf * m))
    mp, m = m, min(2 ** n_layers, 8)
    chans.append((ndf * mp, ndf * m))
    chans.append((ndf * m, 1))
    idx = 0
    for li, (ci, co) in enumerate(chans):
        st[f"main.{idx}.weight"] = torch.randn(co, ci, 4, 4, 4, generator=g) * 0.02
        has_bias = li == 0 or li == len(chans) - 1
        if has_bias:
            st[f"main.{idx}.bias"] = _uniform((co,), 1.0 / math.sqrt(ci * 64), g)
        if 0 < li < len(chans) - 1:
            st[f"main.{idx + 1}.weight"] = 1.0 + torch.randn(co, generator=g) * 0.02
            st[f"main.{idx + 1}.bias"] = torch.zeros(co)
            st[f"main.{idx + 1}.running_mean"] = torch.zeros(co)
            st[f"main.{idx + 1}.running_var"] = torch.ones(co)
            idx += 3
        else:
            idx += 2
    return st


def discriminator_forward(st, x, training=True, n_layers=3):
    """discriminator/baseline.py:21-88: conv k4 (s2,s2,s2,s1,s1) + BN + LeakyReLU(0.2)."""
    idx = 0
    n_convs = n_layers + 2
    for li in range(n_convs):
        stride = 2 if li < n_layers else 1
        x = F.conv3d(x, st[f"main.{idx}.weight"], st.get(f"main.{idx}.bias"), stride=stride, padding=1)
        if li == 0:
            x = F.leaky_relu(x, 0.2)
            idx += 2
        elif li < n_convs - 1:
            x = F.batch_norm(x, st[f"main.{idx + 1}.running_mean"], st[f"main.{idx + 1}.running_var"],
                             st[f"main.{idx + 1}.weight"], st[f"main.{idx + 1}.bias"], training, 0.1, 1e-5)
            x = F.leaky_relu(x, 0.2)
            idx += 3
    return x
