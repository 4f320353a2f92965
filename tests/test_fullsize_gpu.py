"""GPU, BASELINE.json's full sizes: the config-2 network on 160x224x160 volumes, checked through size-independent properties
(the oracle cannot finish these sizes in seconds): batch independence, re-quantisation idempotence, and agreement between the
two independent kernel families (halo mainloops vs the im2col-order kernels) on every tile of the real geometry."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FLIP_NOISE_RATIO = 4.0    # a flipped index must be a tie within 4 half-precision ulps of z (measured: 0.11 at random init -- 4 flips of 2 800 --; unflipped positions: median 44)
GATE_INDEX_AGREEMENT = 0.9975   # measured 0.9986 (4 of 2 800 differ) minus ~0.1 %; round 3 (bf16 forward operands): 0.9918

NET = dict(n_levels=4, downsample_parameters=((4, 2, 1, 1),) * 4, upsample_parameters=((4, 2, 1, 0, 1),) * 4, n_embed=2048, embed_dim=32, n_channels=256,
           n_res_channels=256, n_res_layers=3)
VOL = (160, 224, 160)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def net():
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    torch.manual_seed(4)
    return BaselineVQVAE(**NET, compute_dtype=torch.bfloat16).cuda()


def _grads(net, x):
    for p in net.parameters():
        p.grad = None
    net.train()
    ema = [b.clone() for b in net.quantizer[0].impl.buffers()]
    cb = net.quantizer[0].impl.weight.detach().clone()
    out = net(x)
    loss = torch.nn.functional.mse_loss(out["reconstruction"][0].float(), x) + out["quantization_losses"][0]
    loss.backward()
    net.quantizer[0].impl.wait_ema()
    torch.cuda.synchronize()
    # undo the EMA side effect so that both runs start from the same codebook
    with torch.no_grad():
        for b, old in zip(net.quantizer[0].impl.buffers(), ema):
            b.copy_(old)
        net.quantizer[0].impl.weight.copy_(cb)
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


def test_full_size_eval_properties(net):
    net.eval()
    g = torch.Generator(device="cuda").manual_seed(4)
    x1 = torch.rand(1, 1, *VOL, generator=g, device="cuda")
    x = torch.cat([x1, x1.flip(2)], 0)                      # two different volumes
    with torch.no_grad():
        rec = net(x)["reconstruction"][0]
        idx = net.index_quantize(x)[0]
        rec0 = net(x[:1])["reconstruction"][0]
        idx1 = net.index_quantize(x[1:])[0]
        # a codebook vector is its own nearest code: quantising the embedded indices gives the indices back (bit-exact)
        zq = net.quantizer[0].embed(idx)
        _, _, idx_again = net.quantizer[0].quantize(zq)
        dec = net.decode_samples([idx])
    assert rec.shape == x.shape and idx.shape == (2, 10, 14, 10) and idx.dtype == torch.int64
    assert torch.isfinite(rec).all()
    assert torch.equal(rec[:1], rec0) and torch.equal(idx[1:], idx1)          # samples do not interact, tiles are batch-agnostic
    assert torch.equal(idx_again, idx)
    assert torch.equal(dec, rec)                                               # decode_samples(index_quantize(x)) is the eval forward
    assert 0 <= int(idx.min()) and int(idx.max()) < 2048


def test_full_size_halo_kernels_agree_with_im2col_order_kernels(net):
    """Forward, data gradients and weight gradients of the whole network, halo mainloops vs SA_NO_HALO=1 (bf16 both: the two families sum
    the same products in a different order)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(1, 1, *VOL, generator=g, device="cuda")

    def codes():
        net.eval()
        with torch.no_grad():
            return net.index_quantize(x)[0].clone()

    idx_h = codes()
    loss_h, gr_h = _grads(net, x)
    from synthanatomy_amd import debug
    with debug.override(no_halo=True, no_fused_1x1_bwd=True):   # ... and the two-launch 1x1x1 backward instead of sa_conv1x1_backward
        idx_r = codes()
        loss_r, gr_r = _grads(net, x)
    # bf16 activations: a different summation order moves a few of the 1 400 encoder outputs across a code boundary (4-8 measured), and every
    # flipped code changes the decoder's input at one voxel outright -- the gradient bound below is per flip (measured 0.4-0.8 % each)
    flips = int((idx_h != idx_r).sum())
    assert flips <= idx_h.numel() // 100, flips
    assert np.isfinite(loss_h) and abs(loss_h - loss_r) <= 2e-3 * abs(loss_r)
    assert gr_h.keys() == gr_r.keys() and len(gr_h) >= 100
    bound = 1e-2 + 8e-3 * flips
    worst = max((_rel(gr_h[n], gr_r[n]), n) for n in gr_h)
    assert worst[0] < bound, (worst, flips)
    fro = max((float((gr_h[n].double() - gr_r[n].double()).norm() / (gr_r[n].double().norm() + 1e-30)), n) for n in gr_h)
    assert fro[0] < bound, (fro, flips)
    # the encoder side (everything before the codes) does not see the flips
    enc = max((float((gr_h[n].double() - gr_r[n].double()).norm() / (gr_r[n].double().norm() + 1e-30)), n) for n in gr_h if n.startswith("encoder"))
    assert enc[0] < bound, enc


def test_bf16_mode_against_the_fp32_product_path_at_full_size():
    """The benchmarked mode (bf16 MFMA, encoder forward on float16 operands -- the reference's AMP dtype) against the fp32 product path -- the mode
    pinned to the reference at 1e-3 by tests/test_width_parity_gpu.py -- on the REAL config: same weights, two 160x224x160 volumes.  Random-init
    weights: the encoder output is a nearly flat distribution over 2 048 codes, the hardest case for an argmin under 16-bit rounding.  Round 3
    (bf16 forward operands): 99.18 % of the 2 800 code indices agreed, z 5.6e-3.  The decoder alone, fed the fp32 path's indices, must agree to
    bf16 rounding.  bench.py reports the same numbers as `fp32_mode.bf16_vs_fp32`."""
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    torch.manual_seed(4)
    ref = BaselineVQVAE(**NET, compute_dtype=torch.float32).cuda().eval()
    low = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16)
    assert low.encoder_forward_dtype == torch.float16
    low.load_state_dict(ref.state_dict())
    low = low.cuda().eval()
    old = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16, encoder_forward_dtype=torch.bfloat16)     # the round-3 mode, for the record
    old.load_state_dict(ref.state_dict())
    old = old.cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand(2, 1, *VOL, generator=g, device="cuda")
    with torch.no_grad():
        i32 = ref.index_quantize(x)[0]
        i16 = low.index_quantize(x)[0]
        ibf = old.index_quantize(x)[0]
        z32, z16, zbf = ref.encode(x)[0].float(), low.encode(x)[0].float(), old.encode(x)[0].float()
        r32 = ref.decode_samples([i32]).float()
        r16 = low.decode_samples([i32]).float()
    agree, agree_bf = float((i32 == i16).float().mean()), float((i32 == ibf).float().mean())
    ez, ezbf, er = _rel(z16, z32), _rel(zbf, z32), _rel(r16, r32)
    print(f"[benchmarked mode vs fp32 product path, full size] index agreement {agree:.4f} ({int((i32 != i16).sum())} of {i32.numel()} differ), z max-rel {ez:.2e}, "
          f"reconstruction (same indices) max-rel {er:.2e};  bf16 forward operands: agreement {agree_bf:.4f}, z {ezbf:.2e}")
    assert agree >= GATE_INDEX_AGREEMENT, agree
    assert ez < 2e-3 and er < 2e-2, (ez, er)
    assert ez < 0.35 * ezbf
    # every index that differs is a tie below the 16-bit path's resolution: the fp32 path's distance from its optimum to the code the f16 path chose, against the
    # distance change ONE half-precision ulp of every channel of z would cause between the two codes (the real z carries the rounding of ~30 layers)
    from synthanatomy_amd.utils.general import index_flip_report
    rep = index_flip_report(z32, z16, ref.quantizer[0].impl.weight.detach(), i32, i16)
    print(f"[flips, random init] {rep['flipped']} of {rep['positions']}; gap / one-ulp noise: flipped max {rep['max_flip_ratio']:.2f}, "
          f"unflipped median {rep['median_ratio_of_unflipped']:.0f}, 1st percentile {rep['p01_ratio_of_unflipped']:.1f}")
    assert rep["flipped"] == int((i32 != i16).sum())
    assert rep["max_flip_ratio"] <= FLIP_NOISE_RATIO, rep["flips"]
    del ref, low, old
    torch.cuda.empty_cache()


def test_index_flips_stay_sub_noise_on_a_trained_codebook():
    """The same evidence after TRAINING: 24 steps of the throughput-mode network (EMA codebook updates, Adam) on synthetic volumes move the codebook away from
    its initialisation -- used codes drift to the data, unused ones decay (dead codes: SURVEY section 7 'hard parts') --, then both paths are compared on the
    trained state.  Gate: index agreement as at initialisation, and every flipped index within the same multiple of one half-precision ulp of z."""
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    from synthanatomy_amd.utils.general import index_flip_report
    torch.manual_seed(5)
    low = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16).cuda().train()
    flat = FlatParams(low.parameters())
    opt = FusedAdam(flat, lr=1.65e-4)
    opt.on_step.append(low.invalidate_packed_weights)
    loss_fn = MSELoss()
    g = torch.Generator(device="cuda").manual_seed(5)
    w0 = low.quantizer[0].impl.weight.detach().clone()
    for _ in range(24):
        x = torch.rand(2, 1, *VOL, generator=g, device="cuda")
        flat.zero_grad()
        loss_fn(low(x), x).backward()
        opt.step()
    state = {k: v.detach().cpu().clone() for k, v in low.state_dict().items()}
    moved = float((low.quantizer[0].impl.weight.detach() - w0).norm() / w0.norm())
    assert moved > 1e-2, moved                       # the EMA really changed the codebook
    ref = BaselineVQVAE(**NET, compute_dtype=torch.float32)
    ref.load_state_dict(state)
    ref = ref.cuda().eval()
    low.eval()
    x = torch.rand(2, 1, *VOL, generator=g, device="cuda")
    with torch.no_grad():
        i32, i16 = ref.index_quantize(x)[0], low.index_quantize(x)[0]
        z32, z16 = ref.encode(x)[0].float(), low.encode(x)[0].float()
    rep = index_flip_report(z32, z16, ref.quantizer[0].impl.weight.detach(), i32, i16)
    used = int(torch.unique(i32).numel())
    print(f"[flips, after 24 training steps] codebook moved {moved:.3f} (relative), {used} codes in use, norms max {rep['codebook_norm_max']:.3g} / median "
          f"{rep['codebook_norm_median']:.3g}; {rep['flipped']} of {rep['positions']} indices differ (agreement {rep['agreement']:.4f}); gap / one-ulp noise: "
          f"flipped max {rep['max_flip_ratio']:.2f}, unflipped median {rep['median_ratio_of_unflipped']:.0f}")
    assert rep["agreement"] >= GATE_INDEX_AGREEMENT, rep["agreement"]
    assert rep["max_flip_ratio"] <= FLIP_NOISE_RATIO, rep["flips"]
    del ref, low, flat, opt
    torch.cuda.empty_cache()


def test_adversarial_iteration_at_full_size():
    """The README's training command (--adversarial_component=True, baseline_discriminator, least-square criteria weight 0.005: reference README.md:62-67,
    src/engines/trainer.py:157-256, src/networks/discriminator/baseline.py:21-88) on 160x224x160 volumes, discriminator at ndf=64: two G + D
    iterations run, every loss is finite, both networks move, the adaptive weight is finite and positive, and the BatchNorm statistics of the
    discriminator were updated three times per iteration (fake for G, fake + real for D) as upstream."""
    from synthanatomy_amd.engines.trainer import AdversarialTrainer
    from synthanatomy_amd.losses.adversarial import get_discriminator_loss, get_generator_loss
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.discriminator.baseline import BaselineDiscriminator
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    torch.manual_seed(4)
    net = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16).cuda().train()
    disc = BaselineDiscriminator(input_nc=1, ndf=64, n_layers=3, compute_dtype=torch.bfloat16).cuda().train()
    flat, d_flat = FlatParams(net.parameters()), FlatParams(disc.parameters())
    opt, d_opt = FusedAdam(flat, lr=1.65e-4), FusedAdam(d_flat, lr=5e-5)
    opt.on_step.append(net.invalidate_packed_weights)
    d_opt.on_step.append(lambda: [s_.op.invalidate() for s_ in disc._stages])
    tr = AdversarialTrainer(net, opt, get_generator_loss({"generator_loss": "least_square"}), MSELoss(), disc, d_opt,
                            get_discriminator_loss({"discriminator_loss": "least_square"}), use_adversarial_adaptive_weight=True,
                            adaptive_adversarial_weight_threshold=0, adaptive_adversarial_weight_value=1.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(2, 1, *VOL, generator=g, device="cuda")
    w0, d0 = flat.data.clone(), d_flat.data.clone()
    nb0 = [int(b) for n_, b in disc.named_buffers() if n_.endswith("num_batches_tracked")]
    for it in range(2):
        res = tr.iteration(x, x, 1)
        torch.cuda.synchronize()
        for k in ("loss", "g_loss", "d_loss", "adversarial_weight"):
            assert np.isfinite(float(res[k])), (it, k, float(res[k]))
        assert float(res["adversarial_weight"]) > 0
    assert res["pred"]["reconstruction"][0].shape == x.shape
    assert float((flat.data - w0).abs().max()) > 0 and float((d_flat.data - d0).abs().max()) > 0
    assert torch.isfinite(flat.data).all() and torch.isfinite(d_flat.data).all()
    nb1 = [int(b) for n_, b in disc.named_buffers() if n_.endswith("num_batches_tracked")]
    assert all(b1 - b0 == 6 for b0, b1 in zip(nb0, nb1)), (nb0, nb1)
    del net, disc, tr
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------------------ Performer, full sizes
PERF = dict(vocab=2048, dim=512, depth=24, heads=16, local_heads=8, window=420)
# README latents (10x14x10 = 1 400 tokens, config 2's own grid) and BASELINE.json configs[3]'s "~14k-token" sequence (20x28x25 = 14 000)
GRIDS = {"n1400": ((10, 14, 10), 2), "n14000": ((20, 28, 25), 1)}


@pytest.fixture(scope="module", params=list(GRIDS))
def performer(request):
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    spatial, batch = GRIDS[request.param]
    torch.manual_seed(4)
    order = Ordering("raster_scan", 3, (1,) + spatial, (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
    net = Performer(num_tokens=PERF["vocab"] + 1, max_seq_len=int(np.prod(spatial)) + 1, dim=PERF["dim"], depth=PERF["depth"], heads=PERF["heads"], ordering=order,
                    local_attn_heads=PERF["local_heads"], local_window_size=PERF["window"], feature_redraw_interval=None, use_rezero=True,
                    spatial_position_emb="absolute", spatial_shape=spatial, compute_dtype=torch.bfloat16).cuda().eval()
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_.endswith(".g"):
                p.fill_(0.2)          # the 1e-3 ReZero init would hide the attention path behind the residual
    yield net, spatial, batch
    del net
    torch.cuda.empty_cache()


def test_performer_full_size_properties(performer):
    """24 layers, N = 1 400 and N = 14 000, bf16 throughput mode (split-bf16 attention / scans / projections): determinism, causality up to the
    global key stabiliser, and agreement with the same network run on the exact-fp32 MFMA kernels."""
    net, spatial, batch = performer
    torch.manual_seed(0)
    N = int(np.prod(spatial))
    cut = N - N // 14
    tok = torch.randint(0, PERF["vocab"], (batch, N), device="cuda")
    with torch.no_grad():
        a = net(tok).float()
        b = net(tok).float()
        assert a.shape == (batch, N, PERF["vocab"] + 1)
        assert torch.isfinite(a).all() and torch.equal(a, b)                      # no atomics-order dependence in the forward
        tok2 = tok.clone()
        tok2[:, cut:] = (tok2[:, cut:] + 7) % PERF["vocab"]
        c = net(tok2).float()
        # positions before the edit move only through the keys' global stabiliser (a scalar shift that cancels in the normaliser up to the +eps term)
        early = _rel(c[:, :cut], a[:, :cut])
        late = _rel(c[:, cut:], a[:, cut:])
        assert early < 2e-2 and late > 10 * early, (early, late)
        from synthanatomy_amd import debug
        with debug.override(scan_exact=7, local_attn_exact=True):
            e = net(tok).float()
        assert not torch.equal(e, a)
        d = (e - a).double().norm() / e.double().norm()
        assert float(d) < 2e-3, float(d)                                          # bf16 rounding of the dense layers amplifies 1e-5 differences


def test_performer_training_step_at_14k_tokens():
    """One training step (fwd + CE + bwd) of the 24-layer network on the 14 000-token sequence: finite loss near ln(2049) at initialisation and a
    finite, non-zero gradient for every parameter -- the configuration bench.py reports as `secondary_14k`."""
    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    spatial = (20, 28, 25)
    N = int(np.prod(spatial))
    torch.manual_seed(4)
    order = Ordering("raster_scan", 3, (1,) + spatial, (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
    net = Performer(num_tokens=PERF["vocab"] + 1, max_seq_len=N + 1, dim=PERF["dim"], depth=PERF["depth"], heads=PERF["heads"], ordering=order,
                    local_attn_heads=PERF["local_heads"], local_window_size=PERF["window"], feature_redraw_interval=1, use_rezero=True,
                    spatial_position_emb="absolute", spatial_shape=spatial, compute_dtype=torch.bfloat16).cuda().train()
    tok = torch.randint(0, PERF["vocab"], (1, N), device="cuda")
    seq = torch.nn.functional.pad(tok, (1, 0), value=PERF["vocab"])
    loss = CELoss()(net(seq[:, :-1]).transpose(1, 2), seq[:, 1:])
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - np.log(PERF["vocab"] + 1)) < 0.5, float(loss)
    skip = ("conditioning",)
    for k, p in net.named_parameters():
        if any(s_ in k for s_ in skip):
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert float(net.to_out.weight.grad.abs().max()) > 0 and float(net.performer.net.layers[0][0].fn.to_q.weight.grad.abs().max()) > 0
