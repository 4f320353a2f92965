"""GPU: the reference's --deterministic flag (run_vqvae.py / src/utils/general.py:336-338).  In deterministic mode the reductions that the throughput path
accumulates with fp32 atomics are taken in a fixed order (csrc/deterministic.hip, BatchNorm partial sums, unfused first / last layer routes): two runs of the
same training step give BIT-IDENTICAL gradients, codebook statistics and parameters -- for the VQ-VAE, the adversarial G + D iteration and the
Performer's embedding gradients -- and the values agree with the default path to summation-order noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_fixed_order_reduction_kernels():
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    g = torch.Generator().manual_seed(1)
    # column sums (bias gradients): fp32 and bf16 rows with a padded channel stride
    for dtype, M, C, cs in ((torch.float32, 70001, 33, 36), (torch.bfloat16, 12345, 128, 128), (torch.float32, 5000, 1, 1)):
        x = torch.randn(M, cs, generator=g).to(dtype).cuda()
        nb = lib.sa_colsum_det_workspace_bytes(C)
        outs = []
        for _ in range(2):
            db = torch.full((C,), 0.5, device="cuda")
            ws = torch.empty(nb // 4, device="cuda")
            _ffi.check(lib.sa_colsum_det(_ffi.ptr(x), _ffi.dtype_id(dtype), M, C, cs, _ffi.ptr(db), _ffi.ptr(ws), nb, st))
            outs.append(db.clone())
        assert torch.equal(outs[0], outs[1])
        ref = x[:, :C].double().sum(0).cpu() + 0.5
        assert float((outs[0].double().cpu() - ref).abs().max()) < 1e-3 * float(ref.abs().max() + 1)
    # quantizer statistics from the indices
    M, K, D = 3000, 64, 32
    rows = torch.randn(M, D, generator=g).cuda()
    cb = torch.randn(K, D, generator=g).cuda()
    idx = torch.randint(0, K - 3, (M,), generator=g).cuda()       # the last codes stay unused
    res = []
    for _ in range(2):
        counts, dw, sq, err = torch.full((K,), 9.0, device="cuda"), torch.full((K, D), 9.0, device="cuda"), torch.full((1,), 9.0, device="cuda"), torch.empty(K, device="cuda")
        _ffi.check(lib.sa_vq_stats_det(_ffi.ptr(rows), _ffi.ptr(cb), _ffi.ptr(idx), M, K, D, _ffi.ptr(counts), _ffi.ptr(dw), _ffi.ptr(sq), _ffi.ptr(err), st))
        res.append((counts.clone(), dw.clone(), sq.clone()))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    oh = torch.nn.functional.one_hot(idx.cpu(), K).double()
    assert torch.equal(res[0][0].cpu().double(), oh.sum(0))
    assert _rel(res[0][1], oh.t() @ rows.cpu().double()) < 1e-5
    assert _rel(res[0][2], ((cb[idx] - rows).double() ** 2).sum().reshape(1)) < 1e-5
    # embedding gradient, per-row and per-position index tables
    R, N, dim, nrows = 600, 100, 48, 37
    dy = torch.randn(R, dim, generator=g).cuda()
    for per_pos, ix in ((0, torch.randint(-1, nrows, (R,), generator=g)), (1, torch.randint(-1, nrows, (N,), generator=g))):
        ixd = ix.cuda()
        outs = []
        for _ in range(2):
            tab = torch.zeros(nrows, dim, device="cuda")
            _ffi.check(lib.sa_embed_scatter_det(_ffi.ptr(dy), _ffi.ptr(tab), _ffi.ptr(ixd), per_pos, dim, N, R, nrows, st))
            outs.append(tab.clone())
        assert torch.equal(outs[0], outs[1])
        full = ix if not per_pos else ix.repeat(R // N)
        ref = torch.zeros(nrows, dim, dtype=torch.float64)
        m = full >= 0
        ref.index_add_(0, full[m], dy.cpu().double()[m])
        assert _rel(outs[0], ref) < 1e-5


NET = dict(n_levels=2, downsample_parameters=((4, 2, 1, 1),) * 2, upsample_parameters=((4, 2, 1, 0, 1),) * 2, n_embed=256, embed_dim=32, n_channels=256,
           n_res_channels=256, n_res_layers=1)


def _vq_step(dtype, x, det):
    """One training step of a 2-level network whose first level has the production width (128 channels: one-channel first / last layer kernels, fused residual
    block, fused 1x1x1 backward, halo weight gradients) -> gradients, EMA state."""
    from synthanatomy_amd import debug
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    torch.manual_seed(3)
    net = BaselineVQVAE(**NET, compute_dtype=dtype).cuda().train()
    with debug.override(deterministic=det):
        out = net(x)
        loss = MSELoss()(out, x)      # the product's loss: sa_mse_det in deterministic mode (the logged loss decides the key-metric checkpoint)
        loss.backward()
        net.quantizer[0].impl.wait_ema()
        torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    state = {k: v.detach().clone() for k, v in net.quantizer[0].impl.state_dict().items()}
    return float(loss.detach()), grads, state


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_vqvae_training_step_is_bit_reproducible_in_deterministic_mode(dtype):
    x = torch.rand(2, 1, 32, 48, 32, generator=torch.Generator().manual_seed(7)).cuda()
    l0, g0, s0 = _vq_step(dtype, x, True)
    l1, g1, s1 = _vq_step(dtype, x, True)
    assert len(g0) >= 20
    assert l0 == l1, (l0, l1)      # the loss VALUE bit for bit (fixed-order squared-error sum)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    # ... and it computes the same thing as the default path (fp32 atomics, fused first / last layer), up to summation order
    l2, g2, s2 = _vq_step(dtype, x, False)
    assert abs(l0 - l2) <= 1e-5 * abs(l2)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4      # (bf16: the unfused first layer rounds its im2col matrix once more)
    for k in g0:
        assert _rel(g0[k], g2[k]) < tol, (k, _rel(g0[k], g2[k]))
    for k in s0:
        assert _rel(s0[k], s2[k]) < 1e-4, k


def test_adversarial_iteration_is_bit_reproducible_in_deterministic_mode():
    from synthanatomy_amd import debug
    from synthanatomy_amd.engines.trainer import AdversarialTrainer
    from synthanatomy_amd.losses.adversarial import get_discriminator_loss, get_generator_loss
    from synthanatomy_amd.losses.vqvae import MSELoss
    from synthanatomy_amd.networks.discriminator.baseline import BaselineDiscriminator
    from synthanatomy_amd.networks.vqvae.baseline import BaselineVQVAE
    from synthanatomy_amd.runtime.optim import FlatParams, FusedAdam
    x = torch.rand(2, 1, 32, 32, 32, generator=torch.Generator().manual_seed(9)).cuda()

    def run():
        torch.manual_seed(5)
        net = BaselineVQVAE(**NET, compute_dtype=torch.bfloat16).cuda().train()
        disc = BaselineDiscriminator(input_nc=1, ndf=16, n_layers=3, compute_dtype=torch.bfloat16).cuda().train()
        flat, d_flat = FlatParams(net.parameters()), FlatParams(disc.parameters())
        opt, d_opt = FusedAdam(flat, lr=1e-3), FusedAdam(d_flat, lr=5e-4)
        opt.on_step.append(net.invalidate_packed_weights)
        d_opt.on_step.append(lambda: [s_.op.invalidate() for s_ in disc._stages])
        tr = AdversarialTrainer(net, opt, get_generator_loss({"generator_loss": "least_square"}), MSELoss(), disc, d_opt,
                                get_discriminator_loss({"discriminator_loss": "least_square"}), use_adversarial_adaptive_weight=True)
        with debug.override(deterministic=True):
            for _ in range(3):
                res = tr.iteration(x, x, 1)
            net.quantizer[0].impl.wait_ema()
            torch.cuda.synchronize()
        return flat.data.clone(), d_flat.data.clone(), {k: v.clone() for k, v in disc.state_dict().items() if "running" in k}, float(res["adversarial_weight"])

    a, b = run(), run()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[3] == b[3]
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()


def test_performer_embedding_gradients_are_bit_reproducible_in_deterministic_mode():
    from synthanatomy_amd import debug
    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    shape, n = (2, 4, 5), 40
    g = torch.Generator().manual_seed(2)
    tok = torch.randint(0, 33, (3, n), generator=g).cuda()
    tgt = torch.randint(0, 32, (3, n), generator=g).cuda()

    def run(det):
        torch.manual_seed(1)
        o = Ordering("raster_scan", 3, (1,) + shape, (False,) * 3, (), ())
        net = Performer(num_tokens=33, max_seq_len=n, dim=64, depth=2, heads=4, ordering=o, dim_head=64, local_attn_heads=2, local_window_size=8, use_rezero=True,
                        spatial_position_emb="absolute", spatial_shape=shape, feature_redraw_interval=None, compute_dtype=torch.bfloat16).cuda().train()
        with debug.override(deterministic=det):
            CELoss()(net(tok).transpose(1, 2), tgt).backward()
            torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None and "emb" in k}

    a, b, c = run(True), run(True), run(False)
    assert len(a) >= 5
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert _rel(a[k], c[k]) < 1e-4, k


@pytest.mark.parametrize("rezero", [True, False])
def test_performer_step_is_bit_reproducible_in_deterministic_mode(rezero):
    """Loss and EVERY parameter gradient of a Performer step (production width, N = 1 400, W = 420; ReZero and pre-LayerNorm forms) are bit-identical between two
    runs in deterministic mode: ReZero gate gradients through sa_dot_det, LayerNorm weight / bias gradients through sa_layernorm_dwprod + sa_colsum_det, the loss
    through sa_cross_entropy_rows + sa_sum_det, embedding gradients through sa_embed_scatter_det; everything else (dense layers, FAVOR+, local attention) has no
    atomics.  Against the default mode the results agree to rounding."""
    from synthanatomy_amd import debug
    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    shape, n = (10, 14, 10), 1400
    g = torch.Generator().manual_seed(4)
    tok = torch.randint(0, 2049, (2, n), generator=g).cuda()
    tgt = torch.randint(0, 2048, (2, n), generator=g).cuda()

    def run(det):
        torch.manual_seed(3)
        o = Ordering("raster_scan", 3, (1,) + shape, (False,) * 3, (), ())
        net = Performer(num_tokens=2049, max_seq_len=n, dim=512, depth=2, heads=16, ordering=o, dim_head=64, local_attn_heads=8, local_window_size=420,
                        use_rezero=rezero, spatial_position_emb="absolute", spatial_shape=shape, feature_redraw_interval=None, compute_dtype=torch.bfloat16).cuda().train()
        if rezero:
            with torch.no_grad():
                for k, p in net.named_parameters():
                    if k.endswith(".g"):
                        p.fill_(0.2)
        with debug.override(deterministic=det):
            loss = CELoss()(net(tok).transpose(1, 2), tgt)
            loss.backward()
            torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    (la, a), (lb, b), (lc, c) = run(True), run(True), run(False)
    assert la == lb
    assert len(a) >= 20
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert abs(la - lc) < 1e-5 * abs(lc)
    worst = max(_rel(a[k], c[k]) for k in a if float(c[k].abs().max()) > 0)
    assert worst < 1e-3, worst
