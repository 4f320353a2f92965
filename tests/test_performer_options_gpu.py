"""The non-README options of the Performer wrapper (round 6; reference src/networks/transformers/performer.py:94-106,134-148,201,270,286-288):
``rotary_position_emb`` (pairwise rotation of q / k of the FAVOR+ heads in every layer, csrc sa_rotary_pairs), ``axial_position_emb`` (two learned axis
tables), ``tie_embed`` (logits through the token table) and ``emb_dropout`` -- logits and every parameter gradient against oracle/performer_ref.py."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import performer_ref as P  # noqa: E402


def _rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def _build(variant, dtype=torch.float32, n=24):
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    shape = (2, 3, 4)
    kw = dict(rotary=dict(rotary_position_emb=True), axial=dict(axial_position_emb=True, axial_position_shape=(5, 6)), tied=dict(tie_embed=True),
              emb_dropout=dict(), axial_default=dict(axial_position_emb=True))[variant]
    cfg = P.PerformerConfig(num_tokens=33, max_seq_len=n, dim=32, depth=2, heads=4, dim_head=64, local_attn_heads=2, local_window_size=6, spatial_shape=shape, **kw)
    st = P.init_state(cfg, seed=5, spatial_index_len=n - 1)
    g = torch.Generator().manual_seed(9)
    for k in st:
        if k.endswith(".g"):
            st[k] = torch.tensor(0.4)
    if variant == "rotary":
        del st["pos_emb.emb.weight"]
        st["pos_emb.emb"] = P.fixed_position_table(32, n)
    if variant.startswith("axial"):
        del st["pos_emb.emb.weight"]
        s0, s1 = cfg.axial_position_shape or (1, 64)
        st["pos_emb.weights_0"] = torch.randn(1, s0, 1, 32, generator=g)
        st["pos_emb.weights_1"] = torch.randn(1, 1, s1, 32, generator=g)
    if variant == "tied":
        del st["to_out.weight"], st["to_out.bias"]
    o = Ordering("raster_scan", 3, (1,) + shape, (False,) * 3, (), ())
    net = Performer(num_tokens=33, max_seq_len=n, dim=32, depth=2, heads=4, ordering=o, dim_head=64, local_attn_heads=2, local_window_size=6, use_rezero=True,
                    spatial_position_emb="absolute", spatial_shape=shape, feature_redraw_interval=None, compute_dtype=dtype,
                    emb_dropout=0.25 if variant == "emb_dropout" else 0.0, **kw)
    missing, unexpected = net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(any(t in k for t in ("spatial_indices_sequence", "calls_since", "layer_pos_emb", "inv_freq")) for k in missing), missing
    return cfg, st, o, net.cuda().train(), shape, g


def test_state_dict_keys_of_the_option_modules():
    """the keys a reference checkpoint with these options would carry: `layer_pos_emb.emb` + `pos_emb.emb` (rotary), `pos_emb.weights_{0,1}` (axial, default grid
    (ceil(max_seq_len / 64), 64), performer.py:142-144), no `to_out.*` when tied"""
    _, _, _, net, _, _ = _build("rotary")
    sd = net.state_dict()
    assert sd["layer_pos_emb.emb"].shape == (24, 64) and sd["pos_emb.emb"].shape == (24, 32)
    assert torch.allclose(sd["layer_pos_emb.emb"].cpu(), P.fixed_position_table(64, 24), atol=1e-6)
    _, _, _, net, _, _ = _build("axial_default")
    sd = net.state_dict()
    assert sd["pos_emb.weights_0"].shape == (1, 1, 1, 32) and sd["pos_emb.weights_1"].shape == (1, 1, 64, 32)
    _, _, _, net, _, _ = _build("tied")
    assert not any(k.startswith("to_out") for k in net.state_dict()) and net.to_out is None


@pytest.mark.parametrize("variant", ["rotary", "axial", "axial_default", "tied", "emb_dropout"])
def test_wrapper_options_match_oracle(variant):
    from synthanatomy_amd.losses.transformer import CELoss
    cfg, st, o, net, shape, g = _build(variant)
    n = cfg.max_seq_len
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    tok = torch.randint(0, 33, (2, n), generator=g)
    tgt = torch.randint(0, 32, (2, n), generator=g)
    out = net(tok.cuda())
    CELoss()(out.transpose(1, 2), tgt.cuda()).backward()
    torch.cuda.synchronize()
    mask = None
    if variant == "emb_dropout":
        mask = net._last_emb_mask.cpu()
        vals = sorted(torch.unique(mask).tolist())
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / 0.75) < 1e-6 and 0.6 < float((mask > 0).float().mean()) < 0.9
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "projection_matrix" not in k and k != "pos_emb.emb"}
    stt = dict(st)
    stt.update(leaf)
    ref = P.forward(stt, cfg, tok, seqs, emb_mask=mask)
    P.ce_loss(ref, tgt).backward()
    assert out.shape == ref.shape == (2, n, 33) and _rel(out, ref) < 1e-3, _rel(out, ref)
    params = dict(net.named_parameters())
    checked = 0
    for k, p in leaf.items():
        if k in params and p.grad is not None and float(p.grad.abs().max()) > 0:
            assert params[k].grad is not None, k
            assert _rel(params[k].grad, p.grad) < 3e-3, (k, _rel(params[k].grad, p.grad))
            checked += 1
    assert checked > 20
    if variant.startswith("axial"):
        assert float(params["pos_emb.weights_1"].grad.abs().max()) > 0
    if variant == "emb_dropout":        # eval(): the dropout is the identity (and the oracle without a mask agrees)
        net.eval()
        with torch.no_grad():
            assert _rel(net(tok.cuda()), P.forward(st, cfg, tok, seqs)) < 1e-3


def test_rotary_global_heads_on_the_fused_bf16_path_and_sampling_falls_back_to_the_quadratic_loop():
    """throughput mode: the rotation sits in front of the fused FAVOR+ kernels (forward) and behind them (adjoint on dq / dk, bf16 mirrors off); the oracle with
    bf16 rounding is not emulated here -- the gate is the bf16 path against the SAME network in fp32.  Stateful sampling refuses the option, the default falls back."""
    from synthanatomy_amd.losses.transformer import CELoss
    cfg, st, o, net32, shape, g = _build("rotary")
    _, _, _, net16, _, _ = _build("rotary", dtype=torch.bfloat16)
    tok = torch.randint(0, 33, (2, cfg.max_seq_len), generator=g).cuda()
    tgt = torch.randint(0, 32, (2, cfg.max_seq_len), generator=g).cuda()
    outs = []
    for net in (net32, net16):
        out = net(tok)
        CELoss()(out.transpose(1, 2), tgt).backward()
        outs.append(out)
    torch.cuda.synchronize()
    assert _rel(outs[1], outs[0]) < 2e-2
    p32, p16 = dict(net32.named_parameters()), dict(net16.named_parameters())
    for k in ("performer.net.layers.0.0.fn.to_q.weight", "performer.net.layers.1.0.fn.to_k.weight", "token_emb.weight"):
        assert _rel(p16[k].grad, p32[k].grad) < 6e-2, (k, _rel(p16[k].grad, p32[k].grad))
    net32.eval()
    with pytest.raises(NotImplementedError):
        net32.sample(torch.zeros(1, 1, dtype=torch.long), stateful=True)
    seq = net32.sample(torch.zeros(1, 1, dtype=torch.long, device="cuda"), sample=False)
    assert seq.numel() == 24          # sequence_to_grid: the 2 x 3 x 4 latent grid of one sample


def test_rotary_pairs_kernel_against_the_oracle_and_its_adjoint():
    """sa_rotary_pairs alone: head blocks inside wider rows (as q | k sit in the q|k|v matrix), two operands in one launch, in place; the transpose flag is the adjoint
    (<R x, y> = <x, R^T y>) and undoes the rotation (R^T R = I)."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    B, N, G, dh, extra = 2, 37, 3, 64, 64
    inner = G * dh + extra
    stride = 3 * inner
    R = B * N
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(R, stride, generator=g)
    tab = P.fixed_position_table(dh, 50)
    d = qkv.clone().cuda()
    tabd = tab.cuda().contiguous()
    _ffi.check(lib.sa_rotary_pairs(_ffi.ptr(d), stride, 0, G, dh, _ffi.ptr(tabd), _ffi.ptr(d), stride, 0, N, R, 0, 2, inner, inner, st), "sa_rotary_pairs")
    torch.cuda.synchronize()
    un = lambda t, o: t[:, o:o + G * dh].reshape(B, N, G, dh).permute(0, 2, 1, 3)
    qr, kr = P.apply_rotary_pos_emb(un(qkv, 0), un(qkv, inner), tab[:N])
    assert torch.allclose(un(d.cpu(), 0), qr, atol=1e-6) and torch.allclose(un(d.cpu(), inner), kr, atol=1e-6)
    keep = qkv.clone()
    keep[:, :G * dh] = 0
    keep[:, inner:inner + G * dh] = 0
    got = d.cpu().clone()
    got[:, :G * dh] = 0
    got[:, inner:inner + G * dh] = 0
    assert torch.equal(got, keep)                     # nothing outside the global-head columns of q and k is touched
    y = torch.randn(R, stride, generator=g).cuda()
    yt = y.clone()
    _ffi.check(lib.sa_rotary_pairs(_ffi.ptr(yt), stride, 0, G, dh, _ffi.ptr(tabd), _ffi.ptr(yt), stride, 0, N, R, 1, 2, inner, inner, st), "sa_rotary_pairs^T")
    cols = torch.cat((torch.arange(G * dh), inner + torch.arange(G * dh))).cuda()
    lhs = (d[:, cols].double() * y[:, cols].double()).sum()
    rhs = (qkv.cuda()[:, cols].double() * yt[:, cols].double()).sum()
    assert abs(float(lhs - rhs)) < 1e-6 * abs(float(lhs)) + 1e-6
    back = d.clone()
    _ffi.check(lib.sa_rotary_pairs(_ffi.ptr(back), stride, 0, G, dh, _ffi.ptr(tabd), _ffi.ptr(back), stride, 0, N, R, 1, 2, inner, inner, st), "sa_rotary_pairs^T")
    torch.cuda.synchronize()
    assert torch.allclose(back.cpu(), qkv, atol=1e-5)
    assert lib.sa_rotary_pairs(_ffi.ptr(d), stride + 1, 0, G, dh, _ffi.ptr(tabd), _ffi.ptr(d), stride, 0, N, R, 0, 1, 0, 0, st) == _ffi.SA_EUNSUPPORTED
