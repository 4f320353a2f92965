"""GPU: the product Performer (HIP kernels) against the CPU oracle restatement (oracle/performer_ref.py; parity UNPINNED
against the third-party package -- see that file) on seeded inputs: logits within 1e-3 relative, every parameter gradient,
and the individual attention kernels against the closed-form (quadratic / dense band) statements."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ordering_ref, performer_ref as P  # noqa: E402

REL = 1e-3


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def _build(cfg, st, dtype=torch.float32, rezero=True):
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    o = Ordering("raster_scan", 3, (1,) + tuple(cfg.spatial_shape), (False,) * 3, (), ())
    net = Performer(num_tokens=cfg.num_tokens, max_seq_len=cfg.max_seq_len, dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, ordering=o,
                    dim_head=cfg.dim_head, local_attn_heads=cfg.local_attn_heads, local_window_size=cfg.local_window_size, use_rezero=rezero,
                    spatial_position_emb="absolute", spatial_shape=cfg.spatial_shape, feature_redraw_interval=None, compute_dtype=dtype)
    missing, unexpected = net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(("spatial_indices_sequence" in k or "inv_freq" in k or "calls_since" in k) for k in missing), missing
    return net.cuda(), o


@pytest.mark.parametrize("rezero,B,shape,window,local", [(True, 2, (2, 3, 4), 6, 2), (False, 1, (3, 3, 5), 64, 1), (True, 3, (2, 2, 5), 7, 4), (True, 2, (2, 3, 3), 5, 0)])
def test_forward_and_gradients_match_oracle(rezero, B, shape, window, local):
    n = int(np.prod(shape))
    cfg = P.PerformerConfig(num_tokens=33, max_seq_len=n, dim=32, depth=2, heads=4, dim_head=64, local_attn_heads=local, local_window_size=window,
                            spatial_shape=shape, use_rezero=rezero)
    st = P.init_state(cfg, seed=n)
    if rezero:
        for k in st:
            if k.endswith(".g"):
                st[k] = torch.tensor(0.4)  # the 1e-3 init would hide errors behind the residual path
    net, o = _build(cfg, st, rezero=rezero)
    net.train()
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    torch.manual_seed(0)
    tok = torch.randint(0, 33, (B, n))
    tgt = torch.randint(0, 32, (B, n))
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "projection_matrix" not in k}
    stt = dict(st)
    stt.update(leaf)
    ref = P.forward(stt, cfg, tok, seqs)
    ref_loss = P.ce_loss(ref, tgt)
    ref_loss.backward()
    from synthanatomy_amd.losses.transformer import CELoss
    out = net(tok.cuda())
    assert out.shape == (B, n, 33)
    assert _rel(out, ref) < REL
    loss = CELoss()(out.transpose(1, 2), tgt.cuda())
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item()) + 1e-6
    loss.backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    worst = ("", 0.0)
    for k, p in leaf.items():
        if p.grad is None:
            continue
        g = params[k].grad
        assert g is not None, k
        e = _rel(g, p.grad)
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < 3e-3, worst


def test_bf16_projections_close_to_fp32_oracle():
    shape = (2, 3, 4)
    cfg = P.PerformerConfig(num_tokens=33, max_seq_len=24, dim=64, depth=2, heads=4, dim_head=64, local_attn_heads=2, local_window_size=8, spatial_shape=shape)
    st = P.init_state(cfg, seed=5)
    net, o = _build(cfg, st, dtype=torch.bfloat16)
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    torch.manual_seed(1)
    tok = torch.randint(0, 33, (2, 24))
    with torch.no_grad():
        out = net.eval()(tok.cuda())
    assert _rel(out, P.forward(st, cfg, tok, seqs)) < 3e-2


def test_bf16_mode_gradients_follow_fp32_oracle():
    """Throughput mode end to end (bf16 dense layers, split-bf16 scans and local attention): parameter gradients against the fp32 oracle.
    The bound is the bf16 rounding of the dense layers; the ill-conditioned query-side gradients are compared in aggregate norm."""
    shape = (2, 4, 4)
    n = 32
    cfg = P.PerformerConfig(num_tokens=33, max_seq_len=n, dim=64, depth=2, heads=4, dim_head=64, local_attn_heads=2, local_window_size=8, spatial_shape=shape)
    st = P.init_state(cfg, seed=9)
    for k in st:
        if k.endswith(".g"):
            st[k] = torch.tensor(0.4)
    net, o = _build(cfg, st, dtype=torch.bfloat16)
    net.train()
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    torch.manual_seed(3)
    tok = torch.randint(0, 33, (2, n))
    tgt = torch.randint(0, 32, (2, n))
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "projection_matrix" not in k}
    stt = dict(st)
    stt.update(leaf)
    P.ce_loss(P.forward(stt, cfg, tok, seqs), tgt).backward()
    from synthanatomy_amd.losses.transformer import CELoss
    loss = CELoss()(net(tok.cuda()).transpose(1, 2), tgt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    num = den = 0.0
    for k, p in leaf.items():
        if p.grad is None:
            continue
        g = params[k].grad.cpu().double()
        num += float((g - p.grad.double()).pow(2).sum())
        den += float(p.grad.double().pow(2).sum())
        if not any(t in k for t in ("to_q", "to_k")):
            assert _rel(g, p.grad) < 5e-2, k
    assert (num / den) ** 0.5 < 2e-2


def test_scan_exact_flag_selects_fp32_products():
    """state_flags bit 2 of the fused scans: exact-fp32 MFMA products; both paths agree to the split-bf16 error (~1e-5) and differ bitwise."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(11)
    B, N, G, m, LDF, dv = 2, 150, 2, 266, 272, 64
    a = torch.zeros(B, N, G, LDF, device="cuda")
    c = torch.zeros(B, N, G, LDF, device="cuda")
    a[..., :m] = torch.rand(B, N, G, m, device="cuda") + 0.01
    c[..., :m] = torch.rand(B, N, G, m, device="cuda") + 0.01
    bb = torch.randn(B * N, G * dv, device="cuda")
    ws = torch.empty(lib.sa_favor_scan_workspace_bytes(B, N, G, LDF, dv) // 4, device="cuda")
    outs = []
    for flags in (0, 4):
        y = torch.zeros(B * N, G * dv, device="cuda")
        inv = torch.zeros(B * N * G, device="cuda")
        _ffi.check(lib.sa_favor_scan_a_norm(_ffi.ptr(a), _ffi.ptr(c), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(y), G * dv, 0, _ffi.ptr(inv), 1e-6, B, N, G, LDF, dv,
                                            _ffi.ptr(ws), flags, st))
        outs.append((y, inv))
    ref = torch.einsum("bigm,bjgm,ij,bjgd->bigd", c.double(), a.double(), torch.tril(torch.ones(N, N, dtype=torch.float64, device="cuda")), bb.view(B, N, G, dv).double())
    ref = ref / torch.einsum("bigm,bigm->big", c.double(), a.double().cumsum(1) + 1e-6)[..., None]
    e_split, e_exact = _rel(outs[0][0].view(B, N, G, dv), ref), _rel(outs[1][0].view(B, N, G, dv), ref)
    assert e_exact < 2e-6 and e_split < 5e-5 and not torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("rows,m,LDF,heads", [(1, 266, 272, 1), (130, 266, 272, 2), (1000, 100, 112, 1), (67200, 266, 272, 8), (77, 256, 256, 1)])
def test_projection_kernels_against_matmul(rows, m, LDF, heads):
    """sa_favor_project / sa_favor_project_bwd (split-bf16, projection matrix staged once per block) against fp64 matmuls: ragged row counts,
    padded feature columns written as zeros, rows that are head blocks of wider rows, addend."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(rows + m)
    wide = heads * 64 + 32                               # head blocks live in a wider matrix
    xw = torch.randn(rows // heads, wide, device="cuda")
    x = xw[:, :heads * 64].reshape(rows, 64)
    P_ = torch.randn(m, 64, device="cuda") * 0.35
    dd = torch.full((rows, LDF), float("nan"), device="cuda")
    _ffi.check(lib.sa_favor_project(_ffi.ptr(xw), wide, heads, _ffi.ptr(P_), _ffi.ptr(dd), None, rows, m, LDF, 64, st))
    ref = x.double() @ P_.double().t()
    assert _rel(dd[:, :m], ref) < 2e-5
    assert LDF == m or float(dd[:, m:].abs().max()) == 0.0
    # the global (value, index) maximum from the accumulators == the separate pass over dd (feature maps built from either agree bitwise)
    gws = torch.empty(2, dtype=torch.int64, device="cuda")
    dd2 = torch.empty_like(dd)
    _ffi.check(lib.sa_favor_project(_ffi.ptr(xw), wide, heads, _ffi.ptr(P_), _ffi.ptr(dd2), _ffi.ptr(gws), rows, m, LDF, 64, st))
    assert torch.equal(dd2, dd)
    fa, fb = torch.empty_like(dd), torch.empty_like(dd)
    gws0 = torch.zeros(2, dtype=torch.int64, device="cuda")
    _ffi.check(lib.sa_favor_features_fwd(_ffi.ptr(dd), _ffi.ptr(xw), wide, 0, heads, 64, 0, _ffi.ptr(fa), _ffi.ptr(gws0), rows, m, LDF, st))
    _ffi.check(lib.sa_favor_features_fwd(_ffi.ptr(dd), _ffi.ptr(xw), wide, 0, heads, 64, 2, _ffi.ptr(fb), _ffi.ptr(gws), rows, m, LDF, st))
    assert int(gws[0]) == int(gws0[0]) and torch.equal(fa, fb)
    flat = dd[:, :m].max()
    assert float(flat) == float(dd.view(-1)[0xffffffff - (int(gws[0]) & 0xffffffff)])
    # projection + query feature map in one launch == the two launches
    fq = torch.empty_like(dd)
    _ffi.check(lib.sa_favor_features_fwd(_ffi.ptr(dd), _ffi.ptr(xw), wide, 0, heads, 64, 1, _ffi.ptr(fq), None, rows, m, LDF, st))
    dd3, fq3 = torch.full_like(dd, float("nan")), torch.full_like(dd, float("nan"))
    _ffi.check(lib.sa_favor_project_features(_ffi.ptr(xw), wide, heads, _ffi.ptr(P_), _ffi.ptr(dd3), _ffi.ptr(fq3), rows, m, LDF, 64, st))
    assert torch.equal(dd3, dd) and _rel(fq3, fq) < 2e-6
    assert LDF == m or float(fq3[:, m:].abs().max()) == 0.0
    g = torch.randn(rows, LDF, device="cuda")
    g[:, m:] = 7.0                                        # padded gradient columns must not leak (the staged projection rows are zero)
    dxw = torch.full((rows // heads, wide), float("nan"), device="cuda")
    dxw[:, :heads * 64] = torch.randn(rows // heads, heads * 64, device="cuda")   # addend in place (what the feature-map backward left there)
    add = dxw[:, :heads * 64].reshape(rows, 64).clone()
    _ffi.check(lib.sa_favor_project_bwd(_ffi.ptr(g), _ffi.ptr(P_), _ffi.ptr(dxw), _ffi.ptr(dxw), wide, heads, rows, m, LDF, 64, st))
    refb = g[:, :m].double() @ P_.double() + add.double()
    assert _rel(dxw[:, :heads * 64].reshape(rows, 64), refb) < 2e-5
    assert bool(torch.isnan(dxw[:, heads * 64:]).all())   # nothing outside the head blocks is touched
    _ffi.check(lib.sa_favor_project_bwd(_ffi.ptr(g), _ffi.ptr(P_), None, _ffi.ptr(dxw), wide, heads, rows, m, LDF, 64, st))
    assert _rel(dxw[:, :heads * 64].reshape(rows, 64), g[:, :m].double() @ P_.double()) < 2e-5


@pytest.mark.parametrize("with_sink", [False, True])
def test_fused_qkv_projection_matches_separate_layers(with_sink):
    """to_q / to_k / to_v run as ONE dense layer when their weights sit back to back in the flat parameter buffer (runtime.optim.FlatParams):
    same logits and gradients as the three separate layers, with the gradient buffers adjacent (sink) or not (scratch matrix)."""
    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.runtime.ddp import GradReducer
    from synthanatomy_amd.runtime.optim import FlatParams
    shape, n = (2, 4, 5), 40
    cfg = P.PerformerConfig(num_tokens=33, max_seq_len=n, dim=64, depth=2, heads=4, dim_head=64, local_attn_heads=2, local_window_size=8, spatial_shape=shape)
    st = P.init_state(cfg, seed=4)
    for k in st:
        if k.endswith(".g"):
            st[k] = torch.tensor(0.3)
    torch.manual_seed(7)
    tok = torch.randint(0, 33, (2, n)).cuda()
    tgt = torch.randint(0, 32, (2, n)).cuda()
    res = []
    from synthanatomy_amd import debug
    for fused in (True, False):
        net, _ = _build(cfg, st, dtype=torch.bfloat16)
        net.train()
        flat = FlatParams(net.parameters())
        if with_sink:
            net.set_grad_sink(GradReducer(flat))
        with debug.override(no_fused_qkv=not fused):   # the layer engines read the switch at launch time
            out = net(tok)
            CELoss()(out.transpose(1, 2), tgt).backward()
            torch.cuda.synchronize()
        eng = net._chain.layers[0]
        assert ("to_qkv" in eng.ops) == fused
        res.append((out.detach().float().clone(), flat.grad.clone()))
    assert _rel(res[0][0], res[1][0]) < 2e-3
    assert _rel(res[0][1], res[1][1]) < 5e-3 and float(res[0][1].abs().max()) > 0


@pytest.mark.parametrize("m,LDF,N", [(120, 128, 100), (256, 256, 130), (16, 16, 70), (200, 208, 64), (266, 272, 257)])
@pytest.mark.parametrize("reverse", [0, 1])
def test_split_scans_other_feature_counts(m, LDF, N, reverse):
    """The split-bf16 chunk kernels stage features in slabs (144 wide for the state sums, 64 wide for the outputs): feature counts that end inside a
    slab / a 32-step reduction block, against the exact-fp32 kernels (state_flags bit 2) on every fused entry point."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(m + N)
    B, G, dv = 2, 2, 64
    a = torch.zeros(B, N, G, LDF, device="cuda")
    c = torch.zeros(B, N, G, LDF, device="cuda")
    a[..., :m] = torch.rand(B, N, G, m, device="cuda") + 0.01
    c[..., :m] = torch.rand(B, N, G, m, device="cuda") + 0.01
    bb = torch.randn(B * N, G * dv, device="cuda")
    cc = torch.randn(B * N, G * dv, device="cuda")
    bs = torch.rand(B, N, G, device="cuda") + 0.5
    ws = torch.empty(lib.sa_favor_scan_workspace_bytes(B, N, G, LDF, dv) // 4, device="cuda")
    outs = []
    for xf in (0, 4):
        yn = torch.zeros(B * N, G * dv, device="cuda")
        inv = torch.zeros(B * N * G, device="cuda")
        if not reverse:
            _ffi.check(lib.sa_favor_scan_a_norm(_ffi.ptr(a), _ffi.ptr(c), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(yn), G * dv, 0, _ffi.ptr(inv), 1e-6, B, N, G, LDF, dv,
                                                _ffi.ptr(ws), xf, st))
        y1 = torch.zeros(B, N, G, LDF, device="cuda")
        _ffi.check(lib.sa_favor_scan_b_cum(_ffi.ptr(a), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(bs), _ffi.ptr(cc), G * dv, 0, None, _ffi.ptr(y1), _ffi.ptr(bs), 1, 0.25,
                                           B, N, G, LDF, dv, reverse, _ffi.ptr(ws), xf, st))
        y2 = torch.zeros(B, N, G, LDF, device="cuda")
        _ffi.check(lib.sa_favor_scan_b_cum(_ffi.ptr(a), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(bs), _ffi.ptr(cc), G * dv, 0, None, _ffi.ptr(y2), _ffi.ptr(bs), 2, 0.0,
                                           B, N, G, LDF, dv, reverse, _ffi.ptr(ws), xf, st))
        y3 = torch.zeros(B * N, G * dv, device="cuda")
        _ffi.check(lib.sa_favor_scan_a_state(_ffi.ptr(a), _ffi.ptr(c), _ffi.ptr(bb), G * dv, 0, _ffi.ptr(bs), _ffi.ptr(y3), G * dv, 0, None, B, N, G, LDF, dv,
                                             reverse, 0, _ffi.ptr(ws), 3 | xf, st))
        outs.append((yn, inv, y1[..., :m], y2[..., :m], y3))
    for s_, e_ in zip(*outs):
        assert bool(torch.isfinite(s_).all())
        assert _rel(s_, e_) < 5e-5


def test_rotary_and_rezero_backward_kernels():
    """sa_rotary (forward, adjoint, accumulate; head blocks of wider rows) against the rotate-half formula, and the 4-wide bf16 path of
    sa_rezero_bwd against torch."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(21)
    B, N, L, dh, wide = 2, 37, 3, 64, 288
    R = B * N
    x = torch.randn(R, wide, device="cuda")
    fr = torch.einsum("i,j->ij", torch.arange(N, device="cuda", dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, dh, 2, device="cuda").float() / dh)))
    fr = torch.cat((fr, fr), -1)
    cosb, sinb = fr.cos().contiguous(), fr.sin().contiguous()
    xs = x[:, 64:64 + L * dh].reshape(B, N, L, dh)
    rot = lambda t: torch.cat((-t[..., dh // 2:], t[..., :dh // 2]), -1)
    ref = xs * cosb[None, :, None, :] + rot(xs) * sinb[None, :, None, :]
    y = torch.zeros(R, L * dh, device="cuda")
    _ffi.check(lib.sa_rotary(_ffi.ptr(x), wide, 64, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(y), L * dh, 0, N, R, 0, 0, st))
    assert _rel(y.view(B, N, L, dh), ref) < 1e-6
    # adjoint: <rot(x), g> == <x, rot^T(g)>, written into a wider matrix, then accumulated once more
    g = torch.randn(R, L * dh, device="cuda")
    gx = torch.zeros(R, wide, device="cuda")
    _ffi.check(lib.sa_rotary(_ffi.ptr(g), L * dh, 0, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(gx), wide, 64, N, R, 1, 0, st))
    lhs = float((y.double() * g.double()).sum())
    rhs = float((x[:, 64:64 + L * dh].double() * gx[:, 64:64 + L * dh].double()).sum())
    assert abs(lhs - rhs) < 1e-6 * abs(lhs) + 1e-6
    assert float(gx[:, :64].abs().max()) == 0.0 and float(gx[:, 64 + L * dh:].abs().max()) == 0.0
    # two operands in one launch (q | k as column blocks of one matrix, 2 * wide apart ... here: two stacked gradient matrices -> two column blocks), with the
    # bf16 mirror of what is written (the operand of the q|k|v weight / data gradient)
    g2 = torch.randn(2, R, L * dh, device="cuda")
    both = torch.full((R, 2 * wide), 5.0, device="cuda")
    both_lp = torch.full((R, 2 * wide), 5.0, device="cuda", dtype=torch.bfloat16)
    _ffi.check(lib.sa_rotary_groups(_ffi.ptr(g2), L * dh, 0, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(both), 2 * wide, 64, N, R, 1, 0, 2, R * L * dh, wide,
                                    _ffi.ptr(both_lp), st))
    for i in range(2):
        one = torch.zeros(R, wide, device="cuda")
        _ffi.check(lib.sa_rotary(_ffi.ptr(g2[i]), L * dh, 0, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(one), wide, 64, N, R, 1, 0, st))
        assert torch.equal(both[:, i * wide + 64:i * wide + 64 + L * dh], one[:, 64:64 + L * dh])
    assert torch.equal(both_lp, both.to(torch.bfloat16))     # written columns: rounded copies; everything else untouched (5.0 is exact in bf16)
    once = gx.clone()
    _ffi.check(lib.sa_rotary(_ffi.ptr(g), L * dh, 0, L, dh, _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(gx), wide, 64, N, R, 1, 1, st))
    assert _rel(gx, 2 * once) < 1e-6
    # ReZero backward, bf16 F / dF
    n = 4096 * 3
    dy = torch.randn(n, device="cuda")
    F_ = torch.randn(n, device="cuda").bfloat16()
    gate = torch.tensor([0.37], device="cuda")
    dF = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    dg = torch.zeros(1, device="cuda")
    _ffi.check(lib.sa_rezero_bwd(_ffi.ptr(dy), _ffi.ptr(F_), _ffi.dtype_id(torch.bfloat16), _ffi.ptr(gate), _ffi.ptr(dF), _ffi.dtype_id(torch.bfloat16), _ffi.ptr(dg), n, st))
    assert torch.equal(dF, (dy * 0.37).bfloat16())
    assert abs(float(dg) - float((dy.double() * F_.double()).sum())) < 1e-3
    # ReZero forward (bf16 F, fp32 y + bf16 copy) and GELU (bf16 -> bf16), vector paths
    y = torch.empty(n, device="cuda")
    y_lp = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    bf = _ffi.dtype_id(torch.bfloat16)
    _ffi.check(lib.sa_rezero_fwd(_ffi.ptr(dy), _ffi.ptr(F_), bf, _ffi.ptr(gate), _ffi.ptr(y), _ffi.ptr(y_lp), bf, n, st))
    ref = dy + 0.37 * F_.float()
    assert _rel(y, ref) < 1e-6 and torch.equal(y_lp, y.bfloat16())
    h = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    _ffi.check(lib.sa_gelu(_ffi.ptr(F_), bf, _ffi.ptr(h), bf, n, st))
    assert _rel(h.float(), torch.nn.functional.gelu(F_.float())) < 4e-3


@pytest.mark.parametrize("R,G,m,LDF", [(37, 2, 266, 272), (1400, 8, 266, 272), (65, 1, 100, 112)])
@pytest.mark.parametrize("is_query", [1, 0])
def test_fused_feature_projection_backward(R, G, m, LDF, is_query):
    """sa_favor_features_project_bwd (d loss / d dd never written) against sa_favor_features_bwd + sa_favor_project_bwd: query rows (own maximum)
    and key rows (global maximum, fix-up launch), head blocks of wider rows."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(R + m + is_query)
    rows, wide = R * G, G * 64 + 64
    xw = torch.randn(R, wide, device="cuda") * 0.7
    P_ = torch.randn(m, 64, device="cuda") * 0.35
    dd = torch.empty(rows, LDF, device="cuda")
    _ffi.check(lib.sa_favor_project(_ffi.ptr(xw), wide, G, _ffi.ptr(P_), _ffi.ptr(dd), None, rows, m, LDF, 64, st))
    feat = torch.empty_like(dd)
    gws = torch.zeros(2, dtype=torch.int64, device="cuda")
    _ffi.check(lib.sa_favor_features_fwd(_ffi.ptr(dd), _ffi.ptr(xw), wide, 0, G, 64, is_query, _ffi.ptr(feat), None if is_query else _ffi.ptr(gws), rows, m, LDF, st))
    dfeat = torch.randn(rows, LDF, device="cuda")
    # two-launch reference
    ddd = torch.empty_like(dd)
    ref = torch.full((R, wide), float("nan"), device="cuda")
    tsum = torch.empty(rows, device="cuda")
    _ffi.check(lib.sa_favor_features_bwd(_ffi.ptr(dfeat), _ffi.ptr(feat), _ffi.ptr(dd), _ffi.ptr(xw), wide, 0, G, 64, is_query, _ffi.ptr(ddd), _ffi.ptr(ref),
                                         None if is_query else _ffi.ptr(gws), None if is_query else _ffi.ptr(tsum), rows, m, LDF, st))
    _ffi.check(lib.sa_favor_project_bwd(_ffi.ptr(ddd), _ffi.ptr(P_), _ffi.ptr(ref), _ffi.ptr(ref), wide, G, rows, m, LDF, 64, st))
    got = torch.full((R, wide), float("nan"), device="cuda")
    tsum2 = torch.empty(rows, device="cuda")
    _ffi.check(lib.sa_favor_features_project_bwd(_ffi.ptr(dfeat), _ffi.ptr(feat), _ffi.ptr(dd), _ffi.ptr(xw), wide, G, _ffi.ptr(P_), is_query, _ffi.ptr(got),
                                                 None if is_query else _ffi.ptr(gws), None if is_query else _ffi.ptr(tsum2), rows, m, LDF, 64, st))
    assert _rel(got[:, :G * 64], ref[:, :G * 64]) < 3e-5
    assert bool(torch.isnan(got[:, G * 64:]).all())
    if not is_query:
        assert abs(float(tsum2[0]) - float(tsum.double().sum())) < 1e-4 * float(tsum.double().abs().sum())


@pytest.mark.parametrize("R,C", [(37, 32), (8400, 512), (130, 200), (65, 640)])
def test_layernorm_kernels(R, C):
    """sa_layernorm_fwd / sa_layernorm_bwd against torch (the backward reduces dw / db per 32-row block for C <= 512, per element above)."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(R + C)
    x = torch.randn(R, C, device="cuda") * 2 + 0.3
    w = torch.randn(C, device="cuda")
    b = torch.randn(C, device="cuda")
    dy = torch.randn(R, C, device="cuda")
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5)
    ref.backward(dy.double())
    y = torch.empty_like(x)
    stats = torch.empty(2 * R, device="cuda")
    _ffi.check(lib.sa_layernorm_fwd(_ffi.ptr(x), _ffi.ptr(w), _ffi.ptr(b), _ffi.ptr(y), None, 0, _ffi.ptr(stats), R, C, 1e-5, st))
    assert _rel(y, ref.detach()) < 1e-5
    dx = torch.empty_like(x)
    dw = torch.zeros(C, device="cuda")
    db = torch.zeros(C, device="cuda")
    _ffi.check(lib.sa_layernorm_bwd(_ffi.ptr(dy), _ffi.ptr(x), _ffi.ptr(w), _ffi.ptr(stats), _ffi.ptr(dx), _ffi.ptr(dw), _ffi.ptr(db), R, C, st))
    assert _rel(dx, xr.grad) < 1e-5 and _rel(dw, wr.grad) < 1e-4 and _rel(db, br.grad) < 1e-4


@pytest.mark.parametrize("N,segmented", [(37, False), (37, True), (300, True), (1400, True)])
def test_causal_scan_kernels_against_quadratic_form(N, segmented):
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(2)
    B, G, m, LDF, dv = 2, 3, 266, 272, 64
    qf = torch.zeros(B, N, G, LDF)
    kf = torch.zeros(B, N, G, LDF)
    qf[..., :m] = torch.rand(B, N, G, m) + 0.01
    kf[..., :m] = torch.rand(B, N, G, m) + 0.01
    v = torch.randn(B * N, 2 * G * dv)  # global heads in the first G*dv columns of a wider row
    ref = P.causal_linear_attention(qf[..., :m].permute(0, 2, 1, 3), kf[..., :m].permute(0, 2, 1, 3), v[:, :G * dv].view(B, N, G, dv).permute(0, 2, 1, 3))
    qd, kd, vd = qf.cuda(), kf.cuda(), v.cuda()
    Z = torch.empty_like(kd)
    inv = torch.empty(B * N * G, device="cuda")
    out = torch.zeros(B * N, 2 * G * dv, device="cuda")
    ws = torch.empty(lib.sa_favor_scan_workspace_bytes(B, N, G, LDF, dv) // 4, device="cuda") if segmented else None
    _ffi.check(lib.sa_cumsum_rows(_ffi.ptr(kd), None, _ffi.ptr(Z), B, N, G, LDF, 0, _ffi.ptr(ws), st))
    _ffi.check(lib.sa_favor_den(_ffi.ptr(qd), _ffi.ptr(Z), 1e-6, _ffi.ptr(inv), B * N * G, m, LDF, st))
    _ffi.check(lib.sa_favor_scan_a(_ffi.ptr(kd), _ffi.ptr(qd), _ffi.ptr(vd), 2 * G * dv, 0, None, _ffi.ptr(out), 2 * G * dv, 0, _ffi.ptr(inv), B, N, G, LDF, dv, 0, 0,
                                   _ffi.ptr(ws), st))
    got = out[:, :G * dv].view(B, N, G, dv).permute(0, 2, 1, 3)
    assert _rel(got, ref) < 1e-4
    assert float(out[:, G * dv:].abs().max()) == 0.0  # only the addressed head block is written


@pytest.mark.parametrize("N,reverse", [(37, 0), (37, 1), (200, 1), (1400, 0)])
def test_causal_scan_variants_against_einsum(N, reverse):
    """Every mode the backward uses: reversed order, per-position scales, accumulate, scan B with its extra term; VALU path == MFMA path."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(5)
    B, G, m, LDF, dv = 2, 2, 266, 272, 64
    a = torch.zeros(B, N, G, LDF, dtype=torch.float64)
    c = torch.zeros(B, N, G, LDF, dtype=torch.float64)
    a[..., :m] = torch.rand(B, N, G, m) + 0.01
    c[..., :m] = torch.rand(B, N, G, m) + 0.01
    bb = torch.randn(B, N, G, dv, dtype=torch.float64)
    cc = torch.randn(B, N, G, dv, dtype=torch.float64)
    bs = torch.rand(B, N, G, dtype=torch.float64) + 0.5
    ys = torch.rand(B, N, G, dtype=torch.float64) + 0.5
    ev = torch.zeros(B, N, G, LDF, dtype=torch.float64)
    ev[..., :m] = torch.randn(B, N, G, m)
    y0 = torch.randn(B, N, G, dv, dtype=torch.float64)
    tri = torch.tril(torch.ones(N, N, dtype=torch.float64))
    if reverse:
        tri = tri.t()                                  # j >= i
    bsc = bb * bs[..., None]
    refA = torch.einsum("bigm,bjgm,ij,bjgd->bigd", c, a, tri, bsc) * ys[..., None] + y0
    refB = torch.einsum("bjgm,bjgd,ij,bigd->bigm", a, bsc, tri, cc * ys[..., None]) + bs[..., None] * (ev + 0.25)
    refB[..., m:] = refB[..., m:]                      # padding columns carry only the extra term
    f = lambda t: t.float().cuda().contiguous()
    ad, cd, bd, ccd, bsd, ysd, evd = f(a), f(c), f(bb.reshape(B * N, G * dv)), f(cc.reshape(B * N, G * dv)), f(bs), f(ys), f(ev)
    ws = torch.empty(lib.sa_favor_scan_workspace_bytes(B, N, G, LDF, dv) // 4, device="cuda")
    outs = []
    for w in (ws, None):
        ya = f(y0.reshape(B * N, G * dv))
        _ffi.check(lib.sa_favor_scan_a(_ffi.ptr(ad), _ffi.ptr(cd), _ffi.ptr(bd), G * dv, 0, _ffi.ptr(bsd), _ffi.ptr(ya), G * dv, 0, _ffi.ptr(ysd), B, N, G, LDF, dv,
                                       reverse, 1, _ffi.ptr(w), st))
        yb = torch.zeros(B, N, G, LDF, device="cuda")
        _ffi.check(lib.sa_favor_scan_b(_ffi.ptr(ad), _ffi.ptr(bd), G * dv, 0, _ffi.ptr(bsd), _ffi.ptr(ccd), G * dv, 0, _ffi.ptr(ysd), _ffi.ptr(yb), _ffi.ptr(bsd),
                                       _ffi.ptr(evd), 0.25, B, N, G, LDF, dv, reverse, _ffi.ptr(w), st))
        outs.append((ya, yb))
        assert _rel(ya.view(B, N, G, dv).cpu().double(), refA) < 1e-4
        assert _rel(yb[..., :m].cpu().double(), refB[..., :m]) < 1e-4
    assert _rel(outs[0][0], outs[1][0]) < 1e-4 and _rel(outs[0][1][..., :m], outs[1][1][..., :m]) < 1e-4
    # ---- the fused running sums (chunked MFMA path): cumulative extra terms and the normaliser without cumsum / den passes
    trif = tri
    cum_a = torch.einsum("ij,bjgm->bigm", trif, a)                                   # sum_{j <= i (scan order)} a_j
    cum_aw = torch.einsum("ij,bjgm->bigm", trif, a * bs[..., None])                  # ... weighted by ex_scale_j
    base = torch.einsum("bjgm,bjgd,ij,bigd->bigm", a, bsc, trif, cc * ys[..., None])
    y1 = torch.zeros(B, N, G, LDF, device="cuda")
    _ffi.check(lib.sa_favor_scan_b_cum(_ffi.ptr(ad), _ffi.ptr(bd), G * dv, 0, _ffi.ptr(bsd), _ffi.ptr(ccd), G * dv, 0, _ffi.ptr(ysd), _ffi.ptr(y1), _ffi.ptr(bsd), 1, 0.25,
                                       B, N, G, LDF, dv, reverse, _ffi.ptr(ws), 0, st))
    assert _rel(y1[..., :m].cpu().double(), (base + bs[..., None] * (cum_a + 0.25))[..., :m]) < 1e-4
    y2 = torch.zeros(B, N, G, LDF, device="cuda")
    _ffi.check(lib.sa_favor_scan_b_cum(_ffi.ptr(ad), _ffi.ptr(bd), G * dv, 0, _ffi.ptr(bsd), _ffi.ptr(ccd), G * dv, 0, _ffi.ptr(ysd), _ffi.ptr(y2), _ffi.ptr(bsd), 2, 0.0,
                                       B, N, G, LDF, dv, reverse, _ffi.ptr(ws), 0, st))
    # the same states serve a plain scan A on the same (a, b, b_scale, reverse): no state / prefix passes, extra-column layout
    ya2 = f(y0.reshape(B * N, G * dv))
    _ffi.check(lib.sa_favor_scan_a_state(_ffi.ptr(ad), _ffi.ptr(cd), _ffi.ptr(bd), G * dv, 0, _ffi.ptr(bsd), _ffi.ptr(ya2), G * dv, 0, _ffi.ptr(ysd), B, N, G, LDF, dv,
                                         reverse, 1, _ffi.ptr(ws), 3, st))
    assert _rel(ya2.view(B, N, G, dv).cpu().double(), refA) < 1e-4
    assert _rel(y2[..., :m].cpu().double(), (base + cum_aw)[..., :m]) < 1e-4
    if not reverse:
        num = torch.einsum("bigm,bjgm,ij,bjgd->bigd", c, a, trif, bb)
        den = (c * (cum_a + 1e-6)).sum(-1, keepdim=True)
        yn = torch.zeros(B * N, G * dv, device="cuda")
        invn = torch.zeros(B * N * G, device="cuda")
        _ffi.check(lib.sa_favor_scan_a_norm(_ffi.ptr(ad), _ffi.ptr(cd), _ffi.ptr(bd), G * dv, 0, _ffi.ptr(yn), G * dv, 0, _ffi.ptr(invn), 1e-6, B, N, G, LDF, dv,
                                            _ffi.ptr(ws), 0, st))
        assert _rel(yn.view(B, N, G, dv).cpu().double(), num / den) < 1e-4
        assert _rel(invn.view(B, N, G, 1).cpu().double(), 1.0 / den) < 1e-4


@pytest.fixture(params=["split-bf16", "exact-fp32"])
def la_path(request):
    """Both arithmetic paths of csrc/local_attn.hip (SA_DBG_LOCAL_ATTN_EXACT of the library's debug word)."""
    from synthanatomy_amd import debug
    with debug.override(local_attn_exact=request.param == "exact-fp32"):
        yield request.param


@pytest.mark.parametrize("N,W", [(23, 5), (40, 8), (17, 32), (100, 420), (333, 64), (1400, 420), (1000, 420), (841, 420)])
def test_local_attention_kernel_against_dense_band(N, W, la_path):
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(N)
    B, L, dh = 2, 3, 64
    q, k, v = (torch.randn(B, L, N, dh) for _ in range(3))
    ref = P.local_attention(q, k, v, W, rotary=False)
    pack = lambda t: t.permute(0, 2, 1, 3).reshape(B * N, L * dh).contiguous().cuda()
    qd, kd, vd = pack(q), pack(k), pack(v)
    o = torch.empty_like(qd)
    lse = torch.empty(B * N * L, device="cuda")
    _ffi.check(lib.sa_local_attn_fwd(_ffi.ptr(qd), L * dh, 0, _ffi.ptr(kd), L * dh, 0, _ffi.ptr(vd), L * dh, 0, _ffi.ptr(o), L * dh, 0, _ffi.ptr(lse), B, N, L, W, dh, None, st))
    got = o.view(B, N, L, dh).permute(0, 2, 1, 3)
    assert _rel(got, ref) < 1e-4


def test_projection_redraw_kernel_is_orthogonal_and_rank_consistent():
    from synthanatomy_amd.networks.transformers.performer import FastAttention
    fa = FastAttention(64).cuda()
    g1 = torch.Generator(device="cuda").manual_seed(123)
    fa.redraw_projection_matrix("cuda", g1)
    a = fa.projection_matrix.clone()
    g2 = torch.Generator(device="cuda").manual_seed(123)
    fa.redraw_projection_matrix("cuda", g2)
    assert torch.equal(a, fa.projection_matrix)  # same seed on every rank -> same matrix, no broadcast needed
    blk = a[:64] / a[:64].norm(dim=1, keepdim=True)
    assert float((blk @ blk.t() - torch.eye(64, device="cuda")).abs().max()) < 1e-4
    assert a.shape == (266, 64) and 5.0 < float(a.norm(dim=1).mean()) < 11.0  # chi(64) row norms ~ 8


def test_sample_post_processing_and_greedy_path():
    """sample(): N forwards over the growing prefix through the HIP path, then revert ordering + reshape (transformer.py:58-101)."""
    shape = (2, 2, 3)
    cfg = P.PerformerConfig(num_tokens=17, max_seq_len=12, dim=32, depth=1, heads=2, dim_head=64, local_attn_heads=1, local_window_size=4, spatial_shape=shape)
    st = P.init_state(cfg, seed=3)
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    o = Ordering("s_curve", 3, (1,) + shape, (False,) * 3, (), ())
    net = Performer(num_tokens=17, max_seq_len=12, dim=32, depth=1, heads=2, ordering=o, dim_head=64, local_attn_heads=1, local_window_size=4, use_rezero=True,
                    spatial_position_emb="absolute", spatial_shape=shape, feature_redraw_interval=None)
    net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=False)
    net = net.cuda()
    prefix = torch.full((2, 1), 16, dtype=torch.long, device="cuda")
    got = net.sample(prefix, sample=False, stateful=False)
    assert got.shape == (2, *shape) and int(got.min()) >= 0 and int(got.max()) <= 16
    # oracle replay of the same greedy chain
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    x = torch.full((2, 1), 16, dtype=torch.long)
    for _ in range(12):
        nxt = P.forward(st, cfg, x, seqs)[:, -1].argmax(-1, keepdim=True)
        x = torch.cat((x, nxt), 1)
    ref = x[:, 1:][:, o.get_revert_sequence_ordering()].reshape(2, *shape)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("rezero,shape,window,local,use_graph", [(True, (2, 3, 4), 5, 1, True), (False, (3, 2, 5), 7, 2, False), (True, (2, 2, 6), 64, 0, True),
                                                                 (True, (2, 3, 5), 4, 3, False)])
def test_stateful_sampler_equals_the_quadratic_loop(rezero, shape, window, local, use_graph):
    """O(N) decoding (FAVOR+ running sums with a rescalable key stabiliser, local key/value caches; one HIP graph per token) against the
    reference-faithful loop that re-runs the network over the growing prefix, and against the CPU oracle's greedy chain."""
    n = int(np.prod(shape))
    heads = 4
    cfg = P.PerformerConfig(num_tokens=19, max_seq_len=n, dim=32, depth=2, heads=heads, dim_head=64, local_attn_heads=local, local_window_size=window,
                            spatial_shape=shape, use_rezero=rezero)
    st = P.init_state(cfg, seed=7)
    for k in st:   # ReZero gates start at 1e-3: make the attention matter
        if k.endswith(".g"):
            st[k] = torch.full_like(st[k], 0.7)
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    o = Ordering("raster_scan", 3, (1,) + shape, (False,) * 3, (), ())
    net = Performer(num_tokens=19, max_seq_len=n, dim=32, depth=2, heads=heads, ordering=o, dim_head=64, local_attn_heads=local, local_window_size=window,
                    use_rezero=rezero, spatial_position_emb="absolute", spatial_shape=shape, feature_redraw_interval=None)
    net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=False)
    net = net.cuda()
    from synthanatomy_amd import debug
    prefix = torch.full((3, 1), 18, dtype=torch.long, device="cuda")
    quad = net.sample(prefix, sample=False, stateful=False)
    fast = net.sample(prefix, sample=False, stateful=True, use_graph=use_graph)
    assert fast.shape == (3, *shape)
    assert torch.equal(fast, quad)
    again = net.sample(prefix, sample=False)          # default = stateful; a second run must not see the first one's state
    assert torch.equal(again, quad)
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    x = torch.full((3, 1), 18, dtype=torch.long)
    for _ in range(n):
        x = torch.cat((x, P.forward(st, cfg, x, seqs)[:, -1].argmax(-1, keepdim=True)), 1)
    ref = x[:, 1:][:, o.get_revert_sequence_ordering()].reshape(3, *shape)
    assert torch.equal(fast.cpu(), ref)
    # stochastic path runs and stays in range
    smp = net.sample(prefix, sample=True, top_k=5, temperature=0.9)       # top-k inside sa_sample_step
    assert int(smp.min()) >= 0 and int(smp.max()) <= 18
    with debug.override(no_sample_step=True):
        smp = net.sample(prefix, sample=True, top_k=5, temperature=0.9)   # top-k: the torch expressions
    assert int(smp.min()) >= 0 and int(smp.max()) <= 18
    smp = net.sample(prefix, sample=True, temperature=0.9)                # no top-k: sa_sample_step
    assert int(smp.min()) >= 0 and int(smp.max()) <= 18
    with debug.override(no_sample_step=True):                            # the same greedy chain with the decision made by torch ops
        assert torch.equal(net.sample(prefix, sample=False, use_graph=use_graph), quad)


@pytest.mark.parametrize("per_row", [False, True])
@pytest.mark.parametrize("B,V,P_,temperature,top_k", [(6, 2049, 1, 1.0, 0), (3, 19, 2, 0.7, 0), (17, 1000, 1, 1.3, 0), (6, 2049, 1, 0.9, 40), (3, 19, 1, 1.0, 5),
                                                     (5, 16384, 1, 1.0, 1), (4, 300, 1, 1.1, 299)])
def test_sample_step_kernel_matches_the_torch_decision(B, V, P_, temperature, top_k, per_row):
    """sa_sample_step (one launch per token: temperature, top-k cut, softmax, inverse-CDF draw with the caller's uniforms or arg-max, sequence update, next token,
    pos += 1) against the torch expressions of the stateful sampler it replaces (transformer.py:11-17,42-54), position by position; one block walking the rows with
    a fresh vector of uniforms per step, and one block per row (ticket word) reading a table of uniforms by position.  Logits are rounded to a coarse grid when
    top-k is on, so that rows hold TIES with the k-th value (the reference keeps them)."""
    from synthanatomy_amd import _ffi
    lib = _ffi.lib()
    g = torch.Generator().manual_seed(B + V)
    total = 9
    mism = 0
    for do_sample in (0, 1):
        seq = torch.randint(0, V, (B, total), generator=g).cuda()
        seq_t = seq.clone()
        posbuf = torch.zeros(2, dtype=torch.int32, device="cuda")
        pos, ticket = posbuf[:1], posbuf[1:]
        tok = torch.zeros(B, dtype=torch.int64, device="cuda")
        table = torch.rand(total, B, generator=g).cuda()
        for step in range(total - 1):
            logits = torch.randn(B, V, generator=g) * 3
            if top_k:
                logits = (logits * 8).round() / 8
                logits[0, :3] = 0.0
                logits[0, 1] = -0.0
            logits = logits.cuda()
            u = table[step] if per_row else torch.rand(B, generator=g).cuda()
            _ffi.check(lib.sa_sample_step(_ffi.ptr(logits), B, V, float(temperature), _ffi.ptr(table if per_row else u), B if per_row else 0, do_sample, top_k,
                                          _ffi.ptr(seq), total, P_, _ffi.ptr(pos), _ffi.ptr(ticket) if per_row else None, _ffi.ptr(tok), _ffi.stream()), "sa_sample_step")
            scaled = logits / temperature
            if top_k:
                kth = torch.topk(scaled, top_k, dim=-1).values[:, -1:]
                scaled = scaled.masked_fill(scaled < kth, float("-inf"))
            probs = torch.softmax(scaled, dim=-1)
            if do_sample:
                cdf = probs.cumsum(-1)
                ix = (cdf < u[:, None] * cdf[:, -1:]).sum(-1).clamp(max=V - 1)
            else:
                ix = probs.argmax(-1)
            if step + 1 >= P_:
                seq_t[:, step + 1] = ix
            torch.cuda.synchronize()
            assert int(pos) == step + 1 and int(ticket) == 0
            assert torch.equal(tok, seq[:, step + 1])
            if do_sample:      # a cdf value within rounding of the target may fall on either side: count, then continue from the kernel's choice
                if step + 1 >= P_:   # ... but never on a token the cut removed
                    assert bool((probs.gather(1, seq[:, step + 1:step + 2]) > 0).all())
                mism += int((seq[:, step + 1] != seq_t[:, step + 1]).sum())
                seq_t[:, step + 1] = seq[:, step + 1]
            else:
                assert torch.equal(seq, seq_t), step
    assert mism <= 1, mism


@pytest.mark.parametrize("V", [256, 2048, 16384])
def test_sample_step_kernel_never_emits_an_unclaimed_token(V):
    """The draw is the first token with mass whose cdf reaches the target (lowest index over all threads), not 'the thread whose interval contains the target':
    with per-thread partial sums associated differently, neighbouring intervals can leave one-ulp gaps, and a target in a gap used to fall through to token V - 1
    (2e-6 to 6e-6 per draw).  Here
    token V - 1 carries zero probability (logit -inf) and every other token a comparable one: across 2^18 draws -- incl. u = 0 and u = 1 - 2^-24 -- the kernel
    must never return it, and an all-NaN row must stay in range on both the arg-max and the sampling path."""
    from synthanatomy_amd import _ffi
    lib = _ffi.lib()
    g = torch.Generator().manual_seed(V)
    B = 64
    logits = torch.randn(B, V, generator=g)
    logits[:, V - 1] = float("-inf")
    logits = logits.cuda()
    steps = 4096
    u = torch.rand(steps, B, generator=g)
    u[0] = 0.0
    u[1] = 1.0 - 2.0 ** -24
    u = u.cuda()
    seq = torch.zeros(B, steps + 1, dtype=torch.int64, device="cuda")
    posbuf = torch.zeros(2, dtype=torch.int32, device="cuda")
    tok = torch.zeros(B, dtype=torch.int64, device="cuda")
    for _ in range(steps):
        _ffi.check(lib.sa_sample_step(_ffi.ptr(logits), B, V, 1.0, _ffi.ptr(u), B, 1, 0, _ffi.ptr(seq), steps + 1, 1, _ffi.ptr(posbuf[:1]), _ffi.ptr(posbuf[1:]),
                                      _ffi.ptr(tok), _ffi.stream()), "sa_sample_step")
    torch.cuda.synchronize()
    assert int(posbuf[0]) == steps
    drawn = seq[:, 1:]
    assert int(drawn.max()) < V - 1 and int(drawn.min()) >= 0
    assert int((drawn[:, 0] != 0).sum()) <= 0        # u = 0: nothing lies below the target
    # the empirical distribution follows the softmax (a coarse check that the count rule is the inverse CDF)
    p = torch.softmax(logits[0].double(), -1).cpu()
    f = torch.bincount(drawn[0].cpu(), minlength=V).double() / steps
    assert float((f - p).abs().sum()) < 2.5 * (V / steps) ** 0.5 + 0.05
    nan_logits = torch.full((2, V), float("nan"), device="cuda")
    for do_sample in (0, 1):
        posbuf.zero_()
        _ffi.check(lib.sa_sample_step(_ffi.ptr(nan_logits), 2, V, 1.0, _ffi.ptr(u), B, do_sample, 0, _ffi.ptr(seq), steps + 1, 1, _ffi.ptr(posbuf[:1]),
                                      _ffi.ptr(posbuf[1:]), _ffi.ptr(tok), _ffi.stream()), "sa_sample_step")
        torch.cuda.synchronize()
        assert 0 <= int(seq[0, 1]) < V and 0 <= int(seq[1, 1]) < V


@pytest.mark.parametrize("N,W", [(23, 5), (100, 420), (150, 64), (200, 70), (321, 128), (1400, 420), (1000, 420)])
def test_local_attention_backward_kernels_against_autograd(N, W, la_path):
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(N + W)
    B, L, dh = 2, 2, 64
    q, k, v = (torch.randn(B, L, N, dh, requires_grad=True) for _ in range(3))
    ref = P.local_attention(q, k, v, W, rotary=False)
    go = torch.randn_like(ref)
    ref.backward(go)
    pack = lambda t: t.detach().permute(0, 2, 1, 3).reshape(B * N, L * dh).contiguous().cuda()
    qd, kd, vd, god = pack(q), pack(k), pack(v), pack(go)
    o = torch.empty_like(qd)
    o_lp = torch.empty_like(qd, dtype=torch.bfloat16)     # bf16 mirrors (the operands of the next dense layers) ride along with the fp32 outputs
    lse = torch.empty(B * N * L, device="cuda")
    _ffi.check(lib.sa_local_attn_fwd(_ffi.ptr(qd), L * dh, 0, _ffi.ptr(kd), L * dh, 0, _ffi.ptr(vd), L * dh, 0, _ffi.ptr(o), L * dh, 0, _ffi.ptr(lse), B, N, L, W, dh, _ffi.ptr(o_lp), st))
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    dv_lp = torch.empty_like(vd, dtype=torch.bfloat16)
    Db = torch.empty(B * N * L, device="cuda")
    _ffi.check(lib.sa_local_attn_bwd(_ffi.ptr(qd), L * dh, 0, _ffi.ptr(kd), L * dh, 0, _ffi.ptr(vd), L * dh, 0, _ffi.ptr(o), _ffi.ptr(god), L * dh, 0, _ffi.ptr(lse),
                                     _ffi.ptr(dq), _ffi.ptr(dk), _ffi.ptr(dv), _ffi.ptr(Db), B, N, L, W, dh, _ffi.ptr(dv_lp), st))
    unpack = lambda t: t.view(B, N, L, dh).permute(0, 2, 1, 3)
    assert torch.equal(o_lp, o.to(torch.bfloat16)) and torch.equal(dv_lp, dv.to(torch.bfloat16))
    assert _rel(unpack(o), ref) < 1e-4
    assert _rel(unpack(dq), q.grad) < 1e-4 and _rel(unpack(dk), k.grad) < 1e-4 and _rel(unpack(dv), v.grad) < 1e-4


@pytest.mark.parametrize("B,cin,segs,act,resid,rnd", [(1, 32, (48,), 0, False, 0), (6, 512, (1024, 1024, 1024), 0, False, 1), (17, 2048, (512,), 0, True, 1),
                                                       (32, 512, (2048,), 1, False, 1), (5, 128, (40, 24), 1, True, 0)])
def test_small_batch_dense_kernel(B, cin, segs, act, resid, rnd):
    """sa_gemv_rows (the dense layers of the decode step on the MFMA, K split over 8 waves) against torch: concatenated weight tensors,
    bias, GELU, gated residual, bf16 operand / output rounding as on the training path."""
    import ctypes
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(B + cin)
    x = torch.randn(B, cin)
    ws = [torch.randn(o, cin) * cin ** -0.5 for o in segs]
    bs = [torch.randn(o) * 0.1 for o in segs]
    res = torch.randn(B, sum(segs)) if resid else None
    gate = torch.tensor(0.37)
    r = (lambda t: t.to(torch.bfloat16).float()) if rnd else (lambda t: t)
    ref = torch.cat([r(x).double() @ r(w).double().t() + b.double() for w, b in zip(ws, bs)], 1).float()
    if rnd:
        ref = r(ref)
    if act:
        ref = torch.nn.functional.gelu(ref)
        if rnd:
            ref = r(ref)
    if resid:
        ref = res + gate * ref
    xd, wd, bd = x.cuda(), [w.cuda() for w in ws], [b.cuda() for b in bs]
    resd, gd = (res.cuda() if resid else None), gate.cuda()
    y = torch.full((B, sum(segs)), float("nan"), device="cuda")
    n = len(segs)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in wd])
    bp = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bd])
    so = (ctypes.c_int32 * n)(*segs)
    _ffi.check(lib.sa_gemv_rows(_ffi.ptr(xd), cin, cin, B, n, wp, bp, so, _ffi.ptr(y), sum(segs), act, _ffi.ptr(resd), sum(segs) if resid else 0,
                                _ffi.ptr(gd) if resid else None, rnd, rnd, rnd, st))
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert _rel(y.cpu(), ref) < (1.5e-2 if rnd else 1e-5)
    if rnd:   # round_w = 2: the weight tensors already hold bf16 copies (what the sampler streams) -> the same bits as rounding the fp32 weights in the kernel
        w16 = [w.to(torch.bfloat16).contiguous() for w in wd]
        wp16 = (ctypes.c_void_p * n)(*[w.data_ptr() for w in w16])
        y2 = torch.full((B, sum(segs)), float("nan"), device="cuda")
        _ffi.check(lib.sa_gemv_rows(_ffi.ptr(xd), cin, cin, B, n, wp16, bp, so, _ffi.ptr(y2), sum(segs), act, _ffi.ptr(resd), sum(segs) if resid else 0,
                                    _ffi.ptr(gd) if resid else None, rnd, 2, rnd, st))
        torch.cuda.synchronize()
        assert torch.equal(y2, y)


@pytest.mark.parametrize("variant", ["fixed", "prepending", "bos_replacement", "fixed_pos"])
def test_embedding_variants_match_oracle(variant):
    """`spatial_position_emb="fixed"` (sinusoids of the coordinate value, reference performer.py:43-66), `fixed_position_emb=True` (the sinusoidal
    positional buffer instead of the learned table, performer.py:138-140; with BOS replacement so that row 0 of the buffer is used too) and the two conditioning types
    (BOS replacement :252-261, prepending :262-264 with the conditioning positions cut off after the norm :279-281): logits and gradients."""
    from synthanatomy_amd.losses.transformer import CELoss
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    shape, n, ncond = (2, 3, 4), 24, (5, 7)
    prep = variant == "prepending"
    cfg = P.PerformerConfig(num_tokens=33, max_seq_len=n + (len(ncond) if prep else 0), dim=32, depth=2, heads=4, dim_head=64, local_attn_heads=2,
                            local_window_size=6, spatial_shape=shape)
    st = P.init_state(cfg, seed=11, spatial_index_len=n - 1)
    g = torch.Generator().manual_seed(3)
    for i, c in enumerate(ncond):
        st[f"conditioning_emb.{i}.weight"] = torch.randn(c, 32, generator=g)
    for k in st:
        if k.endswith(".g"):
            st[k] = torch.tensor(0.4)
    sp = "fixed" if variant == "fixed" else "absolute"
    ctype = {"fixed": "none", "prepending": "prepending", "bos_replacement": "bos_replacement", "fixed_pos": "bos_replacement"}[variant]
    if variant == "fixed_pos":
        del st["pos_emb.emb.weight"]
        st["pos_emb.emb"] = P.fixed_position_table(32, n)
    o = Ordering("raster_scan", 3, (1,) + shape, (False,) * 3, (), ())
    net = Performer(num_tokens=33, max_seq_len=n, dim=32, depth=2, heads=4, ordering=o, dim_head=64, local_attn_heads=2, local_window_size=6, use_rezero=True,
                    spatial_position_emb=sp, spatial_shape=shape, feature_redraw_interval=None, compute_dtype=torch.float32,
                    conditioning_num_tokens=ncond if variant != "fixed" else None, conditioning_type=ctype, fixed_position_emb=variant == "fixed_pos")
    assert net.max_seq_len == n + (2 if prep else 0)
    load = {k: v.clone() for k, v in st.items() if not (variant == "fixed" and "spatial_position_emb" in k) and not (variant == "fixed" and "conditioning_emb" in k)}
    missing, unexpected = net.load_state_dict(load, strict=False)
    assert not unexpected, unexpected
    net = net.cuda().train()
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    if variant == "fixed_pos":
        assert "pos_emb.emb" in net.state_dict() and "pos_emb.emb.weight" not in net.state_dict()
        assert torch.allclose(net.pos_emb.emb.cpu(), st["pos_emb.emb"], atol=1e-6)
    if variant == "fixed":     # the product's buffers are the oracle's tables
        for a in range(3):
            assert torch.allclose(net.spatial_position_emb[a].emb.cpu(), P.fixed_spatial_table(32, seqs[a]), atol=1e-6)
    tok = torch.randint(0, 33, (2, n), generator=g)
    tgt = torch.randint(0, 32, (2, n), generator=g)
    conds = [torch.randint(0, c, (2, 1), generator=g) for c in ncond] if variant != "fixed" else None
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "projection_matrix" not in k}
    stt = dict(st)
    stt.update(leaf)
    ref = P.forward(stt, cfg, tok, seqs, conds, ctype, sp)
    P.ce_loss(ref, tgt).backward()
    out = net(tok.cuda(), conditionings=[c.cuda() for c in conds] if conds else None)
    assert out.shape == ref.shape == (2, n, 33) and _rel(out, ref) < REL
    CELoss()(out.transpose(1, 2), tgt.cuda()).backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    checked = 0
    for k, p in leaf.items():
        if k in params and p.grad is not None and params[k].grad is not None and float(p.grad.abs().max()) > 0:
            assert _rel(params[k].grad, p.grad) < 3e-3, k
            checked += 1
    assert checked > 20
    if variant != "fixed":
        assert float(params["conditioning_emb.0.weight"].grad.abs().max()) > 0


@pytest.mark.parametrize("fixed_pos", [False, True])
def test_stateful_sampler_with_bos_replacement_conditioning(fixed_pos):
    """(`fixed_pos`: the sinusoidal positional buffer of `fixed_position_emb=True` instead of the learned table.)  O(N) decoding with `conditioning_type="bos_replacement"` (reference performer.py:252-261 through transformer.py:58-101): position 0 carries the
    summed conditioning embeddings; token for token equal to the reference-faithful loop and to the CPU oracle's greedy chain."""
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    from synthanatomy_amd.networks.transformers.performer import Performer
    shape, n, ncond = (2, 3, 4), 24, (5, 7)
    cfg = P.PerformerConfig(num_tokens=19, max_seq_len=n, dim=32, depth=2, heads=4, dim_head=64, local_attn_heads=2, local_window_size=6, spatial_shape=shape)
    st = P.init_state(cfg, seed=13)
    g = torch.Generator().manual_seed(5)
    for i, c in enumerate(ncond):
        st[f"conditioning_emb.{i}.weight"] = torch.randn(c, 32, generator=g)
    for k in st:
        if k.endswith(".g"):
            st[k] = torch.full_like(st[k], 0.7)
    if fixed_pos:
        del st["pos_emb.emb.weight"]
        st["pos_emb.emb"] = P.fixed_position_table(32, n)
    o = Ordering("raster_scan", 3, (1,) + shape, (False,) * 3, (), ())
    net = Performer(num_tokens=19, max_seq_len=n, dim=32, depth=2, heads=4, ordering=o, dim_head=64, local_attn_heads=2, local_window_size=6, use_rezero=True,
                    spatial_position_emb="absolute", spatial_shape=shape, feature_redraw_interval=None, conditioning_num_tokens=ncond,
                    conditioning_type="bos_replacement", fixed_position_emb=fixed_pos)
    net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=False)
    net = net.cuda()
    B = 3
    prefix = torch.full((B, 1), 18, dtype=torch.long, device="cuda")
    conds = [torch.randint(0, c, (B, 1), generator=g) for c in ncond]
    cd = [c.cuda() for c in conds]
    quad = net.sample(prefix, conditioning=cd, sample=False, stateful=False)
    fast = net.sample(prefix, conditioning=cd, sample=False)                 # default: stateful for BOS replacement
    fast_eager = net.sample(prefix, conditioning=cd, sample=False, stateful=True, use_graph=False)
    assert torch.equal(fast, quad) and torch.equal(fast_eager, quad)
    other = net.sample(prefix, conditioning=[(c + 1) % k for c, k in zip(cd, ncond)], sample=False)
    assert not torch.equal(other, fast)                                      # the conditioning matters
    seqs = P.spatial_index_sequences(shape, o.get_sequence_ordering())
    x = torch.full((B, 1), 18, dtype=torch.long)
    for _ in range(n):
        x = torch.cat((x, P.forward(st, cfg, x, seqs, conds, "bos_replacement")[:, -1].argmax(-1, keepdim=True)), 1)
    ref = x[:, 1:][:, o.get_revert_sequence_ordering()].reshape(B, *shape)
    assert torch.equal(fast.cpu(), ref)


@pytest.mark.parametrize("B,G,L,N,W", [(2, 2, 2, 60, 7), (3, 1, 3, 1000, 420), (1, 3, 1, 300, 64)])
def test_two_launch_attention_step_equals_the_separate_steps(B, G, L, N, W):
    """sa_attn_step ([projections | local heads over four key segments], [FAVOR+ update | combine]) against sa_favor_step + sa_local_attn_step on the same
    random q | k | v rows, position by position over N steps (window fills, window boundaries, the second window sliding): attention rows <= 1e-5, key / value
    caches and the FAVOR+ state identical."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(B * 100 + W)
    dh, m, LDF = 64, 266, 272
    H = G + L
    inner = H * dh
    proj = (torch.randn(m, dh) * dh ** -0.25).cuda()
    fr = torch.einsum("i,j->ij", torch.arange(N, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, dh, 2).float() / dh)))
    fr = torch.cat((fr, fr), -1).cuda()
    cosb, sinb = fr.cos().contiguous(), fr.sin().contiguous()

    def state():
        s = dict(smax=torch.full((2,), float("-inf"), device="cuda"), kmax=torch.full((2,), -2139095041, dtype=torch.int32, device="cuda"),
                 dd=torch.zeros(2, B * G, LDF, device="cuda"), E=torch.zeros(B * G, LDF * dh, device="cuda"), Ez=torch.zeros(B * G, LDF, device="cuda"),
                 V1=torch.zeros(B * G, dh, device="cuda"), kc=torch.zeros(B, L, N, dh, device="cuda"), vc=torch.zeros(B, L, N, dh, device="cuda"),
                 part=torch.zeros(B * L * 4 * 66, device="cuda"))
        return s
    sa_, sb_ = state(), state()
    pos = torch.zeros(1, dtype=torch.int32, device="cuda")
    worst = 0.0
    qkv_all = torch.randn(N, B, 3 * inner, device="cuda")
    for t in range(N):
        pos.fill_(t)
        qkv = qkv_all[t]
        a1 = torch.zeros(B, inner, device="cuda")
        a2 = torch.zeros(B, inner, device="cuda")
        _ffi.check(lib.sa_attn_step(_ffi.ptr(qkv), 3 * inner, inner, _ffi.ptr(proj), B, G, L, dh, m, LDF, _ffi.ptr(sa_["smax"]), _ffi.ptr(sa_["kmax"]), _ffi.ptr(sa_["dd"]),
                                    _ffi.ptr(sa_["E"]), _ffi.ptr(sa_["Ez"]), _ffi.ptr(sa_["V1"]), _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(sa_["kc"]), _ffi.ptr(sa_["vc"]), N, W,
                                    _ffi.ptr(sa_["part"]), _ffi.ptr(pos), _ffi.ptr(a1), inner, st))
        _ffi.check(lib.sa_favor_step(_ffi.ptr(qkv), 3 * inner, 0, _ffi.ptr(qkv), 3 * inner, inner, _ffi.ptr(qkv), 3 * inner, 2 * inner, _ffi.ptr(proj), B, G, dh, m, LDF,
                                     _ffi.ptr(sb_["smax"]), _ffi.ptr(sb_["kmax"]), _ffi.ptr(sb_["dd"]), _ffi.ptr(sb_["E"]), _ffi.ptr(sb_["Ez"]), _ffi.ptr(sb_["V1"]),
                                     _ffi.ptr(pos), _ffi.ptr(a2), inner, 0, st))
        _ffi.check(lib.sa_local_attn_step(_ffi.ptr(qkv), 3 * inner, G * dh, _ffi.ptr(qkv), 3 * inner, inner + G * dh, _ffi.ptr(qkv), 3 * inner, 2 * inner + G * dh,
                                          _ffi.ptr(cosb), _ffi.ptr(sinb), _ffi.ptr(sb_["kc"]), _ffi.ptr(sb_["vc"]), _ffi.ptr(pos), B, N, L, W, dh, _ffi.ptr(a2),
                                          inner, G * dh, st))
        if t % 37 == 0 or t in (W - 1, W, 2 * W - 1, 2 * W, N - 1):
            assert torch.equal(a1[:, :G * dh], a2[:, :G * dh])                      # global heads: the same code on the same state
            worst = max(worst, _rel(a1[:, G * dh:], a2[:, G * dh:]))
    torch.cuda.synchronize()
    assert worst < 1e-5, worst
    for k in ("E", "Ez", "V1", "kc", "vc", "smax"):
        assert torch.equal(sa_[k], sb_[k]), k
