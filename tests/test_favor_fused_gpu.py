"""GPU: FAVOR+ with the feature maps recomputed on chip (csrc/favor_fused.hip) against the fp64 statement of performer_pytorch's softmax_kernel +
causal_linear_attention (oracle/performer_ref.py) and autograd through it, and against the unfused kernel chain it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import performer_ref as P  # noqa: E402


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def _fro(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def _reference(q, k, v, proj, dattn):
    """fp64: q, k, v [B, G, N, 64] -> attention output and the gradients of <out, dattn>."""
    q, k, v = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    qp = P.softmax_kernel(q, proj.double(), True)
    kp = P.softmax_kernel(k, proj.double(), False)
    out = P.causal_linear_attention(qp, kp, v)
    (out * dattn.double()).sum().backward()
    return out.detach(), q.grad, k.grad, v.grad


def _fused(q, k, v, proj, dattn, local_cols=64, ps=None, force_flag=None):
    """The C ABI on head blocks of wider rows (as q | k | v sit in the fused qkv matrix): returns out, dq, dk, dv [B, G, N, 64]."""
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    B, G, N, dh = q.shape
    m = proj.shape[0]
    inner = G * dh + local_cols                    # the local heads' columns follow the global ones
    stride = 3 * inner
    R = B * N
    pack = lambda t: t.permute(0, 2, 1, 3).reshape(R, G * dh)
    qkv = torch.randn(R, stride)
    qkv[:, :G * dh], qkv[:, inner:inner + G * dh], qkv[:, 2 * inner:2 * inner + G * dh] = pack(q), pack(k), pack(v)
    qkv = qkv.cuda()
    qd, kd, vd = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
    ps = ((proj * dh ** -0.25) if ps is None else ps).contiguous().cuda()
    tiles = torch.empty(5 * 16384, dtype=torch.uint8, device="cuda")
    _ffi.check(lib.sa_favor_fused_proj_tiles(_ffi.ptr(ps), m, _ffi.ptr(tiles), st))
    flag = int(tiles[-16:].view(torch.int32)[0])      # 0: every lo half of the split matrix is zero (csrc/favor_fused.hip: PT_FLAG_OFF)
    if force_flag is not None:
        tiles[-16:].view(torch.int32)[0] = force_flag
    offq = torch.empty(R * G, device="cuda")
    offk = torch.empty(R * G, device="cuda")
    amq = torch.empty(R * G, dtype=torch.int32, device="cuda")
    gws = torch.zeros(1, dtype=torch.int64, device="cuda")
    _ffi.check(lib.sa_favor_fused_prepass(_ffi.ptr(qd), _ffi.ptr(kd), stride, G, _ffi.ptr(tiles), _ffi.ptr(offq), _ffi.ptr(amq), _ffi.ptr(offk), _ffi.ptr(gws), R * G, m, dh, st))
    nst = lib.sa_favor_fused_state_bytes(B, N, G, m) // 4
    state = torch.empty(nst, device="cuda")
    state2 = torch.empty(nst, device="cuda")
    attn = torch.full((R, inner), 7.0, device="cuda")
    attn_lp = torch.full((R, inner), 7.0, device="cuda", dtype=torch.bfloat16)     # bf16 mirrors of the fp32 outputs (operands of the next dense layers)
    inv = torch.empty(R * G, device="cuda")
    _ffi.check(lib.sa_favor_fused_fwd(_ffi.ptr(qd), _ffi.ptr(kd), _ffi.ptr(vd), stride, _ffi.ptr(tiles), _ffi.ptr(ps), _ffi.ptr(offq), _ffi.ptr(offk), _ffi.ptr(gws),
                                      _ffi.ptr(attn), inner, _ffi.ptr(inv), 1e-6, B, N, G, m, _ffi.ptr(state), _ffi.ptr(attn_lp), None, st))
    da = torch.zeros(R, inner)
    da[:, :G * dh] = pack(dattn)
    da = da.cuda()
    dqkv = torch.full((R, stride), 3.0, device="cuda")
    dq, dk, dv = dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:]
    dqkv_lp = torch.full((R, stride), 3.0, device="cuda", dtype=torch.bfloat16)
    dq_lp, dk_lp, dv_lp = dqkv_lp[:, :inner], dqkv_lp[:, inner:2 * inner], dqkv_lp[:, 2 * inner:]
    dden = torch.empty(R * G, device="cuda")
    tsum = torch.zeros(B * G * ((N + 63) // 64), device="cuda")
    _ffi.check(lib.sa_favor_fused_bwd(_ffi.ptr(qd), _ffi.ptr(kd), _ffi.ptr(vd), stride, _ffi.ptr(tiles), _ffi.ptr(ps), _ffi.ptr(offq), _ffi.ptr(amq), _ffi.ptr(offk),
                                      _ffi.ptr(gws), _ffi.ptr(da), _ffi.ptr(attn), inner, _ffi.ptr(inv), _ffi.ptr(dq), _ffi.ptr(dk), _ffi.ptr(dv), B, N, G, m,
                                      _ffi.ptr(state), _ffi.ptr(state2), _ffi.ptr(dden), _ffi.ptr(tsum), _ffi.ptr(dq_lp), _ffi.ptr(dk_lp), _ffi.ptr(dv_lp), None, st))
    torch.cuda.synchronize()
    # the mirrors hold exactly the fp32 outputs rounded to bf16 (incl. the key row the fix-up launch touches), and nothing outside the written columns
    assert torch.equal(attn_lp, attn.to(torch.bfloat16)) and torch.equal(dqkv_lp, dqkv.to(torch.bfloat16))
    assert float((attn[:, G * dh:] - 7.0).abs().max()) == 0.0 and float((dqkv[:, G * dh:inner] - 3.0).abs().max()) == 0.0   # only the global-head columns are written
    un = lambda t: t[:, :G * dh].reshape(B, N, G, dh).permute(0, 2, 1, 3).cpu()
    return un(attn), un(dq), un(dk), un(dv), dict(offq=offq, offk=offk, amq=amq, gws=gws, ps=ps, flag=flag)


@pytest.mark.parametrize("seq_states", [False, True])
@pytest.mark.parametrize("B,G,N,m", [(2, 2, 150, 266), (1, 3, 64, 266), (2, 1, 333, 266), (1, 2, 77, 120), (1, 8, 1400, 266)])
def test_fused_favor_matches_fp64_reference(B, G, N, m, seq_states):
    """Both forms of the chunk states: the parallel launch + prefix launch (what batches below 40 (batch, head) pairs take) and the sequential walk."""
    from synthanatomy_amd import debug
    g = torch.Generator().manual_seed(N + m)
    q, k, v = (torch.randn(B, G, N, 64, generator=g) for _ in range(3))
    dattn = torch.randn(B, G, N, 64, generator=g)
    proj = P.gaussian_orthogonal_random_matrix(m, 64, g)
    ref = _reference(q, k, v, proj, dattn)
    with debug.override(favor_seq_always=seq_states):
        out, dq, dk, dv, aux = _fused(q, k, v, proj, dattn)
    # pre-pass: row offsets, argmax and the global key maximum
    c = 64 ** -0.25
    ddq = torch.einsum("bgnd,md->bgnm", q.double() * c, proj.double())
    ddk = torch.einsum("bgnd,md->bgnm", k.double() * c, proj.double())
    offq_ref = (q.double() ** 2).sum(-1) * c * c / 2 + ddq.max(-1).values
    got_offq = aux["offq"].view(B, N, G).permute(0, 2, 1).cpu().double()
    assert float((got_offq - offq_ref).abs().max()) < 2e-3
    got_am = aux["amq"].view(B, N, G).permute(0, 2, 1).cpu().long()
    picked = ddq.gather(-1, got_am[..., None]).squeeze(-1)
    assert float((picked - ddq.max(-1).values).abs().max()) < 2e-3          # the argmax (ties / near-ties may pick a neighbour of equal value)
    from synthanatomy_amd import _ffi  # noqa: F401
    errs = dict(out=_rel(out, ref[0]), dq=_rel(dq, ref[1]), dk=_rel(dk, ref[2]), dv=_rel(dv, ref[3]))
    fro = dict(out=_fro(out, ref[0]), dq=_fro(dq, ref[1]), dk=_fro(dk, ref[2]), dv=_fro(dv, ref[3]))
    print(f"[fused favor B{B} G{G} N{N} m{m}] max-rel {errs}  fro {fro}")
    assert errs["out"] < 1e-4 and errs["dv"] < 2e-4, errs
    assert fro["dq"] < 2e-3 and fro["dk"] < 2e-3 and errs["dq"] < 1e-2 and errs["dk"] < 1e-2, (errs, fro)


@pytest.mark.parametrize("seq_states", [False, True])
@pytest.mark.parametrize("B,G,N,m", [(2, 2, 150, 266), (1, 8, 1400, 266), (1, 2, 77, 120)])
def test_bf16_representable_projection_takes_the_two_product_path_bit_identically(B, G, N, m, seq_states):
    """Round 6: a projection operand whose lo halves are all zero (the throughput mode's bf16 copy of the folded matrix) is recognised by sa_favor_fused_proj_tiles
    (flag word 0) and every kernel skips the P_lo * x_hi products and the lo half of the slab transfers.  Forcing the flag to 1 runs the three-product path on the
    SAME tiles: every output must agree bit for bit (the skipped terms are exact zeros); an fp32 matrix keeps the flag at 1."""
    from synthanatomy_amd import debug
    g = torch.Generator().manual_seed(N + m + 1)
    q, k, v = (torch.randn(B, G, N, 64, generator=g) for _ in range(3))
    dattn = torch.randn(B, G, N, 64, generator=g)
    proj = P.gaussian_orthogonal_random_matrix(m, 64, g)
    ps16 = (proj * 64 ** -0.25).to(torch.bfloat16).float()
    with debug.override(favor_seq_always=seq_states):
        two = _fused(q, k, v, proj, dattn, ps=ps16)
        three = _fused(q, k, v, proj, dattn, ps=ps16, force_flag=1)
        full = _fused(q, k, v, proj, dattn)
    assert two[4]["flag"] == 0 and full[4]["flag"] == 1
    for a, b, what in zip(two[:4], three[:4], ("out", "dq", "dk", "dv")):
        assert torch.equal(a, b), what
    for key in ("offq", "offk", "amq", "gws"):
        assert torch.equal(two[4][key], three[4][key]), key
    # and the rounded operand stays within bf16 rounding of the fp32 one (what the throughput mode trades: 2^-9 relative on P)
    assert _fro(two[0], full[0]) < 2e-2 and _fro(two[3], full[3]) < 2e-2


def test_fused_favor_is_deterministic_and_ignores_the_future():
    g = torch.Generator().manual_seed(3)
    B, G, N, m = 1, 2, 200, 266
    q, k, v = (torch.randn(B, G, N, 64, generator=g) for _ in range(3))
    dattn = torch.randn(B, G, N, 64, generator=g)
    proj = P.gaussian_orthogonal_random_matrix(m, 64, g)
    a = _fused(q, k, v, proj, dattn)
    b = _fused(q, k, v, proj, dattn)
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)
    v2 = v.clone()
    v2[:, :, 130:] += 5.0
    c = _fused(q, k, v2, proj, dattn)
    assert torch.equal(c[0][:, :, :130], a[0][:, :, :130])      # values of later positions never reach earlier outputs (keys would, through the global maximum)


@pytest.mark.parametrize("B,G,L,N,W", [(2, 2, 2, 333, 64), (1, 3, 1, 1000, 420), (6, 8, 8, 1400, 420)])
def test_colaunched_local_heads_equal_separate_launches(B, G, L, N, W):
    """sa_favor_fused_fwd / _bwd with sa_local_attn_args (the local-window heads' blocks appended to the FAVOR+ launches) against the same calls without it
    followed by sa_local_attn_fwd / _bwd: every output (attention rows and their bf16 mirror, lse, dq / dk in rotated space, dv and its mirror, D, the FAVOR+
    gradients) bit-identical."""
    import ctypes
    from synthanatomy_amd import _ffi
    lib, st = _ffi.lib(), _ffi.stream()
    torch.manual_seed(N + W)
    dh, m = 64, 266
    H = G + L
    inner, R = H * dh, B * N
    stride = 3 * inner
    qkv = torch.randn(R, stride, device="cuda")
    qd, kd, vd = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
    qkr = torch.randn(2, R, L * dh, device="cuda")            # "rotated" q | k of the local heads
    ps = (torch.randn(m, dh) * dh ** -0.25).cuda()
    tiles = torch.empty(5 * 16384, dtype=torch.uint8, device="cuda")
    _ffi.check(lib.sa_favor_fused_proj_tiles(_ffi.ptr(ps), m, _ffi.ptr(tiles), st))
    offq, offk = torch.empty(R * G, device="cuda"), torch.empty(R * G, device="cuda")
    amq, gws = torch.empty(R * G, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    _ffi.check(lib.sa_favor_fused_prepass(_ffi.ptr(qd), _ffi.ptr(kd), stride, G, _ffi.ptr(tiles), _ffi.ptr(offq), _ffi.ptr(amq), _ffi.ptr(offk), _ffi.ptr(gws), R * G, m, dh, st))
    nst = lib.sa_favor_fused_state_bytes(B, N, G, m) // 4
    da = torch.randn(R, inner, device="cuda")

    def run(co):
        state, ws = torch.empty(nst, device="cuda"), torch.empty(nst, device="cuda")
        attn, attn_lp = torch.zeros(R, inner, device="cuda"), torch.zeros(R, inner, device="cuda", dtype=torch.bfloat16)
        inv, lse = torch.empty(R * G, device="cuda"), torch.empty(R * L, device="cuda")
        a = _ffi.LocalAttnArgs()
        a.q, a.k, a.v = qkr[0].data_ptr(), qkr[1].data_ptr(), vd.data_ptr()
        a.q_stride, a.q_off, a.k_stride, a.k_off, a.v_stride, a.v_off, a.o_stride, a.o_off = L * dh, 0, L * dh, 0, stride, G * dh, inner, G * dh
        a.o, a.lse, a.o_lp, a.L, a.W = attn.data_ptr(), lse.data_ptr(), attn_lp.data_ptr(), L, W
        _ffi.check(lib.sa_favor_fused_fwd(_ffi.ptr(qd), _ffi.ptr(kd), _ffi.ptr(vd), stride, _ffi.ptr(tiles), _ffi.ptr(ps), _ffi.ptr(offq), _ffi.ptr(offk), _ffi.ptr(gws),
                                          _ffi.ptr(attn), inner, _ffi.ptr(inv), 1e-6, B, N, G, m, _ffi.ptr(state), _ffi.ptr(attn_lp), ctypes.byref(a) if co else None, st))
        if not co:
            _ffi.check(lib.sa_local_attn_fwd(_ffi.ptr(qkr[0]), L * dh, 0, _ffi.ptr(qkr[1]), L * dh, 0, _ffi.ptr(vd), stride, G * dh, _ffi.ptr(attn), inner, G * dh,
                                             _ffi.ptr(lse), B, N, L, W, dh, _ffi.ptr(attn_lp), st))
        dqkv = torch.zeros(R, stride, device="cuda")
        dqkv_lp = torch.zeros(R, stride, device="cuda", dtype=torch.bfloat16)
        dq, dk, dv = dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:]
        dq_lp, dk_lp, dv_lp = dqkv_lp[:, :inner], dqkv_lp[:, inner:2 * inner], dqkv_lp[:, 2 * inner:]
        dqkr, Db = torch.zeros(2, R, L * dh, device="cuda"), torch.zeros(R * L, device="cuda")
        dden, tsum = torch.empty(R * G, device="cuda"), torch.zeros(B * G * ((N + 63) // 64), device="cuda")
        a.out, a.dout, a.lse_in = attn.data_ptr(), da.data_ptr(), lse.data_ptr()
        a.dq, a.dk, a.dv, a.Dbuf, a.dv_lp = dqkr[0].data_ptr(), dqkr[1].data_ptr(), dv.data_ptr(), Db.data_ptr(), dv_lp.data_ptr()
        _ffi.check(lib.sa_favor_fused_bwd(_ffi.ptr(qd), _ffi.ptr(kd), _ffi.ptr(vd), stride, _ffi.ptr(tiles), _ffi.ptr(ps), _ffi.ptr(offq), _ffi.ptr(amq), _ffi.ptr(offk),
                                          _ffi.ptr(gws), _ffi.ptr(da), _ffi.ptr(attn), inner, _ffi.ptr(inv), _ffi.ptr(dq), _ffi.ptr(dk), _ffi.ptr(dv), B, N, G, m,
                                          _ffi.ptr(state), _ffi.ptr(ws), _ffi.ptr(dden), _ffi.ptr(tsum), _ffi.ptr(dq_lp), _ffi.ptr(dk_lp), _ffi.ptr(dv_lp),
                                          ctypes.byref(a) if co else None, st))
        if not co:
            _ffi.check(lib.sa_local_attn_bwd(_ffi.ptr(qkr[0]), L * dh, 0, _ffi.ptr(qkr[1]), L * dh, 0, _ffi.ptr(vd), stride, G * dh, _ffi.ptr(attn), _ffi.ptr(da), inner,
                                             G * dh, _ffi.ptr(lse), _ffi.ptr(dqkr[0]), _ffi.ptr(dqkr[1]), _ffi.ptr(dv), _ffi.ptr(Db), B, N, L, W, dh, _ffi.ptr(dv_lp), st))
        torch.cuda.synchronize()
        return dict(attn=attn, attn_lp=attn_lp, lse=lse, dqkv=dqkv, dqkv_lp=dqkv_lp, dqkr=dqkr, Db=Db)

    a, b = run(True), run(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert float(a["attn"][:, G * dh:].abs().max()) > 0 and float(a["dqkr"].abs().max()) > 0
