"""GPU: the three-stage 128 x 256 ring for dense layers with few wide tiles (csrc/dense_ring.hip, an opt-in A/B instance: SA_DENSE_RING=1 -- measured slower
than the product kernel, kept with its measurements) against the two-stage im2col-order kernel that runs those shapes (same operands, same K order, same epilogue -> bit-identical) and against torch, on the Performer's 512-column layer shapes, a ragged row count,
every epilogue the dense layers use (bias, ReZero gate + residual + pre-activation and bf16 copies, GELU, the GELU-derivative mask of the data gradient)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _op(K, N, bias=True):
    from synthanatomy_amd import engine
    g = torch.Generator().manual_seed(K + N)
    w = (torch.randn(N, K, 1, 1, 1, generator=g) * K ** -0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    return engine.ConvOp("conv", K, N, 1, 1, 0, w, b, torch.bfloat16), w, b


@pytest.mark.parametrize("R,K,N", [(8400, 2048, 512), (8400, 1024, 512), (8400, 3072, 512), (1000, 512, 256), (77, 512, 512), (2800, 576, 768)])
def test_ring_equals_two_stage_kernel_and_torch(R, K, N):
    from synthanatomy_amd import _ffi, debug
    op, w, b = _op(K, N)
    x = torch.randn(1, 1, 1, R, K, device="cuda").to(torch.bfloat16)
    with debug.override(dense_ring=True), _ffi.kernel_log() as names:
        y = op.fprop(x, out_dtype=torch.float32)
        torch.cuda.synchronize()
    assert "dense_ring_kernel" in "\n".join(names), names
    with _ffi.kernel_log() as names0:
        y0 = op.fprop(x, out_dtype=torch.float32)
        torch.cuda.synchronize()
    assert "dense_ring_kernel" not in "\n".join(names0)      # opt-in instance: the product path stays on the two-stage kernel
    assert torch.equal(y, y0)
    ref = x.view(R, K).float() @ w.view(N, K).to(torch.bfloat16).float().t() + b
    assert _rel(y.view(R, N), ref) < 2e-5


def test_ring_epilogues_match_two_stage_kernel():
    """ReZero residual form (alpha = gate, fp32 addend, pre-activation + bf16 copies), GELU with its pre-activation copy, bf16 output, and the data-gradient
    form (GELU-derivative mask, fp32 addend) -- every variant the Performer's dense layers launch."""
    from synthanatomy_amd import debug
    from synthanatomy_amd._ffi import ACT_GELU, MASK_GELU
    R, K, N = 1400, 1024, 512
    op, w, b = _op(K, N)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 1, 1, R, K, generator=g).cuda().to(torch.bfloat16)
    res = torch.randn(1, 1, 1, R, N, generator=g).cuda()
    msk = torch.randn(1, 1, 1, R, N, generator=g).cuda().to(torch.bfloat16)
    gate = torch.tensor(0.3, device="cuda")

    def variants():
        yield op.fprop(x, out_dtype=torch.float32, alpha=gate, addend=res, want_pre=True, want_lp=True)
        yield op.fprop(x, act=ACT_GELU, want_pre=True)
        yield (op.fprop(x),)
        yield (op.fprop(x, out_dtype=torch.float32, addend=res, mask=msk, mask_mode=MASK_GELU, use_bias=False),)

    with debug.override(dense_ring=True):
        got = [tuple(t.clone() for t in v if t is not None) for v in variants()]
    want = [tuple(t.clone() for t in v if t is not None) for v in variants()]
    for gv, wv in zip(got, want):
        assert len(gv) == len(wv)
        for a, c in zip(gv, wv):
            assert a.dtype == c.dtype and torch.equal(a, c)
    # and the first one against torch
    y, pre, lp = got[0]
    lin = x.view(R, K).float() @ w.view(N, K).to(torch.bfloat16).float().t() + b
    assert _rel(y.view(R, N), res.view(R, N) + 0.3 * lin) < 2e-5
    assert torch.equal(lp, y.to(torch.bfloat16)) and _rel(pre.float().view(R, N), lin) < 8e-3
