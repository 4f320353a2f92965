"""GPU: the Performer at its PRODUCTION dimensions against the CPU oracle (oracle/performer_ref.py -- parity UNPINNED against the third-party
performer-pytorch 1.0.11, see that file; if tests/golden/performer.npz exists it was written by make_goldens_performer.py from the real packages
and test_performer_oracle.py pins the oracle to it).

README network widths (reference README.md:126-141, src/networks/transformers/performer.py:194-221,229-288): dim 512, 16 heads of 64 (inner
1024, m = 266 random features), 8 local heads with window 420, ReZero, absolute + absolute-spatial embeddings, raster ordering with the README
transforms; N = 1 400 (10x14x10: 3 1/3 windows) and N = 1 000 (10x10x10: the sequence ends in the middle of window 3).  Depth 2, batch 2 (the
oracle's quadratic forms need ~10 s for it).  The toy tests of test_performer_gpu.py never reach the kernels the 512-wide network runs on:
here the kernel log asserts that the chunked split-bf16 scans, the flash-style local attention, the fused feature-map / projection backward and
the single q|k|v dense layer are what actually ran.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import performer_ref as P  # noqa: E402

DIM, HEADS, LOCAL, WINDOW, VOCAB = 512, 16, 8, 420, 2048
README_ORDER = dict(ordering_type="raster_scan", spatial_dims=3, reflected_spatial_dims=(False, False, False), transpositions_axes=((2, 0, 1),),
                    rot90_axes=((0, 1),), transformation_order=("rotate_90", "transpose"))


def _rel_max(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def _rel_fro(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def _ordering(spatial):
    from synthanatomy_amd.networks.transformers.img2seq_ordering import Ordering
    return Ordering(README_ORDER["ordering_type"], README_ORDER["spatial_dims"], (1,) + tuple(spatial), README_ORDER["reflected_spatial_dims"],
                    README_ORDER["transpositions_axes"], README_ORDER["rot90_axes"], README_ORDER["transformation_order"])


def _setup(spatial, depth=2, seed=11):
    n = int(np.prod(spatial))
    cfg = P.PerformerConfig(num_tokens=VOCAB + 1, max_seq_len=n, dim=DIM, depth=depth, heads=HEADS, dim_head=64, local_attn_heads=LOCAL, local_window_size=WINDOW,
                            spatial_shape=tuple(spatial), use_rezero=True)
    st = P.init_state(cfg, seed=seed)
    for k in st:
        if k.endswith(".g"):
            st[k] = torch.tensor(0.2)      # the 1e-3 ReZero init would hide the attention path behind the residual
    return cfg, st, n


def _product(cfg, st, order, dtype):
    from synthanatomy_amd.networks.transformers.performer import Performer
    from synthanatomy_amd.runtime.optim import FlatParams
    net = Performer(num_tokens=cfg.num_tokens, max_seq_len=cfg.max_seq_len, dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, ordering=order, dim_head=cfg.dim_head,
                    local_attn_heads=cfg.local_attn_heads, local_window_size=cfg.local_window_size, use_rezero=True, spatial_position_emb="absolute",
                    spatial_shape=cfg.spatial_shape, feature_redraw_interval=None, compute_dtype=dtype)
    missing, unexpected = net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(("spatial_indices_sequence" in k or "inv_freq" in k or "calls_since" in k) for k in missing), missing
    net = net.cuda().train()
    flat = FlatParams(net.parameters())     # the layout the CLIs and bench.py train in: to_q / to_k / to_v adjacent -> ONE dense layer
    return net, flat


def _oracle(cfg, st, tok, tgt, seqs):
    leaf = {k: v.clone().requires_grad_(True) for k, v in st.items() if "projection_matrix" not in k}
    stt = dict(st)
    stt.update(leaf)
    ref = P.forward(stt, cfg, tok, seqs)
    loss = P.ce_loss(ref, tgt)
    loss.backward()
    return ref.detach(), float(loss.detach()), {k: p.grad for k, p in leaf.items() if p.grad is not None}


@pytest.fixture(scope="module", params=[(10, 14, 10), (10, 10, 10)], ids=["n1400", "n1000"])
def case(request):
    spatial = request.param
    cfg, st, n = _setup(spatial)
    order = _ordering(spatial)
    seqs = P.spatial_index_sequences(spatial, order.get_sequence_ordering())
    g = torch.Generator().manual_seed(5)
    tok = torch.randint(0, VOCAB + 1, (2, n), generator=g)
    tok[:, 0] = VOCAB                                               # BOS, as prepare_batch pads it
    tgt = torch.randint(0, VOCAB, (2, n), generator=g)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref, ref_loss, ref_grads = _oracle(cfg, st, tok, tgt, seqs)
    return dict(cfg=cfg, st=st, n=n, order=order, tok=tok, tgt=tgt, ref=ref, ref_loss=ref_loss, ref_grads=ref_grads)


def _run(case, dtype):
    from synthanatomy_amd import _ffi
    from synthanatomy_amd.losses.transformer import CELoss
    net, flat = _product(case["cfg"], case["st"], case["order"], dtype)
    with _ffi.kernel_log() as names:
        out = net(case["tok"].cuda())
        loss = CELoss()(out.transpose(1, 2), case["tgt"].cuda())
        loss.backward()
        torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    fused = ["to_qkv" in l.ops for l in net._chain.layers]
    return out.detach(), float(loss.detach()), grads, names, fused


def test_fp32_engine_matches_oracle_at_production_width(case):
    """north_star's parity bar on the 512-wide network: logits <= 1e-3 relative, every parameter gradient <= 3e-3 (Frobenius)."""
    out, loss, grads, names, fused = _run(case, torch.float32)
    n = case["n"]
    assert out.shape == (2, n, VOCAB + 1)
    e_logits = _rel_max(out, case["ref"])
    assert e_logits < 1e-3, e_logits
    assert abs(loss - case["ref_loss"]) < 1e-4 * abs(case["ref_loss"]), (loss, case["ref_loss"])
    worst_f, worst_m = ("", 0.0), ("", 0.0)
    checked = 0
    for k, gr in case["ref_grads"].items():
        if float(gr.abs().max()) == 0.0:
            continue
        assert k in grads, k
        f, m = _rel_fro(grads[k], gr), _rel_max(grads[k], gr)
        worst_f = max(worst_f, (k, f), key=lambda t: t[1])
        worst_m = max(worst_m, (k, m), key=lambda t: t[1])
        checked += 1
    print(f"[performer width parity fp32 N={n}] logits max-rel {e_logits:.2e}; worst gradient fro {worst_f[1]:.2e} ({worst_f[0]}), max-rel {worst_m[1]:.2e} ({worst_m[0]})")
    assert checked >= 2 * 8 + 7, checked
    assert worst_f[1] < 3e-3, worst_f
    assert worst_m[1] < 1e-2, worst_m
    assert all(fused), "to_q / to_k / to_v did not run as one dense layer"
    joined = "\n".join(names)
    # fp32 engine: exact-fp32 chunked scans (state_flags bit 2), flash-style local attention, exact dense layers
    for need in ("favor_chunk_state_kernel", "favor_chunk_out_a_kernel", "favor_chunk_out_b_kernel", "local_attn_q_split_kernel", "local_attn_kv_split_kernel",
                 "conv_fprop_dma_kernel<float"):
        assert need in joined, (need, names)


def test_bf16_engine_follows_oracle_at_production_width(case):
    """The benchmarked mode (bf16 dense layers, split-bf16 FAVOR+ / local attention) against the fp32 oracle: deviations printed and gated at the
    bf16 rounding of a 2-layer, 512-wide network; the kernels of the throughput path are the ones that ran."""
    from synthanatomy_amd import debug
    with debug.override(favor_seq_always=True):     # the README batch's form of the chunk states (a batch of one would take the parallel launches)
        out, loss, grads, names, fused = _run(case, torch.bfloat16)
    n = case["n"]
    e_logits = _rel_max(out, case["ref"])
    e_fro = _rel_fro(out, case["ref"])
    num = den = 0.0
    worst = ("", 0.0)
    for k, gr in case["ref_grads"].items():
        if float(gr.abs().max()) == 0.0:
            continue
        d = (grads[k].double().cpu() - gr.double())
        num += float(d.pow(2).sum())
        den += float(gr.double().pow(2).sum())
        f = _rel_fro(grads[k], gr)
        if not any(t in k for t in ("to_q", "to_k")):       # the query / key side of FAVOR+ is ill-conditioned: compared in aggregate below
            worst = max(worst, (k, f), key=lambda t: t[1])
    agg = (num / den) ** 0.5
    print(f"[performer width parity bf16 N={n}] logits max-rel {e_logits:.2e} fro {e_fro:.2e}; loss {loss:.5f} vs {case['ref_loss']:.5f}; "
          f"gradients: aggregate fro {agg:.2e}, worst single {worst[1]:.2e} ({worst[0]})")
    assert e_logits < 3e-2 and e_fro < 1.5e-2, (e_logits, e_fro)
    assert abs(loss - case["ref_loss"]) < 2e-3 * abs(case["ref_loss"])
    assert agg < 2e-2 and worst[1] < 5e-2, (agg, worst)
    assert all(fused)
    joined = "\n".join(names)
    # FAVOR+ with the feature maps recomputed on chip (csrc/favor_fused.hip), flash-style local attention, bf16 dense layers
    # (backward: the independent chunk kernels share launches -- [scan B dq | reversed states], [scan B dk | scan A dv])
    #  and the local-window heads' split-bf16 blocks ride in the FAVOR+ launches: the *_la_kernel instances)
    for need in ("favor_prepass_kernel", "favor_fseq_la_kernel", "favor_fout_a_kernel", "favor_fpair_seq_b_la_kernel", "favor_fpair_b_a_la_kernel",
                 "conv_fprop_dma_kernel<unsigned short"):
        assert need in joined, (need, names)
    for gone in ("favor_project_fwd_kernel", "favor_feat_fwd_kernel", "favor_feat_proj_bwd_kernel", "favor_chunk_out_b_split_kernel"):
        assert gone not in joined, (gone, names)     # nothing writes dd / phi / d phi to HBM any more


def test_fused_and_unfused_favor_agree_at_production_width(case):
    """The same bf16 network through the unfused chain (projection -> feature map -> chunked split-bf16 scans -> fused feature / projection backward,
    SA_NO_FUSED_FAVOR) and through the on-chip feature maps: logits and every gradient agree to the split-bf16 error, amplified by the bf16 dense layers."""
    from synthanatomy_amd import debug
    out_f, loss_f, grads_f, names_f, _ = _run(case, torch.bfloat16)
    with debug.override(no_fused_favor=True):
        out_u, loss_u, grads_u, names_u, _ = _run(case, torch.bfloat16)
    ju = "\n".join(names_u)
    for need in ("favor_chunk_state_split_kernel", "favor_chunk_out_a_split_kernel", "favor_chunk_out_b_split_kernel", "favor_feat_proj_bwd_kernel"):
        assert need in ju, (need, names_u)
    assert "favor_fout_b_kernel" not in ju and "favor_fpair_b_a" not in ju
    e = _rel_fro(out_f, out_u)
    num = sum(float((grads_f[k].double() - grads_u[k].double()).pow(2).sum()) for k in grads_u)
    den = sum(float(grads_u[k].double().pow(2).sum()) for k in grads_u)
    print(f"[fused vs unfused N={case['n']}] logits fro {e:.2e}, gradients aggregate fro {(num / den) ** 0.5:.2e}")
    assert e < 3e-3 and (num / den) ** 0.5 < 6e-3 and abs(loss_f - loss_u) < 1e-4 * abs(loss_u)
