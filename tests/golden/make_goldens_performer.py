#!/usr/bin/env python3
"""Generate ``performer.npz`` from the REAL third-party packages the reference's Performer delegates to -- the day they are installable:

    pip install performer-pytorch==1.0.11          # docker/requirements.txt:10; pulls local-attention (un-pinned upstream)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens_performer.py

``/root/reference/src/networks/transformers/performer.py:8-16,194-219`` builds ``performer_pytorch.Performer`` with 24 positional
arguments; neither that package nor ``local_attention`` / ``axial_positional_embedding`` / ``fast_transformers`` is in the offline image this
build was made in, so oracle/performer_ref.py is a restatement of the published algorithm and the Performer rows of the parity table are
"unpinned".  This script closes that gap without touching any test: when the packages import it runs

* the layer stack alone (``performer_pytorch.Performer`` with the reference's argument order: causal, ReZero, local heads + window, fixed
  projections) on seeded inputs -- output, and the gradient of a seeded scalar with respect to the input and every parameter;
* ``FastAttention`` pieces in isolation (``softmax_kernel`` for queries and keys, ``causal_linear_attention_noncuda``) -- the two places where
  the restatement had to decide something the source could not be consulted for (key-stabiliser scope, +eps placement);
* ``local_attention.LocalAttention`` alone with the wrapper's settings (window, causal, look_backward 1, autopad, rel_pos_emb_config=(64, 8)) --
  the version-dependent piece (rotary vs learned relative positions); the package version is recorded;
* and, if ``axial_positional_embedding`` is importable too, the reference's own wrapper class end to end (token / spatial / positional
  embeddings, final norm, vocabulary projection).

Everything is written as tensors under ``case/...`` keys plus a JSON ``meta`` (package versions, shapes, seeds).  tests/test_performer_oracle.py
::test_oracle_against_third_party_golden picks the file up when present (and skips, saying "parity unpinned", when it is not); the GPU parity tests
compare the HIP path with that same oracle, so the pin carries through.  No source of any package is stored -- data only.
"""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _version(mod_name):
    try:
        from importlib import metadata
        return metadata.version(mod_name)
    except Exception:
        return "unknown"


def _np(t):
    return t.detach().cpu().numpy()


def stack_case(out, tag, dim, depth, heads, dim_head, local_heads, window, n, batch, seed, nb_features=None):
    """performer_pytorch.Performer exactly as performer.py:194-219 constructs it (use_rezero=True, causal, qkv / out bias False)."""
    from performer_pytorch import Performer
    torch.manual_seed(seed)
    net = Performer(dim, depth, heads, dim_head, local_heads, window, True, 4, nb_features, None, False, 1, False, torch.nn.ReLU(), False, True, False, 0.0, 0.0,
                    False, False, True, False, False)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if k.endswith(".g"):
                p.fill_(0.3)          # ReZero gates start at 1e-3: make the blocks matter
    net.train()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, n, dim, generator=g, requires_grad=True)
    w = torch.randn(batch, n, dim, generator=g)
    y = net(x)
    (y * w).sum().backward()
    out[f"{tag}/x"], out[f"{tag}/w"], out[f"{tag}/y"], out[f"{tag}/dx"] = _np(x), _np(w), _np(y), _np(x.grad)
    for k, v in net.state_dict().items():
        out[f"{tag}/sd/{k}"] = _np(v)
    for k, p in net.named_parameters():
        if p.grad is not None:
            out[f"{tag}/grad/{k}"] = _np(p.grad)
    return dict(dim=dim, depth=depth, heads=heads, dim_head=dim_head, local_heads=local_heads, window=window, n=n, batch=batch, seed=seed,
                nb_features=nb_features)


def favor_case(out, tag, n, d, m, seed):
    import performer_pytorch.performer_pytorch as pp
    g = torch.Generator().manual_seed(seed)
    proj = pp.gaussian_orthogonal_random_matrix(m, d)
    q, k, v = (torch.randn(2, 3, n, d, generator=g) for _ in range(3))
    qp = pp.softmax_kernel(q, projection_matrix=proj, is_query=True)
    kp = pp.softmax_kernel(k, projection_matrix=proj, is_query=False)
    o = pp.causal_linear_attention_noncuda(qp, kp, v, chunk_size=16)
    for name, t in (("proj", proj), ("q", q), ("k", k), ("v", v), ("qp", qp), ("kp", kp), ("out", o)):
        out[f"{tag}/{name}"] = _np(t)
    return dict(n=n, d=d, m=m, seed=seed)


def local_case(out, tag, n, d, window, seed):
    from local_attention import LocalAttention
    torch.manual_seed(seed)
    la = LocalAttention(window_size=window, causal=True, autopad=True, dropout=0.0, look_forward=0, rel_pos_emb_config=(d, 8))
    g = torch.Generator().manual_seed(seed + 1)
    q, k, v = (torch.randn(2, 3, n, d, generator=g) for _ in range(3))
    o = la(q, k, v)
    for name, t in (("q", q), ("k", k), ("v", v), ("out", o)):
        out[f"{tag}/{name}"] = _np(t)
    params = {kk: _np(vv) for kk, vv in la.state_dict().items()}
    for kk, vv in params.items():
        out[f"{tag}/sd/{kk}"] = vv
    return dict(n=n, d=d, window=window, seed=seed, state_keys=sorted(params))


def wrapper_case(out, tag, spatial, dim, depth, heads, local_heads, window, seed):
    """The reference's own class (needs axial_positional_embedding for its import line, performer.py:7)."""
    sys.path.insert(0, REF)
    from src.networks.transformers.img2seq_ordering import Ordering
    from src.networks.transformers.performer import Performer
    n = int(np.prod(spatial))
    torch.manual_seed(seed)
    order = Ordering("raster_scan", 3, (1,) + tuple(spatial), (False, False, False), ((2, 0, 1),), ((0, 1),), ("rotate_90", "transpose"))
    net = Performer(num_tokens=33, max_seq_len=n, dim=dim, depth=depth, heads=heads, ordering=order, local_attn_heads=local_heads, local_window_size=window,
                    feature_redraw_interval=None, use_rezero=True, spatial_position_emb="absolute", spatial_shape=tuple(spatial))
    with torch.no_grad():
        for k, p in net.named_parameters():
            if k.endswith(".g"):
                p.fill_(0.3)
    net.train()
    g = torch.Generator().manual_seed(seed + 1)
    tok = torch.randint(0, 33, (2, n), generator=g)
    tgt = torch.randint(0, 32, (2, n), generator=g)
    logits = net(tok)
    loss = torch.nn.functional.cross_entropy(logits.transpose(1, 2).float(), tgt.long())
    loss.backward()
    out[f"{tag}/tok"], out[f"{tag}/tgt"], out[f"{tag}/logits"], out[f"{tag}/loss"] = _np(tok), _np(tgt), _np(logits), _np(loss)
    out[f"{tag}/ordering"] = np.asarray(order.get_sequence_ordering())
    for k, v in net.state_dict().items():
        out[f"{tag}/sd/{k}"] = _np(v)
    for k, p in net.named_parameters():
        if p.grad is not None:
            out[f"{tag}/grad/{k}"] = _np(p.grad)
    return dict(spatial=list(spatial), dim=dim, depth=depth, heads=heads, local_heads=local_heads, window=window, seed=seed)


def main():
    try:
        import local_attention  # noqa: F401
        import performer_pytorch  # noqa: F401
    except ImportError as exc:
        print(f"performer_pytorch / local_attention are not importable here ({exc}); nothing written -- the Performer oracle stays parity-unpinned.")
        return 1
    out, meta = {}, {"versions": {p: _version(p) for p in ("performer-pytorch", "local-attention", "torch", "einops")}, "cases": {}}
    # toy widths (what test_performer_gpu.py's oracle comparisons use) and one production-width, multi-window case
    meta["cases"]["stack_toy"] = stack_case(out, "stack_toy", dim=32, depth=2, heads=4, dim_head=64, local_heads=2, window=6, n=24, batch=2, seed=3)
    meta["cases"]["stack_global_only"] = stack_case(out, "stack_global_only", dim=32, depth=1, heads=2, dim_head=64, local_heads=0, window=6, n=20, batch=2, seed=4)
    meta["cases"]["stack_wide"] = stack_case(out, "stack_wide", dim=512, depth=1, heads=16, dim_head=64, local_heads=8, window=420, n=1000, batch=1, seed=5)
    meta["cases"]["favor"] = favor_case(out, "favor", n=48, d=64, m=266, seed=6)
    meta["cases"]["local"] = local_case(out, "local", n=50, d=64, window=16, seed=7)
    meta["cases"]["local_w420"] = local_case(out, "local_w420", n=1000, d=64, window=420, seed=8)
    try:
        meta["cases"]["wrapper"] = wrapper_case(out, "wrapper", spatial=(2, 3, 4), dim=32, depth=2, heads=4, local_heads=2, window=6, seed=9)
    except Exception as exc:   # axial_positional_embedding / reference tree missing: the stack cases already pin the arithmetic
        meta["wrapper_skipped"] = f"{type(exc).__name__}: {exc}"
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "performer.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays; versions {meta['versions']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
