"""ORACLE (test infrastructure only) -- CPU restatement of the Performer hot path (path B).

PARITY UNPINNED.  The arithmetic of this path does not live in the reference tree: ``src/networks/transformers/
performer.py:8-16,194-219`` delegates to the third-party ``performer-pytorch==1.0.11`` (docker/requirements.txt:10),
which uses the un-pinned ``local-attention`` package and, on CUDA, ``fast_transformers.causal_product`` (docker/Dockerfile:20).
None of them is installed or vendored here and the reference has no tests or golden vectors for this path, so this file
restates the PUBLISHED algorithms as best known and is the build's frozen spec ("spec by restatement"):

* wrapper        reference performer.py:70-288 (token / absolute-spatial / absolute positional embeddings, LayerNorm, to_out)
* layer stack    performer_pytorch.Performer with use_rezero=True: x += g_a * Attn(x); x += g_f * FF(x)  (no LayerNorm inside)
* SelfAttention  to_q/to_k/to_v (no bias) -> heads; first (heads - local_heads) heads FAVOR+, the rest local; to_out (no bias)
* FAVOR+         softmax_kernel: ratio * (exp(c x P^T - |x|^2 c^2 / 2 - stab) + 1e-4), c = d^-1/4, ratio = m^-1/2;
                 stab = per-row max for queries, GLOBAL max over the whole key tensor for keys (1.0.11 `torch.max(data_dash)`);
                 causal: out_i = (q'_i . sum_{j<=i} k'_j (x) v_j) / (q'_i . (sum_{j<=i} k'_j + 1e-6))
* local attention  local-attention >= 1.2 semantics: rotary (sinusoidal) embedding on q,k of the local heads, autopad to a
                 multiple of the window, look_backward=1, look_forward=0, causal mask q_pos < k_pos, softmax over <= 2*window keys
* FeedForward    Linear(dim, 4 dim) -> GELU(erf) -> Linear(4 dim, dim)
* projection     gaussian_orthogonal_random_matrix(nb_rows = int(d ln d), d, scaling=0)
* non-README options of the wrapper (round 6; performer.py:94-106,134-148,201,286-288): ``rotary_position_emb`` (performer_pytorch 1.0.11
                 ``apply_rotary_pos_emb``: q, k of the GLOBAL heads rotated pairwise -- rotate_every_two -- with the sin | cos rows of
                 FixedPositionalEmbedding(dim_head); the sinusoidal table of width dim is still added to x), ``axial_position_emb``
                 (axial_positional_embedding: parameters weights_0 [1, s0, 1, dim], weights_1 [1, 1, s1, dim], summed and flattened),
                 ``tie_embed`` (logits = x @ token_emb.weight^T), ``emb_dropout`` (nn.Dropout on the summed embeddings)

Known ambiguities: the key-stabiliser scope and the local-attention relative-position variant.
What IS pinned: ordering / batch preparation / sampling post-processing (tests/golden/{ordering,sample}.npz) and the
closed-form identities in tests/test_performer_oracle.py (chunk-free quadratic forms).

Only tests / smoke / bench's cpu_baseline may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class PerformerConfig:
    num_tokens: int = 2049
    max_seq_len: int = 1400
    dim: int = 512
    depth: int = 24
    heads: int = 16
    dim_head: int = 64
    local_attn_heads: int = 8
    local_window_size: int = 420
    ff_mult: int = 4
    nb_features: Optional[int] = None
    spatial_shape: tuple = (10, 14, 10)
    use_rezero: bool = True
    # the non-README options of the wrapper (reference performer.py:94-106,134-148,201,286-288)
    rotary_position_emb: bool = False       # sinusoidal pos_emb on x + rotate-every-two of q / k of the GLOBAL heads in every layer (performer_pytorch 1.0.11)
    axial_position_emb: bool = False        # axial_positional_embedding.AxialPositionalEmbedding(dim, axial_position_shape), summed form
    axial_position_shape: Optional[tuple] = None
    tie_embed: bool = False                 # logits = x @ token_emb.weight^T, no to_out

    @property
    def m(self):
        return self.nb_features or int(self.dim_head * math.log(self.dim_head))

    @property
    def inner(self):
        return self.heads * self.dim_head


def orthogonal_matrix_chunk(cols, gen):
    block = torch.randn((cols, cols), generator=gen)
    q, _ = torch.linalg.qr(block)
    return q.t()


def gaussian_orthogonal_random_matrix(nb_rows, nb_cols, gen):
    blocks = []
    full = nb_rows // nb_cols
    for _ in range(full):
        blocks.append(orthogonal_matrix_chunk(nb_cols, gen))
    rem = nb_rows - full * nb_cols
    if rem > 0:
        blocks.append(orthogonal_matrix_chunk(nb_cols, gen)[:rem])
    final = torch.cat(blocks)
    mult = torch.randn((nb_rows, nb_cols), generator=gen).norm(dim=1)
    return torch.diag(mult) @ final


def init_state(cfg: PerformerConfig, seed: int = 0, spatial_index_len: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Parameter dict with performer_pytorch's / the wrapper's state_dict names."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}

    def lin(prefix, out_f, in_f, bias=True):
        b = 1.0 / math.sqrt(in_f)
        st[prefix + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * b
        if bias:
            st[prefix + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * b

    n = cfg.max_seq_len
    st["token_emb.weight"] = torch.randn(cfg.num_tokens, cfg.dim, generator=g)
    st["pos_emb.emb.weight"] = torch.randn(n, cfg.dim, generator=g)
    for a in range(len(cfg.spatial_shape)):
        st[f"spatial_position_emb.{a}.emb.weight"] = torch.randn(spatial_index_len or (n - 1), cfg.dim, generator=g)
    for i in range(cfg.depth):
        p = f"performer.net.layers.{i}"
        if cfg.use_rezero:
            st[p + ".0.g"] = torch.tensor(1e-3)
            st[p + ".1.g"] = torch.tensor(1e-3)
        else:
            for j in (0, 1):
                st[p + f".{j}.norm.weight"] = torch.ones(cfg.dim) + 0.1 * torch.randn(cfg.dim, generator=g)
                st[p + f".{j}.norm.bias"] = 0.1 * torch.randn(cfg.dim, generator=g)
        lin(p + ".0.fn.to_q", cfg.inner, cfg.dim, bias=False)
        lin(p + ".0.fn.to_k", cfg.inner, cfg.dim, bias=False)
        lin(p + ".0.fn.to_v", cfg.inner, cfg.dim, bias=False)
        lin(p + ".0.fn.to_out", cfg.dim, cfg.inner, bias=False)
        st[p + ".0.fn.fast_attention.projection_matrix"] = gaussian_orthogonal_random_matrix(cfg.m, cfg.dim_head, g)
        lin(p + ".1.fn.fn.w1", cfg.dim * cfg.ff_mult, cfg.dim)
        lin(p + ".1.fn.fn.w2", cfg.dim, cfg.dim * cfg.ff_mult)
    st["norm.weight"] = torch.ones(cfg.dim)
    st["norm.bias"] = torch.zeros(cfg.dim)
    lin("to_out", cfg.num_tokens, cfg.dim)
    return st


# ------------------------------------------------------------------------------------------------ FAVOR+
def softmax_kernel(data, proj, is_query, eps=1e-4):
    d = data.shape[-1]
    c = d ** -0.25
    ratio = proj.shape[0] ** -0.5
    dash = torch.einsum("...id,jd->...ij", c * data, proj)
    diag = (data ** 2).sum(dim=-1, keepdim=True) / 2.0 * (c ** 2)
    if is_query:
        stab = dash.max(dim=-1, keepdim=True).values
    else:
        stab = dash.max()
    return ratio * (torch.exp(dash - diag - stab) + eps)


def causal_linear_attention(qp, kp, v, eps=1e-6):
    """Quadratic (chunk-free) statement: out_i = sum_{j<=i} (q'_i.k'_j) v_j / (q'_i . (sum_{j<=i} k'_j + eps))."""
    n = qp.shape[-2]
    a = torch.einsum("...im,...jm->...ij", qp, kp)
    a = a * torch.tril(torch.ones(n, n, dtype=a.dtype))
    num = a @ v
    kc = kp.cumsum(dim=-2) + eps
    den = (qp * kc).sum(dim=-1, keepdim=True)
    return num / den


def causal_linear_attention_scan(qp, kp, v, eps=1e-6):
    """Same quantity by the running-state recurrence (what fast_transformers' CausalDotProduct computes)."""
    out = torch.empty(*qp.shape[:-1], v.shape[-1], dtype=qp.dtype)
    S = torch.zeros(*qp.shape[:-2], qp.shape[-1], v.shape[-1], dtype=qp.dtype)
    z = torch.zeros(*qp.shape[:-2], qp.shape[-1], dtype=qp.dtype)
    for i in range(qp.shape[-2]):
        S = S + kp[..., i, :, None] * v[..., i, None, :]
        z = z + kp[..., i, :]
        num = torch.einsum("...m,...md->...d", qp[..., i, :], S)
        den = (qp[..., i, :] * (z + eps)).sum(-1, keepdim=True)
        out[..., i, :] = num / den
    return out


# ------------------------------------------------------------------------------------------------ local attention
def rotary_tables(n, dim):
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    t = torch.arange(n).float()
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    freqs = torch.cat((freqs, freqs), dim=-1)
    return freqs.cos(), freqs.sin()


def apply_rotary(x, cos, sin):
    half = x.shape[-1] // 2
    x1, x2 = x[..., :half], x[..., half:]
    rot = torch.cat((-x2, x1), dim=-1)
    return x * cos + rot * sin


def local_attention(q, k, v, window, rotary=True):
    """q,k,v [..., n, e] -> [..., n, e]; dense statement with the band mask (identical to the bucketed form)."""
    n, e = q.shape[-2], q.shape[-1]
    if rotary:
        cos, sin = rotary_tables(n, e)
        q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
    dots = torch.einsum("...ie,...je->...ij", q, k) * (e ** -0.5)
    i = torch.arange(n)[:, None]
    j = torch.arange(n)[None, :]
    lo = (i // window - 1).clamp(min=0) * window
    allowed = (j <= i) & (j >= lo)
    dots = dots.masked_fill(~allowed, -torch.finfo(dots.dtype).max)
    return dots.softmax(dim=-1) @ v


def local_attention_bucketed(q, k, v, window, rotary=True):
    """The bucketed form of the package (pad to a multiple of the window, look back one window)."""
    shape = q.shape
    n, e = shape[-2], shape[-1]
    q, k, v = (t.reshape(-1, n, e) for t in (q, k, v))
    if rotary:
        cos, sin = rotary_tables(n, e)
        q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
    pad = (-n) % window
    if pad:
        q, k, v = (F.pad(t, (0, 0, 0, pad)) for t in (q, k, v))
    t = n + pad
    w = t // window
    b = q.shape[0]
    ticker = torch.arange(t, dtype=q.dtype)[None, :].reshape(1, w, window)

    def look(x, padv):
        prev = F.pad(x, (0, 0) * (x.dim() - 2) + (1, 0), value=padv)[:, :-1] if x.dim() == 3 else F.pad(x, (0, 0, 0, 0, 1, 0), value=padv)[:, :-1]
        return torch.cat((prev, x), dim=2)

    bq, bk, bv = (x.reshape(b, w, window, e) for x in (q, k, v))
    bk, bv = look(bk, 0.0), look(bv, 0.0)
    tq = ticker
    tk = torch.cat((F.pad(ticker, (0, 0, 1, 0), value=-1)[:, :-1], ticker), dim=2)
    dots = torch.einsum("bhie,bhje->bhij", bq, bk) * (e ** -0.5)
    neg = -torch.finfo(dots.dtype).max
    dots = dots.masked_fill(tq[:, :, :, None] < tk[:, :, None, :], neg)
    dots = dots.masked_fill(tk[:, :, None, :] == -1, neg)
    out = torch.einsum("bhij,bhje->bhie", dots.softmax(dim=-1), bv).reshape(-1, t, e)[:, :n]
    return out.reshape(shape)


# ------------------------------------------------------------------------------------------------ network
def rotate_every_two(x):
    """performer_pytorch 1.0.11: (x0, x1, x2, x3, ...) -> (-x1, x0, -x3, x2, ...)"""
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).reshape(x.shape)


def apply_rotary_pos_emb(q, k, sinu_pos):
    """performer_pytorch 1.0.11 ``apply_rotary_pos_emb``: sinu_pos [n, dim_head] = sin | cos halves of FixedPositionalEmbedding(dim_head); every frequency
    serves a PAIR of consecutive dimensions."""
    half = sinu_pos.shape[-1] // 2
    sin, cos = sinu_pos[:, :half], sinu_pos[:, half:]
    sin, cos = sin.repeat_interleave(2, dim=-1), cos.repeat_interleave(2, dim=-1)
    return q * cos + rotate_every_two(q) * sin, k * cos + rotate_every_two(k) * sin


def axial_position_table(st, shape):
    """axial_positional_embedding.AxialPositionalEmbedding(dim, shape) in its summed form: position t of the flattened (s0, s1) grid gets
    weights_0[t // s1] + weights_1[t % s1]."""
    s0, s1 = shape
    w0, w1 = st["pos_emb.weights_0"], st["pos_emb.weights_1"]
    dim = w0.shape[-1]
    return (w0.expand(1, s0, s1, dim) + w1.expand(1, s0, s1, dim)).reshape(s0 * s1, dim)


def self_attention(st, p, cfg: PerformerConfig, x):
    b, n, _ = x.shape
    h, gh = cfg.heads, cfg.heads - cfg.local_attn_heads
    q = F.linear(x, st[p + ".to_q.weight"], st.get(p + ".to_q.bias"))
    k = F.linear(x, st[p + ".to_k.weight"], st.get(p + ".to_k.bias"))
    v = F.linear(x, st[p + ".to_v.weight"], st.get(p + ".to_v.bias"))
    q, k, v = (t.reshape(b, n, h, cfg.dim_head).permute(0, 2, 1, 3) for t in (q, k, v))
    outs = []
    if gh > 0:
        proj = st[p + ".fast_attention.projection_matrix"]
        qg, kg = q[:, :gh], k[:, :gh]
        if cfg.rotary_position_emb:      # SelfAttention.forward of performer_pytorch 1.0.11: only the global heads see pos_emb
            qg, kg = apply_rotary_pos_emb(qg, kg, fixed_position_table(cfg.dim_head, cfg.max_seq_len)[:n])
        qp = softmax_kernel(qg, proj, True)
        kp = softmax_kernel(kg, proj, False)
        outs.append(causal_linear_attention(qp, kp, v[:, :gh]))
    if gh < h:
        outs.append(local_attention(q[:, gh:], k[:, gh:], v[:, gh:], cfg.local_window_size))
    out = torch.cat(outs, dim=1).permute(0, 2, 1, 3).reshape(b, n, h * cfg.dim_head)
    return F.linear(out, st[p + ".to_out.weight"], st.get(p + ".to_out.bias"))


def feed_forward(st, p, x):
    return F.linear(F.gelu(F.linear(x, st[p + ".w1.weight"], st[p + ".w1.bias"])), st[p + ".w2.weight"], st[p + ".w2.bias"])


def layer_stack(st, cfg: PerformerConfig, x):
    for i in range(cfg.depth):
        p = f"performer.net.layers.{i}"
        if cfg.use_rezero:  # performer_pytorch.ReZero: fn(x) * g
            x = x + self_attention(st, p + ".0.fn", cfg, x) * st[p + ".0.g"]
            x = x + feed_forward(st, p + ".1.fn.fn", x) * st[p + ".1.g"]
        else:  # performer_pytorch.PreLayerNorm: fn(norm(x))
            x = x + self_attention(st, p + ".0.fn", cfg, F.layer_norm(x, (cfg.dim,), st[p + ".0.norm.weight"], st[p + ".0.norm.bias"]))
            x = x + feed_forward(st, p + ".1.fn.fn", F.layer_norm(x, (cfg.dim,), st[p + ".1.norm.weight"], st[p + ".1.norm.bias"]))
    return x


def fixed_spatial_table(dim, seq):
    """FixedSpatialPositionalEmbedding (performer.py:43-66): sinusoid of the coordinate VALUE, gathered into sequence order, last row dropped.
    Upstream writes the outer product as einsum("d,j->ij", ...) (:51), a subscript typo that raises at construction, so the reference itself
    cannot run this setting; "i,j->ij" is the evident intent (the same expression as performer_pytorch's FixedPositionalEmbedding)."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    position = torch.arange(0, int(seq.max()) + 1, dtype=torch.float)
    sin_inp = torch.einsum("i,j->ij", position, inv_freq)[seq.long(), :]
    return torch.cat((sin_inp.sin(), sin_inp.cos()), dim=-1)[:-1]


def fixed_position_table(dim, max_seq_len):
    """performer_pytorch 1.0.11 FixedPositionalEmbedding (the wrapper's `fixed_position_emb=True`, performer.py:138-140): sin | cos of position x inverse frequency."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    position = torch.arange(0, max_seq_len, dtype=torch.float)
    sin_inp = torch.einsum("i,j->ij", position, inv_freq)
    return torch.cat((sin_inp.sin(), sin_inp.cos()), dim=-1)


def embed(st, cfg: PerformerConfig, tokens, spatial_index_sequences, conditionings=None, conditioning_type="none", spatial_position_emb="absolute", emb_mask=None):
    """performer.py:241-266: token emb + zero-front-padded spatial embeddings (learned `absolute` tables indexed by coordinate, or `fixed`
    sinusoids) [+ conditioning: BOS replacement or prepending] + absolute positional embedding."""
    b, n = tokens.shape
    x = F.embedding(tokens, st["token_emb.weight"])
    for a, seq in enumerate(spatial_index_sequences):
        if spatial_position_emb == "fixed":
            sc = fixed_spatial_table(cfg.dim, seq)[None, : n - 1]
        else:
            sc = F.embedding(seq[:-1], st[f"spatial_position_emb.{a}.emb.weight"])[None, : n - 1]
        x = x + F.pad(sc, (0, 0, 1, 0, 0, 0))
    if conditionings and conditioning_type == "bos_replacement":       # performer.py:252-261
        c = sum(F.embedding(cond, st[f"conditioning_emb.{i}.weight"])[:, 0, :] for i, cond in enumerate(conditionings))
        x = torch.cat((c[:, None, :], x[:, 1:, :]), dim=1)
    elif conditionings and conditioning_type == "prepending":          # performer.py:262-264 (the last conditioning ends up first)
        for i, cond in enumerate(conditionings):
            x = torch.cat((F.embedding(cond, st[f"conditioning_emb.{i}.weight"]), x), dim=1)
    if cfg.axial_position_emb:
        shape = cfg.axial_position_shape or (math.ceil(cfg.max_seq_len / 64), 64)       # performer.py:141-144
        pos = axial_position_table(st, shape)
    else:
        pos = st["pos_emb.emb"] if "pos_emb.emb" in st else st["pos_emb.emb.weight"]      # the fixed sinusoidal buffer, or the learned table
    x = x + pos[: x.shape[1]][None]
    if emb_mask is not None:           # nn.Dropout(emb_dropout) in training (performer.py:201,270): the mask = keep / (1 - p), handed in by the test
        x = x * emb_mask
    return x


def forward(st, cfg: PerformerConfig, tokens, spatial_index_sequences, conditionings=None, conditioning_type="none", spatial_position_emb="absolute", emb_mask=None):
    x = embed(st, cfg, tokens, spatial_index_sequences, conditionings, conditioning_type, spatial_position_emb, emb_mask)
    x = layer_stack(st, cfg, x)
    x = F.layer_norm(x, (cfg.dim,), st["norm.weight"], st["norm.bias"])
    if conditionings and conditioning_type == "prepending":            # performer.py:279-281
        x = x[:, len(conditionings):, :]
    if cfg.tie_embed:                                                  # performer.py:286-288
        return x @ st["token_emb.weight"].t()
    return F.linear(x, st["to_out.weight"], st["to_out.bias"])


def ce_loss(logits, target):
    """losses/transformer/transformer.py:24-33 with the inferer's transpose (inferer/transformer.py:28-29)."""
    return F.cross_entropy(logits.transpose(1, 2).float(), target.long(), reduction="mean")


def spatial_index_sequences(spatial_shape, ordering):
    """performer.py:159-171: per-axis ij-meshgrid coordinate, flattened and re-ordered."""
    import numpy as np
    coords = np.array(np.meshgrid(*[np.arange(s) for s in spatial_shape], indexing="ij"))
    return [torch.from_numpy(coords[a].flatten()[ordering]).long() for a in range(len(spatial_shape))]
