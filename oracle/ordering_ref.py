"""ORACLE (test infrastructure only) -- restatement of the latent-grid -> sequence ordering.

Follows ``/root/reference/src/networks/transformers/img2seq_ordering.py:24-201`` and the generalized
Hilbert curve of ``/root/reference/gilbert/gilbert{2d,3d}.py`` (BSD-2, J. Cerveny's published algorithm),
written as plain loops / recursion over integer tuples.  Parity: PINNED by ``tests/golden/ordering.npz``
(permutations + SHA-1s produced by the reference itself).

Only tests / smoke / bench's cpu_baseline may import this module.
"""
from __future__ import annotations

import numpy as np


def _sgn(v):
    return tuple((c > 0) - (c < 0) for c in v)


def _add(*vs):
    return tuple(sum(c) for c in zip(*vs))


def _neg(v):
    return tuple(-c for c in v)


def _sub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def _half(v):
    return tuple(c // 2 for c in v)  # floor division, as the reference (gilbert3d.py:68-70)


def _len(v):
    return abs(sum(v))


def _walk(p, step, n, out):
    for _ in range(n):
        out.append(p)
        p = _add(p, step)


def _g3(p, a, b, c, out):
    """gilbert3d.py:36-163 -- fill the box spanned by axis vectors a (major), b, c from corner p."""
    w, h, d = _len(a), _len(b), _len(c)
    da, db, dc = _sgn(a), _sgn(b), _sgn(c)
    if h == 1 and d == 1:
        return _walk(p, da, w, out)
    if w == 1 and d == 1:
        return _walk(p, db, h, out)
    if w == 1 and h == 1:
        return _walk(p, dc, d, out)
    a2, b2, c2 = _half(a), _half(b), _half(c)
    if _len(a2) % 2 and w > 2:
        a2 = _add(a2, da)
    if _len(b2) % 2 and h > 2:
        b2 = _add(b2, db)
    if _len(c2) % 2 and d > 2:
        c2 = _add(c2, dc)
    if 2 * w > 3 * h and 2 * w > 3 * d:  # wide: split along a only
        _g3(p, a2, b, c, out)
        _g3(_add(p, a2), _sub(a, a2), b, c, out)
    elif 3 * h > 4 * d:  # do not split c
        _g3(p, b2, c, a2, out)
        _g3(_add(p, b2), a, _sub(b, b2), c, out)
        _g3(_add(p, _sub(a, da), _sub(b2, db)), _neg(b2), c, _neg(_sub(a, a2)), out)
    elif 3 * d > 4 * h:  # do not split b
        _g3(p, c2, a2, b, out)
        _g3(_add(p, c2), a, b, _sub(c, c2), out)
        _g3(_add(p, _sub(a, da), _sub(c2, dc)), _neg(c2), _neg(_sub(a, a2)), b, out)
    else:  # regular: split all three
        _g3(p, b2, c2, a2, out)
        _g3(_add(p, b2), c, a2, _sub(b, b2), out)
        _g3(_add(p, _sub(b2, db), _sub(c, dc)), a, _neg(b2), _neg(_sub(c, c2)), out)
        _g3(_add(p, _sub(a, da), b2, _sub(c, dc)), _neg(c), _neg(_sub(a, a2)), _sub(b, b2), out)
        _g3(_add(p, _sub(a, da), _sub(b2, db)), _neg(b2), c2, _neg(_sub(a, a2)), out)


def gilbert3d(width, height, depth):
    """gilbert3d.py:6-29 -- the longest side becomes the major axis."""
    out = []
    if width >= height and width >= depth:
        _g3((0, 0, 0), (width, 0, 0), (0, height, 0), (0, 0, depth), out)
    elif height >= width and height >= depth:
        _g3((0, 0, 0), (0, height, 0), (width, 0, 0), (0, 0, depth), out)
    else:
        _g3((0, 0, 0), (0, 0, depth), (width, 0, 0), (0, height, 0), out)
    return out


def _g2(p, a, b, out):
    """gilbert2d.py generate2d."""
    w, h = _len(a), _len(b)
    da, db = _sgn(a), _sgn(b)
    if h == 1:
        return _walk(p, da, w, out)
    if w == 1:
        return _walk(p, db, h, out)
    a2, b2 = _half(a), _half(b)
    w2, h2 = _len(a2), _len(b2)
    if 2 * w > 3 * h:
        if w2 % 2 and w > 2:
            a2 = _add(a2, da)
        _g2(p, a2, b, out)
        _g2(_add(p, a2), _sub(a, a2), b, out)
    else:
        if h2 % 2 and h > 2:
            b2 = _add(b2, db)
        _g2(p, b2, a2, out)
        _g2(_add(p, b2), a, _sub(b, b2), out)
        _g2(_add(p, _sub(a, da), _sub(b2, db)), _neg(b2), _neg(_sub(a, a2)), out)


def gilbert2d(width, height):
    out = []
    if width >= height:
        _g2((0, 0), (width, 0), (0, height), out)
    else:
        _g2((0, 0), (0, height), (width, 0), out)
    return out


def coordinate_sequence(ordering_type, shape):
    """img2seq_ordering.py:142-201 -- list of grid coordinates in visiting order."""
    nd = len(shape)
    if ordering_type == "hilbert_curve":
        return gilbert3d(*shape) if nd == 3 else gilbert2d(*shape)
    seq = []
    rows, cols = shape[0], shape[1]
    depths = shape[2] if nd == 3 else None
    snake = ordering_type == "s_curve"
    for r in range(rows):
        cs = range(cols - 1, -1, -1) if (snake and r % 2) else range(cols)
        for c in cs:
            if depths:
                ds = range(depths - 1, -1, -1) if (snake and c % 2) else range(depths)
                for d in ds:
                    seq.append((r, c, d))
            else:
                seq.append((r, c))
    if ordering_type == "random":
        arr = np.array(seq)
        np.random.shuffle(arr)  # global numpy RNG, as img2seq_ordering.py:192
        return [tuple(int(v) for v in e) for e in arr]
    if ordering_type not in ("raster_scan", "s_curve"):
        raise AssertionError(ordering_type)
    return seq


def ordering(ordering_type, spatial_dims, dimensions, reflected_spatial_dims=(), transpositions_axes=(), rot90_axes=(),
             transformation_order=("transpose", "rotate_90", "reflect")):
    """img2seq_ordering.py:79-138 -> (sequence_ordering, revert_ordering) as int64 arrays."""
    assert len(dimensions) == spatial_dims + 1
    shape = tuple(dimensions[1:])
    t = np.arange(int(np.prod(shape))).reshape(shape)
    for tr in transformation_order:
        if tr == "transpose":
            for axes in transpositions_axes:
                t = np.transpose(t, axes=axes)
        elif tr == "rotate_90":
            for axes in rot90_axes:
                t = np.rot90(t, axes=axes)
        elif tr == "reflect":
            for ax, flag in enumerate(reflected_spatial_dims):
                if flag:
                    t = np.flip(t, axis=ax)
        else:
            raise ValueError(tr)
    seq = coordinate_sequence(ordering_type, t.shape)
    order = np.array([t[tuple(e)] for e in seq], dtype=np.int64)
    return order, np.argsort(order)


def prepare_batch(quantization: np.ndarray, index_sequence: np.ndarray, vocab_size: int):
    """src/utils/transformer.py:239-282 -- flatten, reorder, left-pad BOS, shift."""
    enc = quantization.reshape(quantization.shape[0], -1)[:, index_sequence]
    enc = np.concatenate([np.full((enc.shape[0], 1), vocab_size, dtype=np.int64), enc.astype(np.int64)], axis=1)
    return enc[:, :-1], enc[:, 1:]
