"""Latent grid -> 1-D token sequence permutation (and its inverse).

Mirror of the reference plugin ``Ordering`` (``src/networks/transformers/img2seq_ordering.py:24-201``): same
constructor arguments, validation errors, attributes (``dimensions``, ``template``) and accessors, so the
Performer wrapper, batch preparation and ``sample()`` use it unchanged.  The permutation is integer host work done
once at start-up; it is built with vectorised numpy index arithmetic instead of Python coordinate loops.
"""
from __future__ import annotations

from enum import Enum
from typing import Sequence, Tuple

import numpy as np
import torch

from .gilbert import gilbert2d, gilbert3d


class OrderingType(Enum):
    RASTER_SCAN = "raster_scan"
    S_CURVE = "s_curve"
    RANDOM = "random"
    HILBERT = "hilbert_curve"


class OrderingTransformations(Enum):
    ROTATE_90 = "rotate_90"
    TRANSPOSE = "transpose"
    REFLECT = "reflect"


def _grid_coords(shape: Sequence[int]) -> np.ndarray:
    """All coordinates of ``shape`` in C (raster) order, ``[prod(shape), ndim]``."""
    return np.stack(np.unravel_index(np.arange(int(np.prod(shape))), shape), axis=1).astype(np.int64)


class Ordering:
    def __init__(
        self,
        ordering_type: str,
        spatial_dims: int,
        dimensions: Tuple[int, ...],
        reflected_spatial_dims: Tuple[bool, ...],
        transpositions_axes: Tuple[Tuple[int, ...], ...],
        rot90_axes: Tuple[Tuple[int, ...], ...],
        transformation_order: Tuple[str, ...] = (
            OrderingTransformations.TRANSPOSE.value,
            OrderingTransformations.ROTATE_90.value,
            OrderingTransformations.REFLECT.value,
        ),
    ):
        valid_types = [e.value for e in OrderingType]
        assert ordering_type in valid_types, (
            f"ordering_type must be one of the following {valid_types}, but got {ordering_type}."
        )
        assert len(dimensions) == spatial_dims + 1, f"Dimensions must have length {spatial_dims + 1}."
        if len(set(transformation_order)) != len(transformation_order):
            raise ValueError(f"No duplicates are allowed. Received {transformation_order}.")
        valid_tr = [t.value for t in OrderingTransformations]
        for tr in transformation_order:
            if tr not in valid_tr:
                raise ValueError(f"Valid transformations are {valid_tr} but received {tr}.")

        self.ordering_type = ordering_type
        self.spatial_dims = spatial_dims
        self.dimensions = dimensions
        self.reflected_spatial_dims = reflected_spatial_dims
        self.transpositions_axes = transpositions_axes
        self.rot90_axes = rot90_axes
        self.transformation_order = transformation_order

        self.template = self._create_template()
        self._sequence_ordering = self._create_ordering()
        self._revert_sequence_ordering = np.argsort(self._sequence_ordering)

    # ---------------------------------------------------------------- public surface
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return x[self._sequence_ordering]

    def get_sequence_ordering(self) -> np.ndarray:
        return self._sequence_ordering

    def get_revert_sequence_ordering(self) -> np.ndarray:
        return self._revert_sequence_ordering

    # ---------------------------------------------------------------- construction
    def _create_template(self) -> np.ndarray:
        spatial = self.dimensions[1:]
        return np.arange(int(np.prod(spatial))).reshape(*spatial)

    def _create_ordering(self) -> np.ndarray:
        t = self.template
        for tr in self.transformation_order:
            if tr == OrderingTransformations.TRANSPOSE.value:
                for axes in self.transpositions_axes:
                    t = np.transpose(t, axes=axes)
            elif tr == OrderingTransformations.ROTATE_90.value:
                for axes in self.rot90_axes:
                    t = np.rot90(t, axes=axes)
            elif tr == OrderingTransformations.REFLECT.value:
                for axis, flag in enumerate(self.reflected_spatial_dims):
                    if flag:
                        t = np.flip(t, axis=axis)
        self.template = t
        coords = getattr(self, f"{self.ordering_type}_idx")(*t.shape)
        return np.asarray(t[tuple(coords.T)])

    # ---------------------------------------------------------------- coordinate generators
    @staticmethod
    def raster_scan_idx(rows: int, cols: int, depths: int = None) -> np.ndarray:
        return _grid_coords((rows, cols, depths) if depths else (rows, cols))

    @staticmethod
    def s_curve_idx(rows: int, cols: int, depths: int = None) -> np.ndarray:
        idx = _grid_coords((rows, cols, depths) if depths else (rows, cols))
        r = idx[:, 0]
        # odd rows walk the columns backwards; inside a column of odd *actual* index the depth runs backwards
        idx[:, 1] = np.where(r % 2 == 1, cols - 1 - idx[:, 1], idx[:, 1])
        if depths:
            idx[:, 2] = np.where(idx[:, 1] % 2 == 1, depths - 1 - idx[:, 2], idx[:, 2])
        return idx

    @staticmethod
    def random_idx(rows: int, cols: int, depths: int = None) -> np.ndarray:
        idx = _grid_coords((rows, cols, depths) if depths else (rows, cols))
        np.random.shuffle(idx)  # global numpy RNG, like the reference (seed-dependent by design)
        return idx

    @staticmethod
    def hilbert_curve_idx(rows: int, cols: int, depths: int = None) -> np.ndarray:
        return gilbert3d(rows, cols, depths) if depths else gilbert2d(rows, cols)
