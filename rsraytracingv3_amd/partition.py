"""Row-strip partition of a frame over ranks — host-side index maths.

The frame is cut into strips of ``strip_rows`` rows dealt round-robin over ``world``
ranks (contiguous bands would be badly imbalanced: sky rows finish after one segment).
Rank r renders strips s with s % world == r into a compact buffer whose local row order
is ascending global row.  Per-pixel results do not depend on the partition (the RNG state
is per pixel, random.hlsl:1-2,45-47), so reassembled frames are bit-identical to a
single-rank frame.  ``librt355`` implements the same mapping on the device side
(rt_set_partition)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def local_row_indices(height: int, rank: int, world: int, strip_rows: int) -> np.ndarray:
    """Global row of every local row of ``rank`` (ascending)."""
    rows = np.arange(height)
    return rows[(rows // strip_rows) % world == rank]


def local_row_count(height: int, rank: int, world: int, strip_rows: int) -> int:
    return int(local_row_indices(height, rank, world, strip_rows).size)


def max_local_rows(height: int, world: int, strip_rows: int) -> int:
    """Rows of the largest piece (rank 0 always owns the most) — the padded gather size."""
    return local_row_count(height, 0, world, strip_rows)


def assemble_rows(pieces: Sequence[np.ndarray], height: int, strip_rows: int) -> np.ndarray:
    """Interleave per-rank compact buffers (rows, W, C) back into the (height, W, C) frame.
    Pieces may carry padding rows at the end (padded gather); they are ignored."""
    world = len(pieces)
    out = np.empty((height,) + tuple(pieces[0].shape[1:]), dtype=pieces[0].dtype)
    for rank, piece in enumerate(pieces):
        idx = local_row_indices(height, rank, world, strip_rows)
        out[idx] = piece[: idx.size]
    return out
