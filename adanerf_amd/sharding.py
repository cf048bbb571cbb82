"""Host-side mirror of the image-strip sharding used for multi-GPU frames.

Rows are cut into strips of ``strip_rows`` rows; strip ``s`` belongs to rank ``s % world``.  A rank's
local ray list is its strips in ascending order, rows then columns.  The same mapping lives in the
device ray generator (``ray_pixel`` in csrc/k_common.hip.hpp) and in ``assemble_strips_kernel``; this
module is what the host uses to size gather payloads and what the CPU tests check the protocol with.
"""
import numpy as np


def balanced_strip_rows(h: int, world: int, max_rows: int = 8) -> int:
    """Largest strip height <= ``max_rows`` that gives every rank the same number of full strips (800 rows on 8
    ranks: 5, not 8 -- 100 strips of 8 rows would leave ranks 0-3 with 13 strips and ranks 4-7 with 12, an 8% longer
    critical path).  Falls back to ``max_rows`` when no such height exists."""
    for sr in range(max_rows, 0, -1):
        if h % sr == 0 and (h // sr) % world == 0:
            return sr
    return max_rows


def rows_of_rank(h: int, strip_rows: int, world: int, rank: int) -> np.ndarray:
    """Global image rows owned by ``rank``, in local order."""
    n_strips = -(-h // strip_rows)
    rows = []
    for s in range(rank, n_strips, world):
        rows.extend(range(s * strip_rows, min((s + 1) * strip_rows, h)))
    return np.asarray(rows, dtype=np.int64)


def rays_local(w: int, h: int, strip_rows: int, world: int, rank: int) -> int:
    return int(rows_of_rank(h, strip_rows, world, rank).size) * w


def rays_local_max(w: int, h: int, strip_rows: int, world: int) -> int:
    return rays_local(w, h, strip_rows, world, 0)


def local_to_pixel(w: int, h: int, strip_rows: int, world: int, rank: int) -> np.ndarray:
    """Global pixel index (row * w + col) of every local ray of ``rank``."""
    rows = rows_of_rank(h, strip_rows, world, rank)
    return (rows[:, None] * w + np.arange(w, dtype=np.int64)[None, :]).reshape(-1)


def assemble(gathered: np.ndarray, w: int, h: int, strip_rows: int, world: int) -> np.ndarray:
    """[world, rays_local_max, C] rank-major padded payloads -> [h*w, C] row-major image
    (numpy mirror of adanerf_assemble_strips)."""
    out = np.zeros((h * w,) + gathered.shape[2:], dtype=gathered.dtype)
    for rank in range(world):
        pix = local_to_pixel(w, h, strip_rows, world, rank)
        out[pix] = gathered[rank, :pix.size]
    return out
