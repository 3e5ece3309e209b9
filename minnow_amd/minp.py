"""minp particle vectors over the C ABI: the host side of minp.Writer.Vectors / minp.Reader.Vectors
(go/minp/minp.go:85-118, 170-198) with the particle grid resident on the GPU.

File I/O, headers and the block index stay with the container (the reference's minnow.Writer/Reader, or
include/minnow.hpp); this module produces / consumes the 3 * sub_cells^3 FloatGroup blocks of one file
cell in file order (component-major).
"""
import numpy as np

from . import _lib


def n_file(n_side, file_cells):
    """Cell.NFile (go/minp/minp.go): particles per side of one file cell"""
    return int(n_side) // int(file_cells)


def limits(ctx, vec_ptr, n, periodic, L, stream=None):
    """The (min, max) per component that Vectors hands to FloatGroup (minp.go:86-96): the box for a periodic
    snapshot, else bounds() with every maximum stepped by Nextafter32(max, 2*max)."""
    if periodic:
        return [0.0, 0.0, 0.0], [float(np.float32(L))] * 3
    mn, mx = ctx.vectors_bounds(vec_ptr, n, stream=stream)
    mx = [float(np.nextafter(np.float32(m), np.float32(2) * np.float32(m))) for m in mx]
    return mn, mx


def vectors_encode(ctx, vec_ptr, n_file_, sub_cells, dx, periodic, L, dst_ptr, slot_bytes, stream=None):
    """-> (metas in file order, low[3], high[3], pixels[3]).  Block (k, sc) lands at
    dst_ptr + (k * sub_cells**3 + sc) * slot_bytes."""
    n = n_file_ ** 3
    low, high = limits(ctx, vec_ptr, n, periodic, L, stream=stream)
    pixels = [_lib.lib().mnw_float_group_pixels(float(np.float32(low[k])), float(np.float32(high[k])), float(np.float32(dx)))
              for k in range(3)]
    metas = ctx.vectors_encode(vec_ptr, n_file_, sub_cells, low, high, pixels, dst_ptr, slot_bytes, stream=stream)
    return metas, low, high, pixels


def vectors_decode(ctx, src_ptr, slot_bytes, metas, n_file_, sub_cells, low, high, pixels, periodic, L, vec_out_ptr,
                   seed=0, first_block=0, stream=None):
    ctx.vectors_decode(src_ptr, slot_bytes, metas, n_file_, sub_cells, low, high, pixels, periodic, L, seed, first_block,
                       vec_out_ptr, stream=stream)
