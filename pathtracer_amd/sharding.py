"""Tile sharding layout shared by the C-ABI (``mi_render_device`` / ``mi_assemble_rgb8``)
and the multi-process driver code.

The image is cut into ``tile_w x tile_h`` tiles (default 32x32, the reference's work
unit: src/CpuTracer.cu:55-69), numbered row-major; tile ``k`` belongs to shard
``k % count``.  A shard's *local pixel* ``q`` is its ``q // (tile_w*tile_h)``-th owned
tile, offset ``q % (tile_w*tile_h)`` inside it (row-major).  The BVH is replicated;
the RNG is keyed by global pixel and sample, so the assembled image is identical for
any shard count.
"""
import numpy as np


def shard_tiles(width, height, rank, count, tile=(32, 32)):
    tw, th = tile
    tiles_x = (width + tw - 1) // tw
    tiles_y = (height + th - 1) // th
    return np.arange(rank, tiles_x * tiles_y, count), tiles_x


def shard_pixels(width, height, rank, count, tile=(32, 32)):
    tiles, _ = shard_tiles(width, height, rank, count, tile)
    return len(tiles) * tile[0] * tile[1]


def local_pixel_coords(width, height, rank, count, tile=(32, 32)):
    """(px, py, inside) arrays of length shard_pixels: image coordinates of every local pixel."""
    tw, th = tile
    tiles, tiles_x = shard_tiles(width, height, rank, count, tile)
    off = np.arange(tw * th)
    px = (tiles % tiles_x)[:, None] * tw + (off % tw)[None, :]
    py = (tiles // tiles_x)[:, None] * th + (off // tw)[None, :]
    px, py = px.reshape(-1), py.reshape(-1)
    return px, py, (px < width) & (py < height)


def extract_local(frame, rank, count, tile=(32, 32)):
    """Full frame [h,w,c] -> this shard's local buffer [shard_pixels, c] (zeros outside the image)."""
    h, w = frame.shape[:2]
    px, py, inside = local_pixel_coords(w, h, rank, count, tile)
    out = np.zeros((len(px),) + frame.shape[2:], frame.dtype)
    out[inside] = frame[py[inside], px[inside]]
    return out


def assemble(local_buffers, width, height, tile=(32, 32)):
    """List of per-shard local buffers [>=shard_pixels, c] (index = rank) -> frame [h,w,c]."""
    count = len(local_buffers)
    c = local_buffers[0].shape[1:]
    frame = np.zeros((height, width) + c, local_buffers[0].dtype)
    for rank, buf in enumerate(local_buffers):
        px, py, inside = local_pixel_coords(width, height, rank, count, tile)
        frame[py[inside], px[inside]] = np.asarray(buf)[: len(px)][inside]
    return frame
