"""Host-side mirror of the image partition the render and resolve kernels use.

The image is cut into 8x8-pixel tiles (one wave64 each).  Tile (tx, ty) belongs to rank
(tx + ty) mod world; a rank stores its tiles row by row, `tiles_per_row_local` per tile row,
64 pixels (lane = x%8 + 8*(y%8)) of 4 floats each.  Every pixel is summed by exactly one rank
in sample order, so the gathered image does not depend on the number of ranks.
"""
import numpy as np

TILE = 8


def geometry(width, height, world):
    tiles_x = (width + TILE - 1) // TILE
    tiles_y = (height + TILE - 1) // TILE
    tprl = (tiles_x + world - 1) // world
    return tiles_x, tiles_y, tprl


def payload_float4s(width, height, world):
    _, tiles_y, tprl = geometry(width, height, world)
    return tiles_y * tprl * 64


def _pixel_maps(width, height, world):
    tiles_x, tiles_y, tprl = geometry(width, height, world)
    y, x = np.mgrid[0:height, 0:width]
    tx, ty = x // TILE, y // TILE
    owner = (tx + ty) % world
    base = (owner - ty) % world
    k = (tx - base) // world
    lane = (x % TILE) + TILE * (y % TILE)
    slot = (ty * tprl + k) * 64 + lane
    return owner, slot


def payload_from_image(img, rank, world):
    """The accumulation buffer rank `rank` would hold for the row-major image `img` (HxWx4)."""
    h, w = img.shape[:2]
    owner, slot = _pixel_maps(w, h, world)
    out = np.zeros((payload_float4s(w, h, world), 4), img.dtype)
    m = owner == rank
    out[slot[m]] = img[m]
    return out


def image_from_payloads(payloads, width, height, world):
    """De-interleaves `world` rank payloads (world x N x 4) into the row-major image."""
    owner, slot = _pixel_maps(width, height, world)
    payloads = np.asarray(payloads).reshape(world, -1, 4)
    return payloads[owner, slot]
