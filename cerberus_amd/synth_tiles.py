"""Seeded STRUCTURED uint8 RGB tiles -- the inputs a slide really feeds the network besides texture: a smooth H&E-like stain field,
a tile half of which is glass (white), an all-white and an all-black tile.  Input generator for the parity fixtures
(the golden-vector generator's "structured" case), tests and bench.py; no checker lives here.

Integer arithmetic only (numpy int64, floor divisions): the build container, the conda interpreter and the GPU box rebuild the same bytes
whatever their libm does -- the fixture stores the tiles' sha256 beside the seed."""
import hashlib

import numpy as np


def _smooth_field(rs, hw, cells):
    """(hw, hw) int64 in 0..255: a (cells+1)^2 grid of seeded levels, bilinearly interpolated with integer weights."""
    g = rs.randint(0, 256, (cells + 1, cells + 1)).astype(np.int64)
    step = (hw + cells - 1) // cells
    idx = np.arange(hw, dtype=np.int64)
    c, f = idx // step, idx % step
    g00, g01 = g[c][:, c], g[c][:, c + 1]
    g10, g11 = g[c + 1][:, c], g[c + 1][:, c + 1]
    fy, fx = f[:, None], f[None, :]
    return (g00 * (step - fy) * (step - fx) + g01 * (step - fy) * fx + g10 * fy * (step - fx) + g11 * fy * fx) // (step * step)


def stain_field(hw, seed):
    """(hw, hw, 3) uint8: white light through two smooth stain densities -- eosin (pink: absorbs green most) everywhere, haematoxylin
    (blue-purple: absorbs red and green) in broad patches; the kind of low-frequency image a tile of stroma is, as far from
    `randint(0, 256)` noise as an input gets."""
    rs = np.random.RandomState(seed)
    eos = _smooth_field(rs, hw, 4)                       # 0..255 density
    hae = np.maximum(_smooth_field(rs, hw, 8) - 96, 0)   # patches: 0 over ~ 40 % of the tile, up to 159 elsewhere
    absorb_e = np.array([20, 140, 70], np.int64)         # per 256 of density, out of 255 of light
    absorb_h = np.array([170, 200, 60], np.int64)
    rgb = 245 - (eos[..., None] * absorb_e) // 256 - (hae[..., None] * absorb_h) // 160
    return np.clip(rgb, 0, 255).astype(np.uint8)


def structured_tiles(hw=256, seed=0):
    """(4, hw, hw, 3) uint8: [0] stain field, [1] stain field whose right half is glass (255), [2] all 255, [3] all 0."""
    a = stain_field(hw, seed)
    b = stain_field(hw, seed + 1)
    b[:, hw // 2:] = 255
    return np.stack([a, b, np.full((hw, hw, 3), 255, np.uint8), np.zeros((hw, hw, 3), np.uint8)])


def tiles_sha256(tiles):
    return hashlib.sha256(np.ascontiguousarray(tiles).tobytes()).hexdigest()
