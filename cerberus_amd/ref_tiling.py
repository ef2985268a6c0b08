"""The REFERENCE's tiled nuclei post-processing of a slide (`--reference_tiling`), on the GPU.

By default this package labels a slide band in one pass with exact ownership of every instance (cerberus_amd/shard_postproc.py); the
reference instead post-processes the nuclei canvas in 4096 x 4096 tiles and repairs the seams with three more tile sets
(infer/wsi.py:642-684): vertical strips and horizontal strips astride the inner tile edges and `cross sections` at the inner corners,
each with its own rule for which instances of the tile are dropped and which of the already accumulated ones are removed
(`_process_tile_predictions`, infer/wsi.py:81-268).  That scheme loses the few instances that lie wholly inside a margin zone and touch
the edge line of the strip that should have re-found them; the band scheme keeps them -- so the two results are NOT identical, and a user
who needs the reference's own instance set bit for bit asks for it here.

What runs where: every tile is labelled by cerb_postproc_nuclei with `exact_ties` on (the tile is exactly what the reference hands to
skimage, so its heap order is reproducible) and turned into the instance dictionary by cerb_inst_table / cerb_inst_contour_*; the margin
logic is a few vectorised closed-interval box tests on the host (shapely's `STRtree.query` = envelope intersection, touching included;
`box.contains(box)`).  Tile sets follow tiatoolbox 1.3.1 `NucleusInstanceSegmentor._get_tile_info` (un-vendored in the reference's tree,
absent from this image: restated, unpinned -- the test-suite compares this module with an independent CPU restatement of the same scheme).
"""
from collections import OrderedDict

import numpy as np
import torch

from .postproc import get_inst_info_dict, postproc_device


def _grid(image_wh, tile_wh):
    """Output boxes [x0, y0, x1, y1] of a regular tile grid from the origin; the last row / column may reach past the image."""
    w, h = image_wh
    tw, th = int(tile_wh[0]), int(tile_wh[1])
    xs = np.arange(max(-(-w // tw), 1), dtype=np.int64) * tw
    ys = np.arange(max(-(-h // th), 1), dtype=np.int64) * th
    gx, gy = np.meshgrid(xs, ys)
    gx, gy = gx.ravel(), gy.ravel()
    return np.stack([gx, gy, gx + tw, gy + th], axis=1)


def _hits(boxes, bounds):
    """closed-interval intersection of every box [n, 4] with one box (touching counts)"""
    return (boxes[:, 0] <= bounds[2]) & (bounds[0] <= boxes[:, 2]) & (boxes[:, 1] <= bounds[3]) & (bounds[1] <= boxes[:, 3])


def _inside(boxes, bounds):
    return (bounds[0] <= boxes[:, 0]) & (bounds[1] <= boxes[:, 1]) & (boxes[:, 2] <= bounds[2]) & (boxes[:, 3] <= bounds[3])


def get_tile_info(image_wh, tile_shape, margin, patch_output_shape):
    """[(boxes [n, 4], removal flags [n, 4] = (top, bottom, left, right))] for tile modes 0 (grid), 1 (vertical strips), 2 (horizontal
    strips), 3 (cross sections); a slide that fits one tile has the grid set alone with no flag raised."""
    w, h = int(image_wh[0]), int(image_wh[1])
    pos = np.array(patch_output_shape, dtype=np.int64).reshape(-1)
    pos = np.array([pos[0], pos[-1]])
    tile = (np.array(tile_shape, dtype=np.int64).reshape(-1)[:2] // pos) * pos  # whole output patches per tile
    grid = _grid((w, h), tile)
    if w <= tile[0] and h <= tile[1]:
        return [(grid, np.zeros((len(grid), 4), np.int64))]
    image_edges = [(0, 0, w, 0), (0, h, w, h), (0, 0, 0, h), (w, 0, w, h)]  # top, bottom, left, right

    def clear_at_image_edges(boxes, flags):
        for side, edge in enumerate(image_edges):
            flags[_hits(boxes, edge), side] = 0
        return flags

    flags = clear_at_image_edges(grid, np.ones((len(grid), 4), np.int64))
    info = [(grid, flags)]
    right_inner = np.nonzero(flags[:, 3])[0]  # a vertical strip astride every right edge that lies inside the slide
    vb = np.stack([grid[right_inner, 2] - margin, grid[right_inner, 1], grid[right_inner, 2] + margin, grid[right_inner, 3]], axis=1)
    vf = np.zeros((len(vb), 4), np.int64)
    vf[:, :2] = 1
    info.append((vb, clear_at_image_edges(vb, vf)))
    bottom_inner = np.nonzero(flags[:, 1])[0]  # a horizontal strip astride every bottom edge inside the slide
    hb = np.stack([grid[bottom_inner, 0], grid[bottom_inner, 3] - margin, grid[bottom_inner, 2], grid[bottom_inner, 3] + margin], axis=1)
    hf = np.zeros((len(hb), 4), np.int64)
    hf[:, 2:] = 1
    info.append((hb, clear_at_image_edges(hb, hf)))
    corner = np.nonzero(flags[:, 1] * flags[:, 3])[0]  # a square of four margins around every inner bottom-right corner
    cb = np.stack([grid[corner, 2] - 2 * margin, grid[corner, 3] - 2 * margin, grid[corner, 2] + 2 * margin, grid[corner, 3] + 2 * margin], axis=1)
    info.append((cb, np.ones((len(cb), 4), np.int64)))
    return info


def _tile_instances(inst_canvas, type_canvas, bounds, exact_ties):
    """Label one tile on the GPU -> (list of per-instance dictionaries in TILE coordinates, boxes [n, 4] as x0, y0, x1, y1)."""
    H, W = int(inst_canvas.shape[0]), int(inst_canvas.shape[1])
    x0, y0, x1, y1 = max(int(bounds[0]), 0), max(int(bounds[1]), 0), min(int(bounds[2]), W), min(int(bounds[3]), H)
    if x1 <= x0 or y1 <= y0:
        return [], np.zeros((0, 4), np.int64)
    lab, _ = postproc_device(inst_canvas[y0:y1, x0:x1], "Nuclei", exact_ties=exact_ties)
    tmap = None if type_canvas is None else type_canvas[y0:y1, x0:x1].contiguous()
    info = get_inst_info_dict(lab, tmap, flat_box=True)
    items = list(info.values())
    boxes = np.array([np.asarray(v["box"]) for v in items], dtype=np.int64).reshape(-1, 4)
    return items, boxes


def process_tile_predictions(inst_canvas, type_canvas, tile_bounds, tile_flag, tile_mode, ref_boxes, margin, exact_ties=True):
    """infer/wsi.py:81-268 for one tile.  ref_boxes: [k, 4] boxes (slide coordinates) of what has been accumulated before this tile SET.
    -> (kept instance dictionaries in slide coordinates, indices into ref_boxes to remove)."""
    tl = np.array([int(tile_bounds[0]), int(tile_bounds[1])], dtype=np.int64)
    w, h = int(tile_bounds[2]) - int(tile_bounds[0]), int(tile_bounds[3]) - int(tile_bounds[1])
    items, boxes = _tile_instances(inst_canvas, type_canvas, tile_bounds, exact_ties)
    if not items:
        return [], np.zeros(0, np.int64)
    m = int(margin)
    boundary_lines = [(0, 0, w, 1), (0, h - 1, w, h), (0, 0, 1, h), (w - 1, 0, w, h)]
    margin_boxes = [(0, 0, w, m), (0, h - m, w, h), (0, 0, m, h), (w - m, 0, w, h)]
    drop = np.zeros(len(items), bool)
    if tile_mode in (0, 3):  # grid tiles / cross sections: instances lying wholly inside a flagged margin zone
        for side, zone in enumerate(margin_boxes):
            if tile_flag[side] or tile_mode == 3:
                drop |= _hits(boxes, zone) & _inside(boxes, zone)
    elif tile_mode in (1, 2):  # strips: everything touching a flagged margin zone, or the one-pixel line of an unflagged side
        for side, flag in enumerate(tile_flag):
            drop |= _hits(boxes, margin_boxes[side] if flag else boundary_lines[side])
    else:
        raise ValueError("Unknown tile mode %r." % (tile_mode,))
    remove = np.zeros(0, np.int64)
    if tile_mode == 3 and len(ref_boxes):  # a cross section also evicts accumulated instances touching its inner margin lines
        lines = [(m, m, w - m, m), (m, h - m, w - m, h - m), (m, m, m, h - m), (w - m, m, w - m, h - m)]
        sel = np.zeros(len(ref_boxes), bool)
        for a, b, c, d in lines:
            sel |= _hits(ref_boxes, (a + tl[0], b + tl[1], c + tl[0], d + tl[1]))
        remove = np.nonzero(sel)[0]
    off = np.concatenate([tl, tl])
    kept = []
    for i in np.nonzero(~drop)[0].tolist():
        v = dict(items[i])
        v["box"] = np.asarray(v["box"]) + off
        v["centroid"] = np.asarray(v["centroid"]) + tl
        v["contour"] = np.asarray(v["contour"]) + tl
        kept.append(v)
    return kept, remove


def reference_tiled_nuclei(inst_canvas, type_canvas=None, tile_shape=4096, margin=64, patch_output_shape=144, exact_ties=True):
    """The nuclei loop of infer/wsi.py:642-684 over a device-resident INST canvas [H, W, 2] (and uint8 TYPE canvas [H, W]):
    OrderedDict {uuid4 hex -> {'box': [x1, y1, x2, y2], 'centroid', 'contour', 'type', 'type_prob'}} in slide coordinates.
    All tiles of a set see the accumulated dictionary as it was BEFORE the set (the reference merges futures after the set's tiles have been
    submitted), additions and evictions are applied in tile order."""
    import uuid

    assert inst_canvas.is_cuda and inst_canvas.dim() == 3
    H, W = int(inst_canvas.shape[0]), int(inst_canvas.shape[1])
    acc = OrderedDict()
    shape2 = [tile_shape, tile_shape] if np.isscalar(tile_shape) else tile_shape
    pos2 = [patch_output_shape, patch_output_shape] if np.isscalar(patch_output_shape) else patch_output_shape
    for mode, (bounds, flags) in enumerate(get_tile_info((W, H), shape2, margin, pos2)):
        keys = list(acc.keys())
        ref_boxes = np.array([np.asarray(acc[k]["box"]) for k in keys], dtype=np.int64).reshape(-1, 4)
        results = [process_tile_predictions(inst_canvas, type_canvas, tb, tf, mode, ref_boxes, margin, exact_ties) for tb, tf in zip(bounds, flags)]
        for kept, remove in results:
            for v in kept:
                acc[uuid.uuid4().hex] = v
            for r in remove.tolist():
                acc.pop(keys[r], None)
    torch.cuda.synchronize()
    return acc
